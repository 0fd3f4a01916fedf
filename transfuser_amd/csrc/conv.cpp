// Convolution fwd / dgrad / wgrad as implicit GEMM on the MFMA engine.
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

static int check_geom(const tf_conv_geom* g, const char* who) {
    TF_REQUIRE(g, "%s: null geometry", who);
    TF_REQUIRE(g->groups >= 1 && g->Cin % g->groups == 0 && g->Cout % g->groups == 0, "%s: bad groups", who);
    TF_REQUIRE(g->ksize >= 1 && g->ksize <= 7 && g->stride >= 1 && g->pad >= 0, "%s: ksize %d / stride %d / pad %d unsupported (1 <= ksize <= 7)", who, g->ksize, g->stride, g->pad);
    TF_REQUIRE(g->Ho == (g->Hi + 2 * g->pad - g->ksize) / g->stride + 1 && g->Wo == (g->Wi + 2 * g->pad - g->ksize) / g->stride + 1,
               "%s: output size mismatch", who);
    return 0;
}

static int conv2d_fwd_impl(const tf_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int relu, float* colstat, int* colstat_nparts,
                           void* stream) {
    if (int e = check_geom(g, "tf_conv2d_fwd_f32")) return e;
    const int Cig = g->Cin / g->groups, Cog = g->Cout / g->groups, taps = g->ksize * g->ksize;
    const int M = g->B * g->Ho * g->Wo, K = taps * Cig;
    Im2colOp A;
    A.x = x; A.Hi = g->Hi; A.Wi = g->Wi; A.Ct = g->Cin; A.Ho = g->Ho; A.Wo = g->Wo; A.ks = g->ksize; A.stride = g->stride;
    A.pad = g->pad; A.Cg = Cig; A.rows = M; A.cols = K; A.coff = 0;
    A.vec = (aligned16(x) && Cig % 4 == 0 && g->Cin % 4 == 0) ? 1 : 0;
    PlainOp Bw;
    Bw.p = w; Bw.ld = K; Bw.rows = Cog; Bw.cols = K; Bw.s_outer = 0; Bw.s_inner = (long)Cog * K; Bw.inner = g->groups;
    Bw.vec = (aligned16(w) && K % 4 == 0) ? 1 : 0;
    GemmEpi ep;
    ep.C = y; ep.ldc = g->Cout; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = Cog; ep.inner = g->groups; ep.bias = bias; ep.sbias = Cog;
    ep.res = nullptr; ep.ldres = 0; ep.alpha = 1.f; ep.relu = relu; ep.mode = 0;
    if (colstat) { ep.stat = colstat; ep.stat_ld = g->Cout; ep.stat_nparts = colstat_nparts; *colstat_nparts = 0; }
    return launch_gemm<Im2colOp, true, PlainOp, true>(A, Bw, ep, M, Cog, K, g->groups, false, stream, "tf_conv2d_fwd_f32");
}

extern "C" int tf_conv2d_fwd_f32(const tf_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int relu, void* stream) {
    return conv2d_fwd_impl(g, x, w, bias, y, relu, nullptr, nullptr, stream);
}
extern "C" int tf_conv2d_fwd_colstat_f32(const tf_conv_geom* g, const float* x, const float* w, const float* bias, float* y, float* colstat,
                                         int* colstat_nparts, void* stream) {
    TF_REQUIRE(colstat && colstat_nparts, "tf_conv2d_fwd_colstat_f32: colstat / colstat_nparts missing");
    return conv2d_fwd_impl(g, x, w, bias, y, 0, colstat, colstat_nparts, stream);
}

extern "C" int tf_conv2d_dgrad_f32(const tf_conv_geom* g, const float* dy, const float* w, float* dx, int accumulate, void* stream) {
    if (int e = check_geom(g, "tf_conv2d_dgrad_f32")) return e;
    const int Cig = g->Cin / g->groups, Cog = g->Cout / g->groups, taps = g->ksize * g->ksize;
    const int M = g->B * g->Hi * g->Wi, K = taps * Cog;
    Im2colTOp A;
    A.dy = dy; A.Hi = g->Hi; A.Wi = g->Wi; A.Ct = g->Cout; A.Ho = g->Ho; A.Wo = g->Wo; A.ks = g->ksize; A.stride = g->stride;
    A.pad = g->pad; A.Cg = Cog; A.rows = M; A.cols = K; A.coff = 0;
    A.vec = (aligned16(dy) && Cog % 4 == 0 && g->Cout % 4 == 0) ? 1 : 0;
    WDgradOp Bw;
    Bw.w = w; Bw.taps = taps; Bw.Cog = Cog; Bw.Cig = Cig; Bw.rows = K; Bw.cols = Cig; Bw.gstride = (long)Cog * taps * Cig;
    Bw.vec = (aligned16(w) && Cig % 4 == 0) ? 1 : 0;
    GemmEpi ep;
    ep.C = dx; ep.ldc = g->Cin; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = Cig; ep.inner = g->groups; ep.bias = nullptr; ep.sbias = 0;
    ep.res = nullptr; ep.ldres = 0; ep.alpha = 1.f; ep.relu = 0; ep.mode = accumulate ? 1 : 0;
    return launch_gemm<Im2colTOp, true, WDgradOp, false>(A, Bw, ep, M, Cig, K, g->groups, false, stream, "tf_conv2d_dgrad_f32");
}

extern "C" int tf_conv2d_wgrad_f32(const tf_conv_geom* g, const float* dy, const float* x, float* dw, int accumulate, void* stream) {
    if (int e = check_geom(g, "tf_conv2d_wgrad_f32")) return e;
    const int Cig = g->Cin / g->groups, Cog = g->Cout / g->groups, taps = g->ksize * g->ksize;
    const int Mred = g->B * g->Ho * g->Wo, Ncols = taps * Cig;
    // dW[g][co][(tap,ci)] = sum_m dY[m][g*Cog + co] * im2col(X)[m][(tap,ci)]
    PlainOp A;  // rows = m (reduction), cols = co
    A.p = dy; A.ld = g->Cout; A.rows = Mred; A.cols = Cog; A.s_outer = 0; A.s_inner = Cog; A.inner = g->groups;
    A.vec = (aligned16(dy) && g->Cout % 4 == 0 && Cog % 4 == 0) ? 1 : 0;
    Im2colOp Bx;  // rows = m, cols = (tap, ci)
    Bx.x = x; Bx.Hi = g->Hi; Bx.Wi = g->Wi; Bx.Ct = g->Cin; Bx.Ho = g->Ho; Bx.Wo = g->Wo; Bx.ks = g->ksize; Bx.stride = g->stride;
    Bx.pad = g->pad; Bx.Cg = Cig; Bx.rows = Mred; Bx.cols = Ncols; Bx.coff = 0;
    Bx.vec = (aligned16(x) && Cig % 4 == 0 && g->Cin % 4 == 0) ? 1 : 0;
    GemmEpi ep;
    ep.C = dw; ep.ldc = Ncols; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = (long)Cog * Ncols; ep.inner = g->groups; ep.bias = nullptr; ep.sbias = 0;
    ep.res = nullptr; ep.ldres = 0; ep.alpha = 1.f; ep.relu = 0; ep.mode = accumulate ? 1 : 0;
    if (Cog <= 32) {
        // few output channels per group (RegNet group width 24, decoder / head tails with 32, 7, 1): compute dW^T -
        // rows = (tap, ci), cols = co - so Cog sits in a 32-wide column tile instead of a 128-row tile.
        ep.ldc = 1; ep.ldcj = Ncols;
        return launch_gemm<Im2colOp, false, PlainOp, false>(Bx, A, ep, Ncols, Cog, Mred, g->groups, true, stream, "tf_conv2d_wgrad_f32[swapped]");
    }
    return launch_gemm<PlainOp, false, Im2colOp, false>(A, Bx, ep, Cog, Ncols, Mred, g->groups, true, stream, "tf_conv2d_wgrad_f32");
}

// Stem convolutions on the model inputs (NCHW, Cin <= 4): 3x3 stride 2 pad 1, no bias; output NHWC.
static Im2colNchwOp make_stem(const tf_conv_geom* g, const float* s0, int C0, const float* s1, int C1, int normalize) {
    Im2colNchwOp A;
    A.s0 = s0; A.s1 = s1; A.C0 = C0; A.C1 = C1; A.Hi = g->Hi; A.Wi = g->Wi; A.Ho = g->Ho; A.Wo = g->Wo; A.ks = g->ksize; A.stride = g->stride;
    A.pad = g->pad; A.Cg = g->Cin; A.rows = g->B * g->Ho * g->Wo; A.cols = g->ksize * g->ksize * g->Cin; A.vec = 0; A.normalize = normalize;
    const float mean[4] = {0.485f, 0.456f, 0.406f, 0.f}, stdv[4] = {0.229f, 0.224f, 0.225f, 1.f};
    for (int i = 0; i < 4; ++i) { A.mean[i] = mean[i]; A.stdv[i] = stdv[i]; }
    return A;
}

extern "C" int tf_stem_conv_fwd_f32(const tf_conv_geom* g, const float* s0, int C0, const float* s1, int C1, int normalize, const float* w,
                                    float* y, void* stream) {
    if (int e = check_geom(g, "tf_stem_conv_fwd_f32")) return e;
    TF_REQUIRE(s0 && w && y && g->groups == 1 && C0 + C1 == g->Cin && g->Cin <= 4 && (C1 == 0 || s1), "tf_stem_conv_fwd_f32: bad arguments");
    const int K = g->ksize * g->ksize * g->Cin, M = g->B * g->Ho * g->Wo;
    Im2colNchwOp A = make_stem(g, s0, C0, s1, C1, normalize);
    PlainOp Bw;
    Bw.p = w; Bw.ld = K; Bw.rows = g->Cout; Bw.cols = K; Bw.s_outer = 0; Bw.s_inner = 0; Bw.inner = 1; Bw.vec = 0;
    GemmEpi ep;
    ep.C = y; ep.ldc = g->Cout; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = 0; ep.inner = 1; ep.bias = nullptr; ep.sbias = 0; ep.res = nullptr; ep.ldres = 0;
    ep.alpha = 1.f; ep.relu = 0; ep.mode = 0;
    return launch_gemm<Im2colNchwOp, true, PlainOp, true>(A, Bw, ep, M, g->Cout, K, 1, false, stream, "tf_stem_conv_fwd_f32");
}

extern "C" int tf_stem_conv_wgrad_f32(const tf_conv_geom* g, const float* dy, const float* s0, int C0, const float* s1, int C1, int normalize,
                                      float* dw, int accumulate, void* stream) {
    if (int e = check_geom(g, "tf_stem_conv_wgrad_f32")) return e;
    TF_REQUIRE(s0 && dy && dw && g->groups == 1 && C0 + C1 == g->Cin && g->Cin <= 4 && (C1 == 0 || s1), "tf_stem_conv_wgrad_f32: bad arguments");
    const int K = g->ksize * g->ksize * g->Cin, M = g->B * g->Ho * g->Wo;
    PlainOp A;
    A.p = dy; A.ld = g->Cout; A.rows = M; A.cols = g->Cout; A.s_outer = 0; A.s_inner = 0; A.inner = 1;
    A.vec = (aligned16(dy) && g->Cout % 4 == 0) ? 1 : 0;
    Im2colNchwOp Bx = make_stem(g, s0, C0, s1, C1, normalize);
    GemmEpi ep;
    ep.C = dw; ep.ldc = K; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = 0; ep.inner = 1; ep.bias = nullptr; ep.sbias = 0; ep.res = nullptr; ep.ldres = 0;
    ep.alpha = 1.f; ep.relu = 0; ep.mode = accumulate ? 1 : 0;
    return launch_gemm<PlainOp, false, Im2colNchwOp, false>(A, Bx, ep, g->Cout, K, M, 1, true, stream, "tf_stem_conv_wgrad_f32");
}

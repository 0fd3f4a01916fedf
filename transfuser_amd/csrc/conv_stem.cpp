// Stem convolutions on the NCHW model inputs (normalize_imagenet folded into the loader) - forward and weight gradient.
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"
#include "conv_common.h"

using namespace tf;

namespace tf {   // stem_direct.cpp: the RegNet stems (K = ks^2 Cin <= 32, Cout = 32) without the im2col engine
bool stem_direct_ok(const tf_conv_geom* g, int C0, int C1);
long stem_direct_wgrad_ws_floats(const tf_conv_geom* g);
int stem_direct_fwd(const tf_conv_geom* g, const float* s0, int C0, const float* s1, int C1, int normalize, const float* w, float* y, void* stream);
int stem_direct_wgrad(const tf_conv_geom* g, const float* dy, const float* s0, int C0, const float* s1, int C1, int normalize, float* dw, int accumulate,
                      float* ws, void* stream);
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
    if (stem_direct_ok(g, C0, C1)) return stem_direct_fwd(g, s0, C0, s1, C1, normalize, w, y, stream);
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

// The same with caller scratch for the direct kernels' partial panels (tf_stem_conv_wgrad_ws_floats() floats, 0 = the engine path is used and no
// scratch is needed): the RegNet stems take the direct path, every other stem (7x7 ResNet, 4x4 ConvNeXt) the implicit-GEMM engine.
extern "C" long tf_stem_conv_wgrad_ws_floats(const tf_conv_geom* g, int C0, int C1) {
    return (g && stem_direct_ok(g, C0, C1)) ? stem_direct_wgrad_ws_floats(g) : 0;
}
extern "C" int tf_stem_conv_wgrad_ws_f32(const tf_conv_geom* g, const float* dy, const float* s0, int C0, const float* s1, int C1, int normalize, float* dw,
                                         int accumulate, float* ws, long ws_floats, void* stream) {
    if (int e = check_geom(g, "tf_stem_conv_wgrad_ws_f32")) return e;
    if (ws && stem_direct_ok(g, C0, C1) && ws_floats >= stem_direct_wgrad_ws_floats(g)) {
        TF_REQUIRE(s0 && dy && dw && C0 + C1 == g->Cin && (C1 == 0 || s1), "tf_stem_conv_wgrad_ws_f32: bad arguments");
        return stem_direct_wgrad(g, dy, s0, C0, s1, C1, normalize, dw, accumulate, ws, stream);
    }
    return tf_stem_conv_wgrad_f32(g, dy, s0, C0, s1, C1, normalize, dw, accumulate, stream);
}

// Convolution forward as implicit GEMM on the MFMA engine (im2col A-operand loader, weights as the [n][k] operand).
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"
#include "conv_common.h"

using namespace tf;

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


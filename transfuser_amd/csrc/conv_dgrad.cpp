// Convolution input gradient as implicit GEMM on the MFMA engine (transposed-im2col loader over dY, tap-flipped weight loader).
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"
#include "conv_common.h"

using namespace tf;

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


// Convolution weight gradient as implicit GEMM on the MFMA engine (contraction over the output pixels; im2col loader on the B side).
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"
#include "conv_common.h"

using namespace tf;

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


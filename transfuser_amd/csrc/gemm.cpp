// tf_gemm_f32: batched strided fp32 GEMM on the MFMA engine (plain operands).
#include "tf_gemm_engine.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

static PlainOp make_plain(const float* p, long ld, int rows, int cols, long so, long si, int inner, int batch) {
    PlainOp o;
    o.p = p; o.ld = ld; o.rows = rows; o.cols = cols; o.s_outer = so; o.s_inner = si; o.inner = inner > 0 ? inner : 1;
    bool v = aligned16(p) && (ld % 4 == 0) && (cols % 4 == 0);
    if (batch > 1) v = v && (so % 4 == 0) && (si % 4 == 0);
    o.vec = v ? 1 : 0;
    return o;
}

extern "C" int tf_gemm_f32(const tf_gemm_desc* d, void* stream) {
    TF_REQUIRE(d && d->a && d->b && d->c, "tf_gemm_f32: null operand");
    TF_REQUIRE(d->m >= 0 && d->n >= 0 && d->k >= 0 && d->batch >= 1, "tf_gemm_f32: bad sizes m=%d n=%d k=%d batch=%d", d->m, d->n,
               d->k, d->batch);
    const int inner = d->inner > 0 ? d->inner : 1;
    GemmEpi ep;
    ep.C = d->c; ep.ldc = d->ldc; ep.sc_outer = d->sc_outer; ep.sc_inner = d->sc_inner; ep.inner = inner;
    ep.bias = d->bias; ep.sbias = 0; ep.res = d->res; ep.ldres = d->ldres; ep.alpha = d->alpha; ep.relu = d->relu;
    ep.mode = d->accumulate ? 1 : 0;
    // A: KC when stored [m][k] (rows = i), IC when stored [k][m] (rows = k)
    PlainOp A = d->a_trans ? make_plain(d->a, d->lda, d->k, d->m, d->sa_outer, d->sa_inner, inner, d->batch)
                           : make_plain(d->a, d->lda, d->m, d->k, d->sa_outer, d->sa_inner, inner, d->batch);
    // B: KC when stored [n][k] (rows = j), IC when stored [k][n] (rows = k)
    PlainOp B = d->b_trans ? make_plain(d->b, d->ldb, d->k, d->n, d->sb_outer, d->sb_inner, inner, d->batch)
                           : make_plain(d->b, d->ldb, d->n, d->k, d->sb_outer, d->sb_inner, inner, d->batch);
    const bool sk = d->accumulate != 0;
    if (!d->a_trans && !d->b_trans) return launch_gemm<PlainOp, true, PlainOp, true>(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[nt]");
    if (!d->a_trans && d->b_trans) return launch_gemm<PlainOp, true, PlainOp, false>(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[nn]");
    if (d->a_trans && d->b_trans) return launch_gemm<PlainOp, false, PlainOp, false>(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[tn]");
    return launch_gemm<PlainOp, false, PlainOp, true>(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[tt]");
}

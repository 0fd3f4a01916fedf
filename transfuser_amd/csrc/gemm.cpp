// tf_gemm_f32: batched strided fp32 GEMM on the MFMA engine (plain operands).
#include "tf_gemm_engine.h"
#include <stdlib.h>
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace tf {
int gemm16_impl(const void* a16, const void* b16, float* c, int m, int n, int k, int lda, int ldb, int ldc, const float* bias, const float* res, int ldres, float alpha,
                int relu, int accumulate, const float* mask, int ldmask, int dtype, float* colstat, int* colstat_nparts, void* stream);
// the four operand layouts of the register-staged engine: one translation unit each (gemm_plain_{nt,nn,tn,tt}.cpp) so hipcc compiles them in parallel
int gemm_plain_nt(const PlainOp& a, const PlainOp& b, const GemmEpi& ep, int M, int N, int K, int batch, bool allow_splitk, void* stream, const char* what);
int gemm_plain_nn(const PlainOp& a, const PlainOp& b, const GemmEpi& ep, int M, int N, int K, int batch, bool allow_splitk, void* stream, const char* what);
int gemm_plain_tn(const PlainOp& a, const PlainOp& b, const GemmEpi& ep, int M, int N, int K, int batch, bool allow_splitk, void* stream, const char* what);
int gemm_plain_tt(const PlainOp& a, const PlainOp& b, const GemmEpi& ep, int M, int N, int K, int batch, bool allow_splitk, void* stream, const char* what);
int smallm_fwd(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* res, long ldres, float* y, long ldy, int M, int N,
               int K, int relu, void* stream);
int smallm_dgrad(const float* dy, long lddy, const float* w, long ldw, const float* res, long ldres, float* dx, long lddx, int M, int N, int K,
                 int accumulate, void* stream);
int smallm_wgrad(const float* dy, long lddy, const float* x, long ldx, float* dw, long lddw, int M, int N, int K, int accumulate, void* stream);
bool skinny_wgrad_ok(int no, int C, int rows, long lddy, long ldx, const float* x, int accumulate);
int skinny_wgrad(const float* dy, long lddy, const float* x, long ldx, float* dw, long lddw, int rows, int no, int C, void* stream);
}

static PlainOp make_plain(const float* p, long ld, int rows, int cols, long so, long si, int inner, int batch) {
    PlainOp o;
    o.p = p; o.ld = ld; o.rows = rows; o.cols = cols; o.s_outer = so; o.s_inner = si; o.inner = inner > 0 ? inner : 1;
    bool v = aligned16(p) && (ld % 4 == 0) && (cols % 4 == 0);
    if (batch > 1) v = v && (so % 4 == 0) && (si % 4 == 0);
    o.vec = v ? 1 : 0;
    return o;
}

// Scratch floats tf_gemm_f32 can use for this call's deterministic two-pass split-K (tf_gemm_desc.splitk_ws): 0 when the cached plan of the
// call's site / shape is not a two-pass plan (so the caller need not allocate anything), the S-slice size when it is, and the 4-slice
// maximum while the autotuner is on or a test pins a two-pass plan (the candidates need the scratch to be tried at all).
extern "C" long tf_gemm_splitk_ws_floats(const tf_gemm_desc* d) {
    if (!d || d->batch != 1 || d->m <= 0 || d->n <= 0 || d->k <= 0) return 0;
    const long slice = (long)d->m * twopass_ldws(d->n);
    GemmPlan p;
    const long sk_floats = (long)kStreamKMaxBlocks / 4 * 128 * 128;       // stream-K: <= 512 resident workgroups x one 128 x 128 partial tile (larger launches use smaller tiles)
    if (forced_plan(&p)) return p.splitk == kStreamK ? sk_floats : p.splitk >= kTwoPass ? 4 * slice : 0;
    const char* site = !d->a_trans ? (d->b_trans ? "tf_gemm_f32[nn]" : "tf_gemm_f32[nt]") : (d->b_trans ? "tf_gemm_f32[tn]" : "tf_gemm_f32[tt]");
    int M = d->m, N = d->n;
    if (d->a_trans && d->b_trans && d->m <= 32 && d->n >= 2 * d->m && !d->bias && !d->res && !d->relu) return 0;       // swapped (column-strided) output: no two-pass
    const bool sk_ok = d->accumulate && !d->bias && !d->res && !d->relu;
    const int acc = (d->accumulate ? (sk_ok ? 2 : 1) : 0) + 4 * (gemm_precision() == 3 ? 1 : gemm_precision());
    if (plan_lookup(site, M, N, d->k, 1, acc, &p)) return p.splitk == kStreamK ? sk_floats : p.splitk >= kTwoPass ? (long)(p.splitk - kTwoPass) * slice : 0;
    return autotune_enabled() ? (4 * slice > sk_floats ? 4 * slice : sk_floats) : 0;
}

extern "C" int tf_gemm_f32(const tf_gemm_desc* d, void* stream) {
    TF_REQUIRE(d && d->a && d->b && d->c, "tf_gemm_f32: null operand");
    TF_REQUIRE(d->m >= 0 && d->n >= 0 && d->k >= 0 && d->batch >= 1, "tf_gemm_f32: bad sizes m=%d n=%d k=%d batch=%d", d->m, d->n,
               d->k, d->batch);
    const int inner = d->inner > 0 ? d->inner : 1;
    // per-sample matmuls (SE excitation, join MLP): rows = batch <= 16 -> streaming kernels instead of a 128-row MFMA tile
    if (d->batch == 1 && d->alpha == 1.0f && d->m > 0 && d->n > 0 && d->k > 0 && !d->mask && !d->colstat && !d->drop_seed) {
        if (!d->a_trans && !d->b_trans && d->m <= 16 && !d->accumulate)
            return smallm_fwd(d->a, d->lda, d->b, d->ldb, d->bias, d->res, d->ldres, d->c, d->ldc, d->m, d->n, d->k, d->relu, stream);
        if (!d->a_trans && d->b_trans && d->m <= 16 && !d->bias && !d->relu && !(d->accumulate && d->res))
            return smallm_dgrad(d->a, d->lda, d->b, d->ldb, d->res, d->ldres, d->c, d->ldc, d->m, d->k, d->n, d->accumulate, stream);
        if (d->a_trans && d->b_trans && d->k <= 16 && !d->bias && !d->res && !d->relu)
            return smallm_wgrad(d->a, d->lda, d->b, d->ldb, d->c, d->ldc, d->k, d->m, d->n, d->accumulate, stream);
        // few output features over many rows (head convolutions 64 -> 1 / 2 / 3 / 12): streaming reduction instead of a 160-way split 128 x 32 tile
        static const bool skinny = [] { const char* e = getenv("TF_SKINNY_WGRAD"); return e ? e[0] != '0' : true; }();
        if (skinny && d->a_trans && d->b_trans && !d->bias && !d->res && !d->relu && (gemm_precision() == 0 || gemm_precision() == 2) &&
            skinny_wgrad_ok(d->m, d->n, d->k, d->lda, d->ldb, d->b, d->accumulate))
            return skinny_wgrad(d->a, d->lda, d->b, d->ldb, d->c, d->ldc, d->k, d->m, d->n, stream);
    }
    GemmEpi ep;
    ep.C = d->c; ep.ldc = d->ldc; ep.ldcj = 1; ep.sc_outer = d->sc_outer; ep.sc_inner = d->sc_inner; ep.inner = inner;
    ep.bias = d->bias; ep.sbias = 0; ep.res = d->res; ep.ldres = d->ldres; ep.alpha = d->alpha; ep.relu = d->relu;
    ep.mode = d->accumulate ? 1 : 0;
    ep.mask = d->mask; ep.ldmask = d->ldmask;
    ep.sk_ws = d->splitk_ws; ep.sk_ws_floats = d->splitk_ws ? d->splitk_ws_floats : 0;
    ep.sk_flags = d->splitk_ws ? d->sk_flags : nullptr;
    if (d->colstat) {
        TF_REQUIRE(d->colstat_nparts && d->batch == 1 && !d->accumulate && !d->res && !d->relu && !d->mask && !d->a_trans,
                   "tf_gemm_f32: colstat needs a plain store (batch 1, no residual / ReLU / mask / accumulate), row-major A and colstat_nparts");
        ep.stat = d->colstat; ep.stat_ld = d->n; ep.stat_nparts = d->colstat_nparts;
        *d->colstat_nparts = 0;
    }
    TF_REQUIRE(!d->mask || (d->batch == 1 && !d->accumulate), "tf_gemm_f32: mask needs batch == 1 and a plain store");
    if (d->drop_seed) {
        TF_REQUIRE(d->batch == 1 && !d->accumulate && !d->a_trans && d->ldc == d->n && d->drop_p >= 0.f && d->drop_p < 1.f && !d->colstat && (long)d->m * d->n < 4294967296L,
                   "tf_gemm_f32: drop_seed needs batch 1, a plain store into a contiguous (m, n) output, row-major A and 0 <= drop_p < 1");
        ep.drop_seed = d->drop_seed; ep.drop_site = d->drop_site; ep.drop_thresh = (uint32_t)((double)d->drop_p * 4294967296.0); ep.drop_scale = 1.f / (1.f - d->drop_p);
    }
    // A: KC when stored [m][k] (rows = i), IC when stored [k][m] (rows = k)
    PlainOp A = d->a_trans ? make_plain(d->a, d->lda, d->k, d->m, d->sa_outer, d->sa_inner, inner, d->batch)
                           : make_plain(d->a, d->lda, d->m, d->k, d->sa_outer, d->sa_inner, inner, d->batch);
    // B: KC when stored [n][k] (rows = j), IC when stored [k][n] (rows = k)
    PlainOp B = d->b_trans ? make_plain(d->b, d->ldb, d->k, d->n, d->sb_outer, d->sb_inner, inner, d->batch)
                           : make_plain(d->b, d->ldb, d->n, d->k, d->sb_outer, d->sb_inner, inner, d->batch);
    const bool sk = d->accumulate != 0;
    if (!d->a_trans && !d->b_trans) return gemm_plain_nt(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[nt]");
    if (!d->a_trans && d->b_trans) return gemm_plain_nn(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[nn]");
    if (d->a_trans && d->b_trans) {
        // weight gradient with few output rows (m = out features <= 32 << n): compute C^T so the short side becomes the
        // 32-wide column tile instead of a 128-row tile (4x less MFMA padding); stores go through the column stride.
        if (d->m <= 32 && d->n >= 2 * d->m && !d->bias && !d->res && !d->relu) {
            ep.ldc = 1; ep.ldcj = d->ldc;
            return gemm_plain_tn(B, A, ep, d->n, d->m, d->k, d->batch, sk, stream, "tf_gemm_f32[tn,swapped]");
        }
        return gemm_plain_tn(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[tn]");
    }
    return gemm_plain_tt(A, B, ep, d->m, d->n, d->k, d->batch, sk, stream, "tf_gemm_f32[tt]");
}

// ---- packed 16-bit operands: C (op)= alpha * A16 . B16^T (+ bias) (+ res) (relu) (mask), fp32 accumulate / output.
// A16 [m][k] and B16 [n][k] are bf16 (dtype 1) or IEEE-half (dtype 2) matrices with k contiguous; k % 8 == 0, lda % 8 == 0, ldb % 8 == 0, 16-byte
// aligned bases (tf_cast16_f32 produces them).  Runs on the LDS-DMA "nt" kernels: the tiles are moved as bytes (a row of 2 BK halves has the
// byte geometry of BK floats), every ds_read_b128 fragment is one operand of v_mfma_f32_32x32x16_{bf16,f16}.
extern "C" int tf_gemm16_nt_colstat_f32(const void* a16, const void* b16, float* c, int m, int n, int k, int lda, int ldb, int ldc, int dtype, float* colstat,
                                        int* colstat_nparts, void* stream);
extern "C" int tf_gemm16_nt_f32(const void* a16, const void* b16, float* c, int m, int n, int k, int lda, int ldb, int ldc, const float* bias, const float* res,
                                int ldres, float alpha, int relu, int accumulate, const float* mask, int ldmask, int dtype, void* stream) {
    return tf::gemm16_impl(a16, b16, c, m, n, k, lda, ldb, ldc, bias, res, ldres, alpha, relu, accumulate, mask, ldmask, dtype, nullptr, nullptr, stream);
}
// plain store + the BatchNorm statistics of the output gathered by the epilogue (tf_gemm_desc.colstat semantics): the 1x1 convolutions of the
// RegNetY bottlenecks in the 16-bit storage modes
extern "C" int tf_gemm16_nt_colstat_f32(const void* a16, const void* b16, float* c, int m, int n, int k, int lda, int ldb, int ldc, int dtype, float* colstat,
                                        int* colstat_nparts, void* stream) {
    TF_REQUIRE(colstat && colstat_nparts, "tf_gemm16_nt_colstat_f32: colstat and colstat_nparts are required");
    return tf::gemm16_impl(a16, b16, c, m, n, k, lda, ldb, ldc, nullptr, nullptr, 0, 1.f, 0, 0, nullptr, 0, dtype, colstat, colstat_nparts, stream);
}
namespace tf {
int gemm16_impl(const void* a16, const void* b16, float* c, int m, int n, int k, int lda, int ldb, int ldc, const float* bias, const float* res, int ldres, float alpha,
                int relu, int accumulate, const float* mask, int ldmask, int dtype, float* colstat, int* colstat_nparts, void* stream) {
    TF_REQUIRE(a16 && b16 && c && m > 0 && n > 0 && k > 0, "tf_gemm16_nt_f32: bad sizes m=%d n=%d k=%d", m, n, k);
    const int pin = dtype >> 4;        // bits 4..: pin an LDS-DMA tile configuration (tests / tuning); 0 = heuristic
    dtype &= 15;
    TF_REQUIRE((dtype == 1 || dtype == 2) && k % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= k && ldb >= k && aligned16(a16) && aligned16(b16),
               "tf_gemm16_nt_f32: needs dtype 1 (bf16) / 2 (fp16), k, lda, ldb multiples of 8 and 16-byte aligned operands (got k=%d lda=%d ldb=%d)", k, lda, ldb);
    TF_REQUIRE(!mask || !accumulate, "tf_gemm16_nt_f32: mask needs a plain store");
    GemmEpi ep;
    ep.C = c; ep.ldc = ldc; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = 0; ep.inner = 1;
    ep.bias = bias; ep.sbias = 0; ep.res = res; ep.ldres = ldres; ep.alpha = alpha; ep.relu = relu; ep.mode = accumulate ? 1 : 0;
    ep.mask = mask; ep.ldmask = ldmask;
    ep.packed16 = dtype;
    if (colstat) { ep.stat = colstat; ep.stat_ld = n; ep.stat_nparts = colstat_nparts; *colstat_nparts = 0; }
    // the operands in units of 4 bytes
    PlainOp A = make_plain(reinterpret_cast<const float*>(a16), lda / 2, m, k / 2, 0, 0, 1, 1);
    PlainOp B = make_plain(reinterpret_cast<const float*>(b16), ldb / 2, n, k / 2, 0, 0, 1, 1);
    TF_REQUIRE(A.vec && B.vec && dma_eligible(A, B), "tf_gemm16_nt_f32: operands are not LDS-DMA eligible (alignment / 2 GiB)");
    // tile choice: 128 x 128 (BK = 64 halves, 2 stages) when that fills the 256 CUs, else 128 x 64, else 64 x 64 (TF_G16_KIND pins a kind for experiments)
    static const int forced = [] { const char* e = getenv("TF_G16_KIND"); return e ? atoi(e) : 0; }();
    const long t128 = (long)cdiv(m, 128) * cdiv(n, 128), t12864 = (long)cdiv(m, 128) * cdiv(n, 64);
    int kind = t128 >= 384 ? 5 : (t12864 >= 256 ? 3 : 2);
    if (forced >= 1 && forced <= kDmaKinds) kind = forced;
    if (pin >= 1 && pin <= kDmaKinds) kind = pin;
    // weight gradients (accumulate, rows contracted: k >> m, n): k-slices that atomically add, enough of them for two workgroups per CU
    int splitk = 1;
    if (accumulate && !bias && !res && !relu && !mask && alpha == 1.f) {
        static const int bm_of[] = {0, 128, 64, 128, 64, 128, 64, 64, 128}, bn_of[] = {0, 128, 64, 64, 128, 128, 64, 128, 64};
        const int kk = kind >= 1 && kind <= 8 ? kind : 8;
        const long tiles = (long)cdiv(m, bm_of[kk]) * cdiv(n, bn_of[kk]);
        if (tiles < 256) {
            long s = cdiv(512, tiles), smax = (k / 2) / 256;     // slices of >= 512 halves
            if (s > smax) s = smax;
            if (s > 128) s = 128;
            if (s >= 2) { splitk = (int)s; ep.mode = 2; }
        }
    }
    launch_dma_plan<true, true>(kind, A, B, ep, m, n, k / 2, 1, splitk, stream);
    return launch_status("tf_gemm16_nt_f32");
}
}  // namespace tf

// fp32 MFMA implicit-GEMM engine for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 157 TF peak).
//
//   C[z](i, j) (op)= alpha * sum_k A[z](i, k) * B[z](k, j)  (+ bias[j]) (+ res(i, j)) (relu)
//
// One kernel template serves every dense contraction on the TransFuser training path
// (SURVEY.md section 2.2 rows K1/K2/K3/K9/K10 and their dgrad/wgrad): the operands are described by
// *loader* structs that map a logical (row, col) - col contiguous in memory - to an address
// (plain strided matrices, NHWC im2col gathers for 3x3/1x1/strided/grouped convolutions, the
// transposed gather of conv dgrad, channels-last weights).  Each operand is either "KC" (rows =
// i or j, cols = k: needs a transpose on its way into LDS) or "IC" (rows = k, cols = i or j:
// copied straight).  LDS tiles are K-major ([BK][BM+4]) so an MFMA operand fetch is a
// conflict-free ds_read_b32 (lanes 0-31 consecutive, lanes 32-63 the next k row).
//
// Tile: BM x BN x BK (16 | 32), NT = 256 threads = 4 waves (the engine's dispatch; 512 is supported by the template), each wave owns
// (BM/WAVES_M) x (BN/WAVES_N) as 32x32 MFMA tiles.  Global->register prefetch of tile t+1 is issued before the MFMAs of tile t and
// written to the other LDS buffer after them: one barrier per K tile.  Plain operands take a mask-free fast loop (gemm_tile).
#pragma once
#include "tf_common.h"
#include <type_traits>
#include <cstdlib>

namespace tf {

constexpr int GEMM_PAD = 4;

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------- loaders
// All loaders are BRANCH-FREE: an out-of-range element loads from a clamped (always valid) address and is
// replaced by 0 with a select.  With branches around the loads hipcc has to wait for them at the merge point,
// i.e. before the MFMAs of the current tile; unconditional loads stay in flight across the whole compute phase and
// are only waited for where the next tile is written to LDS.
// loaders return the RAW (possibly clamped-address) data plus a 4-bit validity mask; gemm_tile applies the mask when it
// writes the tile to LDS, so nothing depends on the loaded registers until after the MFMAs of the current tile.

// CURSORS.  A thread's LDS slot keeps either its ROW fixed over the K loop and walks along the columns (KC operand) or keeps its
// COLUMN fixed and walks down the rows (IC operand).  The expensive index decompositions ((tap, ci) of a column, (b, oh, ow) of a
// pixel row: integer divisions by run-time values) are therefore done ONCE per slot (row() / col()), and stepping by BK is a few
// adds and compares (row_advance / col_advance).  Recomputing them per k-step cost ~390 VALU instructions per 16 MFMAs in the
// 3x3-conv kernels (2x the MFMA time).

struct PlainRow { const float* p; int r, ok; };

// X(r, c) = p[r*ld + c], r < rows, c < cols.  Batch z -> p + (z / inner) * s_outer + (z % inner) * s_inner.
struct PlainOp {
    const float* p; long ld; int rows, cols, vec; long s_outer, s_inner; int inner;
    typedef PlainRow Row;
    typedef int Col;
    static constexpr bool kFast = true;    // plain strided matrix: interior / clamped tiles can skip all masking (gemm_tile fast path)
    __device__ __forceinline__ void set_batch(int z) { p += (long)(z / inner) * s_outer + (long)(z % inner) * s_inner; }
    __device__ __forceinline__ Row row(int r) const { Row w; w.r = r; w.ok = r < rows; w.p = p + (w.ok ? (long)r * ld : 0L); return w; }
    __device__ __forceinline__ void row_advance(Row& w, int d) const { w.r += d; w.ok = w.r < rows; w.p = p + (w.ok ? (long)w.r * ld : 0L); }
    __device__ __forceinline__ Col col(int c) const { return c; }
    __device__ __forceinline__ void col_advance(Col& c, int d) const { c += d; }
    template <bool V> __device__ __forceinline__ float4 load(const Row& w, const Col& c, bool en, unsigned& m) const {
        const bool ok = en && w.ok && c < cols;
        if (V || vec) { m = ok ? 15u : 0u; return *reinterpret_cast<const float4*>(w.p + (ok ? c : 0)); }
        const bool o1 = ok && c + 1 < cols, o2 = ok && c + 2 < cols, o3 = ok && c + 3 < cols;
        m = (ok ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u);
        return make_float4(w.p[ok ? c : 0], w.p[o1 ? c + 1 : 0], w.p[o2 ? c + 2 : 0], w.p[o3 ? c + 3 : 0]);
    }
};

struct WDgradRow { const float* p; int r, tap, co, ok; };   // row r = (tap, co)

// Conv weight in channels-last physical order W[co][tap][ci] seen as rows = (tap, co), cols = ci
// (the B operand of dgrad).  Batch z = group.
struct WDgradOp {
    static constexpr bool kFast = false;
    const float* w; int taps, Cog, Cig, rows, cols, vec; long gstride;
    typedef WDgradRow Row;
    typedef int Col;
    __device__ __forceinline__ void set_batch(int z) { w += (long)z * gstride; }
    __device__ __forceinline__ void fix(Row& r) const { r.ok = r.r < rows; r.p = w + (r.ok ? ((long)r.co * taps + r.tap) * Cig : 0L); }
    __device__ __forceinline__ Row row(int k) const {
        Row r; r.r = k;
        const int kk = k < rows ? k : 0;
        r.tap = kk / Cog; r.co = kk - r.tap * Cog;
        fix(r);
        return r;
    }
    __device__ __forceinline__ void row_advance(Row& r, int d) const {
        r.r += d; r.co += d;
        while (r.co >= Cog) { r.co -= Cog; ++r.tap; }
        fix(r);
    }
    __device__ __forceinline__ Col col(int c) const { return c; }
    __device__ __forceinline__ void col_advance(Col& c, int d) const { c += d; }
    template <bool V> __device__ __forceinline__ float4 load(const Row& r, const Col& c, bool en, unsigned& m) const {
        const bool ok = en && r.ok && c < cols;
        if (V || vec) { m = ok ? 15u : 0u; return *reinterpret_cast<const float4*>(r.p + (ok ? c : 0)); }
        const bool o1 = ok && c + 1 < cols, o2 = ok && c + 2 < cols, o3 = ok && c + 3 < cols;
        m = (ok ? 1u : 0u) | (o1 ? 2u : 0u) | (o2 ? 4u : 0u) | (o3 ? 8u : 0u);
        return make_float4(r.p[ok ? c : 0], r.p[o1 ? c + 1 : 0], r.p[o2 ? c + 2 : 0], r.p[o3 ? c + 3 : 0]);
    }
};

struct ConvRow { long base; int r, b, y, x, h0, w0, ok; };   // pixel row r = (b, y, x); (h0, w0) = window origin / padded position
struct ConvCol { int c, ci, kh, kw; };                       // column c = ((kh, kw), ci)

__device__ __forceinline__ int cidx(int c) { return c; }
__device__ __forceinline__ int cidx(const ConvCol& k) { return k.c; }

// shared cursor arithmetic of the three im2col views: rows walk over (b, y, x) of an (Hy, Wx) map, columns over (tap, channel)
__device__ __forceinline__ void conv_row_split(ConvRow& w, int r, int rows, int Hy, int Wx) {
    w.r = r; w.ok = r < rows;
    const int rr = w.ok ? r : 0;
    w.x = rr % Wx;
    const int t = rr / Wx;
    w.y = t % Hy; w.b = t / Hy;
}
__device__ __forceinline__ void conv_row_step(ConvRow& w, int d, int rows, int Hy, int Wx) {
    w.r += d; w.ok = w.r < rows; w.x += d;
    while (w.x >= Wx) { w.x -= Wx; if (++w.y >= Hy) { w.y = 0; ++w.b; } }
}
__device__ __forceinline__ ConvCol conv_col_split(int c, int Cg, int ks) {
    ConvCol k; k.c = c;
    const int tap = c / Cg;
    k.ci = c - tap * Cg; k.kh = tap / ks; k.kw = tap - k.kh * ks;
    return k;
}
__device__ __forceinline__ void conv_col_step(ConvCol& k, int d, int Cg, int ks) {
    k.c += d; k.ci += d;
    while (k.ci >= Cg) { k.ci -= Cg; if (++k.kw >= ks) { k.kw = 0; ++k.kh; } }
}

// im2col view of an NHWC tensor X (B, Hi, Wi, Ct): rows = output pixels (b, oh, ow), cols = (tap, ci)
// with ci fastest in [0, Cg).  Batch z = group (channel offset z*Cg).
struct Im2colOp {
    static constexpr bool kFast = false;
    const float* x; int Hi, Wi, Ct, Ho, Wo, ks, stride, pad, Cg, rows, cols, vec, coff;
    typedef ConvRow Row;
    typedef ConvCol Col;
    __device__ __forceinline__ void set_batch(int z) { coff += z * Cg; }
    __device__ __forceinline__ void fix(Row& w) const { w.base = (long)w.b * Hi * Wi; w.h0 = w.y * stride - pad; w.w0 = w.x * stride - pad; }
    __device__ __forceinline__ Row row(int r) const { Row w; conv_row_split(w, r, rows, Ho, Wo); fix(w); return w; }
    __device__ __forceinline__ void row_advance(Row& w, int d) const { conv_row_step(w, d, rows, Ho, Wo); fix(w); }
    __device__ __forceinline__ Col col(int c) const { return conv_col_split(c, Cg, ks); }
    __device__ __forceinline__ void col_advance(Col& k, int d) const { conv_col_step(k, d, Cg, ks); }
    // element offset of (row, col) and its validity, no branches
    __device__ __forceinline__ long off(const Row& w, const Col& k, bool& ok) const {
        const int ih = w.h0 + k.kh, iw = w.w0 + k.kw;
        ok = ok && w.ok && k.c < cols && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
        return ok ? (w.base + (long)ih * Wi + iw) * Ct + coff + k.ci : 0L;
    }
    template <bool V> __device__ __forceinline__ float4 load(const Row& w, const Col& k, bool en, unsigned& m) const {
        if (V || vec) {
            bool ok = en;
            const long o = off(w, k, ok);
            m = ok ? 15u : 0u;
            return *reinterpret_cast<const float4*>(x + o);
        }
        Col k1 = k, k2, k3;
        conv_col_step(k1, 1, Cg, ks); k2 = k1; conv_col_step(k2, 1, Cg, ks); k3 = k2; conv_col_step(k3, 1, Cg, ks);
        bool b0 = en, b1 = en, b2 = en, b3 = en;
        const long o0 = off(w, k, b0), o1 = off(w, k1, b1), o2 = off(w, k2, b2), o3 = off(w, k3, b3);
        m = (b0 ? 1u : 0u) | (b1 ? 2u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u);
        return make_float4(x[o0], x[o1], x[o2], x[o3]);
    }
};

// Transposed gather for conv dgrad: dY NHWC (B, Ho, Wo, Ct); rows = INPUT pixels (b, ih, iw),
// cols = (tap, co), co fastest in [0, Cg).  Element = dY[b, (ih+pad-kh)/s, (iw+pad-kw)/s, co] if divisible.
struct Im2colTOp {
    static constexpr bool kFast = false;
    const float* dy; int Hi, Wi, Ct, Ho, Wo, ks, stride, pad, Cg, rows, cols, vec, coff;
    typedef ConvRow Row;
    typedef ConvCol Col;
    __device__ __forceinline__ void set_batch(int z) { coff += z * Cg; }
    __device__ __forceinline__ void fix(Row& w) const { w.base = (long)w.b * Ho * Wo; w.h0 = w.y + pad; w.w0 = w.x + pad; }
    __device__ __forceinline__ Row row(int r) const { Row w; conv_row_split(w, r, rows, Hi, Wi); fix(w); return w; }
    __device__ __forceinline__ void row_advance(Row& w, int d) const { conv_row_step(w, d, rows, Hi, Wi); fix(w); }
    __device__ __forceinline__ Col col(int c) const { return conv_col_split(c, Cg, ks); }
    __device__ __forceinline__ void col_advance(Col& k, int d) const { conv_col_step(k, d, Cg, ks); }
    __device__ __forceinline__ long off(const Row& w, const Col& k, bool& ok) const {
        const int th = w.h0 - k.kh, tw = w.w0 - k.kw;
        int oh = th, ow = tw;
        bool div = true;
        if (stride == 2) { oh = th >> 1; ow = tw >> 1; div = ((th | tw) & 1) == 0; }                       // uniform branches
        else if (stride != 1) { oh = th / stride; ow = tw / stride; div = oh * stride == th && ow * stride == tw; }
        ok = ok && w.ok && k.c < cols && th >= 0 && tw >= 0 && div && oh < Ho && ow < Wo;
        return ok ? (w.base + (long)oh * Wo + ow) * Ct + coff + k.ci : 0L;
    }
    template <bool V> __device__ __forceinline__ float4 load(const Row& w, const Col& k, bool en, unsigned& m) const {
        if (V || vec) {
            bool ok = en;
            const long o = off(w, k, ok);
            m = ok ? 15u : 0u;
            return *reinterpret_cast<const float4*>(dy + o);
        }
        Col k1 = k, k2, k3;
        conv_col_step(k1, 1, Cg, ks); k2 = k1; conv_col_step(k2, 1, Cg, ks); k3 = k2; conv_col_step(k3, 1, Cg, ks);
        bool b0 = en, b1 = en, b2 = en, b3 = en;
        const long o0 = off(w, k, b0), o1 = off(w, k1, b1), o2 = off(w, k2, b2), o3 = off(w, k3, b3);
        m = (b0 ? 1u : 0u) | (b1 ? 2u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u);
        return make_float4(dy[o0], dy[o1], dy[o2], dy[o3]);
    }
};

// im2col view of NCHW inputs for the two stem convolutions (transfuser.py:136,140): channels
// [0,C0) come from s0 (B,C0,H,W), [C0,C0+C1) from s1 (lidar histogram + target-point image, the
// torch.cat of model.py:741-742 never materialises).  normalize != 0 folds normalize_imagenet
// (transfuser.py:419-428) into the load: ((x / 255) - mean) / std, padding stays 0.
struct Im2colNchwOp {
    static constexpr bool kFast = false;
    const float* s0; const float* s1; int C0, C1, Hi, Wi, Ho, Wo, ks, stride, pad, Cg, rows, cols, vec, normalize;
    float mean[4], stdv[4];
    typedef ConvRow Row;
    typedef ConvCol Col;
    __device__ __forceinline__ void set_batch(int) {}
    __device__ __forceinline__ void fix(Row& w) const { w.base = w.b; w.h0 = w.y * stride - pad; w.w0 = w.x * stride - pad; }
    __device__ __forceinline__ Row row(int r) const { Row w; conv_row_split(w, r, rows, Ho, Wo); fix(w); return w; }
    __device__ __forceinline__ void row_advance(Row& w, int d) const { conv_row_step(w, d, rows, Ho, Wo); fix(w); }
    __device__ __forceinline__ Col col(int c) const { return conv_col_split(c, Cg, ks); }
    __device__ __forceinline__ void col_advance(Col& k, int d) const { conv_col_step(k, d, Cg, ks); }
    __device__ __forceinline__ float at(const Row& w, const Col& k, bool en, bool& ok) const {
        const int ih = w.h0 + k.kh, iw = w.w0 + k.kw;
        ok = en && w.ok && k.c < cols && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
        const int ci = ok ? k.ci : 0;
        const bool first = ci < C0;
        const float* src = (first || !s1) ? s0 : s1;
        const int cs = first ? ci : ci - C0, Cn = first ? C0 : C1;
        const long o = ok ? ((w.base * Cn + cs) * Hi + ih) * Wi + iw : 0L;
        float v = src[o];
        if (normalize) v = ((v / 255.0f) - mean[ci & 3]) / stdv[ci & 3];
        return v;
    }
    template <bool V> __device__ __forceinline__ float4 load(const Row& w, const Col& k, bool en, unsigned& m) const {
        Col k1 = k, k2, k3;
        conv_col_step(k1, 1, Cg, ks); k2 = k1; conv_col_step(k2, 1, Cg, ks); k3 = k2; conv_col_step(k3, 1, Cg, ks);
        bool b0, b1, b2, b3;
        const float v0 = at(w, k, en, b0), v1 = at(w, k1, en, b1), v2 = at(w, k2, en, b2), v3 = at(w, k3, en, b3);
        m = (b0 ? 1u : 0u) | (b1 ? 2u : 0u) | (b2 ? 4u : 0u) | (b3 ? 8u : 0u);
        return make_float4(v0, v1, v2, v3);
    }
};

// ---------------------------------------------------------------- epilogue
struct GemmEpi {
    float* C; long ldc; long ldcj; long sc_outer, sc_inner; int inner;   // element (i, j) at C[i*ldc + j*ldcj]
    const float* bias; long sbias;   // per-column bias; batch z adds z*sbias
    const float* res; long ldres;    // residual with C's batch strides
    float alpha; int relu; int mode; // mode 0: store, 1: +=, 2: atomicAdd
    int group_m = 1;                 // tile rasterisation: rows of tiles walked together (set by launch_cfg)
    const float* mask = nullptr; long ldmask = 0;   // optional ReLU mask of a backward GEMM: element (i, j) is zeroed unless mask(i, j) > 0
    long sk_stride = 0;              // two-pass split-K: k-slice blockIdx.y stores its partial tile at C + blockIdx.y * sk_stride (LDS-DMA kernels)
    float* sk_ws = nullptr; long sk_ws_floats = 0;   // caller-owned scratch that makes the two-pass split-K / stream-K eligible (tf_gemm_desc); read on the device by the stream-K kernels
    int* sk_flags = nullptr;         // stream-K: kStreamKMaxBlocks hand-over flags, zero between launches (caller-owned, persistent; tf_gemm_desc.sk_flags)
    // BatchNorm statistics of the OUTPUT, fused into the epilogue (train-mode BN behind a bias-free conv, timm BatchNormAct2d via
    // transfuser.py:380,442): every wave writes, for each of its columns, the Welford triple (n, mean, M2) of the rows it owns into
    // stat[(part * 3 + {0,1,2}) * stat_ld + column], part = first row / rows per wave - plain stores, no atomics; a finalize kernel
    // merges the parts (Chan's formula).  Only for plain stores (mode 0, no residual / ReLU / mask / split-K).  stat_nparts: HOST pointer,
    // set at launch time to the number of parts this launch writes (0: the chosen plan cannot produce them - caller falls back).
    float* stat = nullptr; long stat_ld = 0; int* stat_nparts = nullptr;
    // nn.Dropout on the product BEFORE the residual is added (x + resid_drop(proj(...)), transfuser.py:543-544): element (i, j) survives iff
    // dropout_keep(*drop_seed, drop_site, i * N + j, drop_thresh) - the mask tf_dropout_add_f32 / tf_dropout_f32 generate for the same (seed, site) over
    // the contiguous (M, N) output, so the backward's tf_dropout_f32 regenerates it.  Batch 1, ldc == N.  (round 5: one launch less per residual branch)
    const uint32_t* drop_seed = nullptr; uint32_t drop_site = 0, drop_thresh = 0; float drop_scale = 1.f;
    int prec = 0;                    // 0: exact fp32 MFMA; 1: operands rounded to bf16 on the LDS->register path, bf16 MFMA, fp32 accumulate; 2: bf16x3 split (fp32-accurate, 6 bf16 MFMAs); 3: as 1 with IEEE-half operands (set by launch_cfg)
    int packed16 = 0;                // LDS-DMA kernels, both operands K-contiguous: the operands ARE 16-bit matrices (1: bf16, 2: IEEE half) described in units of
                                     // 4 bytes (ld, cols, K = halves / 2): tiles are moved as bytes, one ds_read_b128 = one 8-deep MFMA operand (tf_gemm16_nt_f32)
};

int gemm_precision();   // api.cpp: process-wide compute precision of the engine (tf_set_precision)

// ---------------------------------------------------------------- epilogue (shared by gemm_kernel and gemm_dma_kernel)
// lane holds column j of 16 rows per 32x32 tile.  Mode / residual / bounds are resolved ONCE per tile
// (wave-uniform) so the 16 x TM x TN stores of a lane are straight-line code, not a branch + wait per element.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TM][TN], const GemmEpi& ep, int M, int N, int i0, int j0, int BM, int BN,
                                              int wm0, int wn0, int z) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const long cz = (long)(z / ep.inner) * ep.sc_outer + (long)(z % ep.inner) * ep.sc_inner;
    float* C = ep.C + cz;
    const float* res = ep.res ? ep.res + cz : nullptr;
    const float* bias = ep.bias ? ep.bias + (long)z * ep.sbias : nullptr;
    const bool full = (i0 + BM <= M) && (j0 + BN <= N);
    if (ep.prec & 0x100) return;
    const uint32_t dseed = ep.drop_seed ? *ep.drop_seed : 0u;
    auto emit = [&](auto mode_c, auto res_c, auto full_c) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool RES = decltype(res_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int j = j0 + wn0 + u * 32 + l31;
                const bool jok = FULL || j < N;
                const float bj = (bias && jok) ? bias[j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = i0 + wm0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (FULL || (jok && i < M)) {
                        float v = ep.alpha * acc[t][u][r] + bj;
                        if (ep.drop_seed) v = dropout_keep(dseed, ep.drop_site, (uint32_t)((long)i * N + j), ep.drop_thresh) ? v * ep.drop_scale : 0.f;
                        if (RES) v += res[(long)i * ep.ldres + j];
                        v = ep.relu ? fmaxf(v, 0.f) : v;
                        if (ep.mask) v = (ep.mask[(long)i * ep.ldmask + j] > 0.f) ? v : 0.f;
                        float* dst = C + (long)i * ep.ldc + (long)j * ep.ldcj;
                        if (MODE == 0) *dst = v;
                        else if (MODE == 1) *dst += v;
                        else atomicAdd(dst, v);
                    }
                }
            }
    };
    auto by_full = [&](auto mode_c, auto res_c) {
        if (full) emit(mode_c, res_c, std::true_type()); else emit(mode_c, res_c, std::false_type());
    };
    auto by_res = [&](auto mode_c) {
        if (res) by_full(mode_c, std::true_type()); else by_full(mode_c, std::false_type());
    };
    if (ep.mode == 0) by_res(std::integral_constant<int, 0>());
    else if (ep.mode == 1) by_res(std::integral_constant<int, 1>());
    else by_res(std::integral_constant<int, 2>());
    if (ep.stat) {      // wave-uniform: column statistics of the stored values over this wave's 32 * TM rows
        constexpr int WMR = 32 * TM;
        const int r0 = i0 + wm0;
        int nrow = M - r0;
        nrow = nrow < 0 ? 0 : (nrow > WMR ? WMR : nrow);
        if (nrow > 0) {
            const int part = r0 / WMR;
            const float inv_n = 1.0f / (float)nrow;
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int j = j0 + wn0 + u * 32 + l31;
                const bool jok = j < N;
                const float bj = (bias && jok) ? bias[j] : 0.f;
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = r0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        s += (i < M) ? ep.alpha * acc[t][u][r] + bj : 0.f;
                    }
                s += shfl_xor(s, 32);
                const float mean = s * inv_n;
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = r0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const float d = (ep.alpha * acc[t][u][r] + bj) - mean;
                        q += (i < M) ? d * d : 0.f;
                    }
                q += shfl_xor(q, 32);
                if (hi == 0 && jok) {
                    float* st = ep.stat + (long)part * 3 * ep.stat_ld + cz + j;
                    st[0] = (float)nrow; st[ep.stat_ld] = mean; st[2 * ep.stat_ld] = q;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- kernel
template <int BM, int BN, int WAVES_M, int BK, class LA, bool A_KC, class LB, bool B_KC, bool ALLVEC, int NT = 256, int PF = 1>
__device__ __forceinline__ void gemm_tile(LA& la, LB& lb, const GemmEpi& ep, int M, int N, int K, int tiles_m, int tiles_n, int kchunk,
                                          float (*As)[BK][BM + GEMM_PAD], float (*Bs)[BK][BN + GEMM_PAD], int blk_x, int blk_y, int blk_z) {
    constexpr int WAVES_N = (NT / 64) / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1, "wave tile must be 32x32 multiples");
    constexpr int KQ = BK / 4;                                   // float4 per row of a KC operand tile
    constexpr int NLA = (BM * KQ + NT - 1) / NT, NLB = (BN * KQ + NT - 1) / NT;   // float4 slots per thread

    const int tid = threadIdx.x;
    const int z = blk_z;      // (blk_x, blk_y, blk_z): the block's coordinates in ITS problem's grid (= blockIdx in gemm_kernel; decoded from a shared grid in gemm_pair_kernel)
    la.set_batch(z);
    lb.set_batch(z);

    // XCD-aware, bijective block -> tile map: the 8 XCDs (private L2 each) get contiguous tile
    // ranges; inside a range tn is fastest so neighbouring tiles share the A row panel.
    int tile;
    {
        const int nt = tiles_m * tiles_n, bid = blk_x;
        const int q = nt >> 3, r = nt & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // grouped rasterisation: inside an XCD's range the tiles of group_m consecutive tile-rows are walked column by column, so the
    // ~128 tiles resident on the XCD reuse every B panel group_m times while their group_m A panels stay in the 4 MB L2 (row-major
    // order re-streams the whole B operand once per tile-row: 1.1 GB of fabric reads for the 47 MB GPT-4 mlp.0 operands, PMC FETCH_SIZE)
    int tm, tn;
    if (ep.group_m > 1) {
        const int per = ep.group_m * tiles_n, sr = tile / per, rem = tile - sr * per;
        int gsz = tiles_m - sr * ep.group_m;
        gsz = gsz < ep.group_m ? gsz : ep.group_m;
        tn = rem / gsz; tm = sr * ep.group_m + (rem - tn * gsz);
    } else { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
    const int i0 = tm * BM, j0 = tn * BN;
    const int kbeg = blk_y * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int nkt = (kend - kbeg + BK - 1) / BK;

    // masked (general) path: per-slot cursors, see CURSORS above.  KC slot: row fixed, column cursor walks k; IC slot: column fixed,
    // row cursor walks k.  fetch() loads the tile at the cursors and steps them by BK, i.e. tiles are fetched in order from gen_seek().
    typename LA::Row arow[NLA];
    typename LB::Row brow[NLB];
    typename LA::Col acol[NLA];
    typename LB::Col bcol[NLB];
    auto gen_seek = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const int f = tid + p * NT;
            if (A_KC) { arow[p] = la.row(i0 + f / KQ); acol[p] = la.col(k0 + (f % KQ) * 4); }
            else { const int kr = f / (BM / 4), cq = f - kr * (BM / 4); arow[p] = la.row(k0 + kr); acol[p] = la.col(i0 + cq * 4); }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const int f = tid + p * NT;
            if (B_KC) { brow[p] = lb.row(j0 + f / KQ); bcol[p] = lb.col(k0 + (f % KQ) * 4); }
            else { const int kr = f / (BN / 4), cq = f - kr * (BN / 4); brow[p] = lb.row(k0 + kr); bcol[p] = lb.col(j0 + cq * 4); }
        }
    };

    float4 ra[NLA], rb[NLB];
    unsigned ma[NLA], mb[NLB];   // validity bits of the prefetched slots
    auto fetch = [&]() {
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const int f = tid + p * NT;
            if (A_KC) {
                ra[p] = la.template load<ALLVEC>(arow[p], acol[p], ((BM * KQ) % NT == 0 || f < BM * KQ) && cidx(acol[p]) < kend, ma[p]);
                la.col_advance(acol[p], BK);
            } else {
                const int kr = f / (BM / 4);
                ra[p] = la.template load<ALLVEC>(arow[p], acol[p], (((BM / 4) * BK) % NT == 0 || kr < BK) && arow[p].r < kend, ma[p]);
                la.row_advance(arow[p], BK);
            }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const int f = tid + p * NT;
            if (B_KC) {
                rb[p] = lb.template load<ALLVEC>(brow[p], bcol[p], ((BN * KQ) % NT == 0 || f < BN * KQ) && cidx(bcol[p]) < kend, mb[p]);
                lb.col_advance(bcol[p], BK);
            } else {
                const int kr = f / (BN / 4);
                rb[p] = lb.template load<ALLVEC>(brow[p], bcol[p], (((BN / 4) * BK) % NT == 0 || kr < BK) && brow[p].r < kend, mb[p]);
                lb.row_advance(brow[p], BK);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const int f = tid + p * NT;
            if (A_KC) {
                if (f < BM * KQ) {
                    const int r = f / KQ, kq = (f % KQ) * 4;
                    As[buf][kq + 0][r] = (ma[p] & 1u) ? ra[p].x : 0.f; As[buf][kq + 1][r] = (ma[p] & 2u) ? ra[p].y : 0.f;
                    As[buf][kq + 2][r] = (ma[p] & 4u) ? ra[p].z : 0.f; As[buf][kq + 3][r] = (ma[p] & 8u) ? ra[p].w : 0.f;
                }
            } else {
                const int kr = f / (BM / 4), cq = f - kr * (BM / 4);
                if (kr < BK) *reinterpret_cast<float4*>(&As[buf][kr][cq * 4]) = make_float4((ma[p] & 1u) ? ra[p].x : 0.f, (ma[p] & 2u) ? ra[p].y : 0.f, (ma[p] & 4u) ? ra[p].z : 0.f, (ma[p] & 8u) ? ra[p].w : 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const int f = tid + p * NT;
            if (B_KC) {
                if (f < BN * KQ) {
                    const int r = f / KQ, kq = (f % KQ) * 4;
                    Bs[buf][kq + 0][r] = (mb[p] & 1u) ? rb[p].x : 0.f; Bs[buf][kq + 1][r] = (mb[p] & 2u) ? rb[p].y : 0.f;
                    Bs[buf][kq + 2][r] = (mb[p] & 4u) ? rb[p].z : 0.f; Bs[buf][kq + 3][r] = (mb[p] & 8u) ? rb[p].w : 0.f;
                }
            } else {
                const int kr = f / (BN / 4), cq = f - kr * (BN / 4);
                if (kr < BK) *reinterpret_cast<float4*>(&Bs[buf][kr][cq * 4]) = make_float4((mb[p] & 1u) ? rb[p].x : 0.f, (mb[p] & 2u) ? rb[p].y : 0.f, (mb[p] & 4u) ? rb[p].z : 0.f, (mb[p] & 8u) ? rb[p].w : 0.f);
            }
        }
    };

    // ---- fast path (plain operands, 16-byte loads): rows / columns beyond the matrix are CLAMPED to valid addresses instead of
    // masked - the products they feed land in output rows / columns the epilogue discards - so full k tiles need no validity
    // bits, no selects and no per-step address arithmetic (one pointer bump per slot).  Only a ragged last k tile takes the
    // masked path.  Measured on the MI355X (tools/probe/gemm_lab.cpp): 64x64 tiles 95 -> 108 TFLOP/s, 128x128 82 -> 94.
    constexpr bool CANFAST = ALLVEC && LA::kFast && LB::kFast;
    const float* fa[NLA];
    const float* fb[NLB];
    long fa_step = 0, fb_step = 0;
    if constexpr (CANFAST) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const int f = tid + p * NT;
            if (A_KC) {
                const int fr = (f < BM * KQ) ? f : 0;
                int r = i0 + fr / KQ; r = r < la.rows ? r : la.rows - 1;
                fa[p] = la.p + (long)r * la.ld + kbeg + (fr % KQ) * 4;
            } else {
                int kr = f / (BM / 4); const int cq = f - kr * (BM / 4);
                kr = kr < BK ? kr : 0;
                int c = i0 + cq * 4; c = c < la.cols ? c : 0;
                fa[p] = la.p + (long)(kbeg + kr) * la.ld + c;
            }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const int f = tid + p * NT;
            if (B_KC) {
                const int fr = (f < BN * KQ) ? f : 0;
                int r = j0 + fr / KQ; r = r < lb.rows ? r : lb.rows - 1;
                fb[p] = lb.p + (long)r * lb.ld + kbeg + (fr % KQ) * 4;
            } else {
                int kr = f / (BN / 4); const int cq = f - kr * (BN / 4);
                kr = kr < BK ? kr : 0;
                int c = j0 + cq * 4; c = c < lb.cols ? c : 0;
                fb[p] = lb.p + (long)(kbeg + kr) * lb.ld + c;
            }
        }
        fa_step = A_KC ? BK : (long)BK * la.ld;
        fb_step = B_KC ? BK : (long)BK * lb.ld;
    }
    auto fetch_set = [&](float4* qa, float4* qb, int kt) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) qa[p] = *reinterpret_cast<const float4*>(fa[p] + (long)kt * fa_step);
#pragma unroll
        for (int p = 0; p < NLB; ++p) qb[p] = *reinterpret_cast<const float4*>(fb[p] + (long)kt * fb_step);
    };
    auto stash_set = [&](const float4* qa, const float4* qb, int buf) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const int f = tid + p * NT;
            if (A_KC) {
                if ((BM * KQ) % NT == 0 || f < BM * KQ) {
                    const int r = f / KQ, kq = (f % KQ) * 4;
                    As[buf][kq + 0][r] = qa[p].x; As[buf][kq + 1][r] = qa[p].y; As[buf][kq + 2][r] = qa[p].z; As[buf][kq + 3][r] = qa[p].w;
                }
            } else {
                const int kr = f / (BM / 4), cq = f - kr * (BM / 4);
                if (((BM / 4) * BK) % NT == 0 || kr < BK) *reinterpret_cast<float4*>(&As[buf][kr][cq * 4]) = qa[p];
            }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const int f = tid + p * NT;
            if (B_KC) {
                if ((BN * KQ) % NT == 0 || f < BN * KQ) {
                    const int r = f / KQ, kq = (f % KQ) * 4;
                    Bs[buf][kq + 0][r] = qb[p].x; Bs[buf][kq + 1][r] = qb[p].y; Bs[buf][kq + 2][r] = qb[p].z; Bs[buf][kq + 3][r] = qb[p].w;
                }
            } else {
                const int kr = f / (BN / 4), cq = f - kr * (BN / 4);
                if (((BN / 4) * BK) % NT == 0 || kr < BK) *reinterpret_cast<float4*>(&Bs[buf][kr][cq * 4]) = qb[p];
            }
        }
    };
    auto fetch_fast = [&](int kt) { fetch_set(ra, rb, kt); };
    auto stash_fast = [&](int buf) { stash_set(ra, rb, buf); };
    const int nfast = CANFAST ? (kend - kbeg) / BK : 0;    // leading k tiles that are complete (block-uniform)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;

    // software-pipelined operand fetch: the LDS reads of step kk+1 are issued BEFORE the MFMAs of step kk (hipcc otherwise emits
    // read -> s_waitcnt lgkmcnt(0) -> MFMAs per step, exposing the LDS latency whenever a SIMD holds fewer than ~3 waves)
    auto compute = [&](int cur) {
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[0][t] = As[cur][hi][wm0 + t * 32 + l31];
#pragma unroll
        for (int t = 0; t < TN; ++t) b[0][t] = Bs[cur][hi][wn0 + t * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int s = kk & 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int t = 0; t < TM; ++t) a[s ^ 1][t] = As[cur][kk * 2 + 2 + hi][wm0 + t * 32 + l31];
#pragma unroll
                for (int t = 0; t < TN; ++t) b[s ^ 1][t] = Bs[cur][kk * 2 + 2 + hi][wn0 + t * 32 + l31];
            }
            TF_SCHED_FENCE();
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) mfma_32x32x2(a[s][t], b[s][u], acc[t][u]);
            TF_SCHED_FENCE();
        }
    };

    // bf16-MFMA variant of compute(): one v_mfma_f32_32x32x16_bf16 per 32x32 tile and 16 k (lane half hi owns k = 8 hi .. 8 hi + 7 of
    // the group); operands are read as fp32 from the same K-major tiles and rounded in registers.  Kept in the same kernel behind a
    // block-uniform flag: its extra live registers exist only inside this branch.
    auto compute_bf16 = [&](int cur) {
#pragma unroll
        for (int g = 0; g < BK / 16; ++g) {
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[t][j] = As[cur][g * 16 + 8 * hi + j][wm0 + t * 32 + l31];
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) b[t][j] = Bs[cur][g * 16 + 8 * hi + j][wn0 + t * 32 + l31];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) mfma_32x32x16_lp(a[t], b[u], acc[t][u], ep.prec);
        }
    };
    const bool lowp = ep.prec == 1 || ep.prec == 3;      // precision 2 (bf16x3 split) exists in the LDS-DMA kernels only: this kernel then stays on the exact fp32 MFMA
    auto mult = [&](int cur) { if (lowp) compute_bf16(cur); else compute(cur); };

    if constexpr (CANFAST) {
        // hot loop: complete k tiles only, nothing but pointer bumps, 16-byte loads, MFMAs and one barrier per tile; a ragged last
        // tile is peeled off below (keeping the masked code out of the loop also keeps its registers out of the loop's allocation)
        if constexpr (PF == 2) {
            // prefetch distance 2: tile kt+2 is requested while tile kt is multiplied (two register sets A = ra/rb, B = ra2/rb2; the
            // loop is unrolled by 2 so the set roles are static); measured +5-10 % on 128x128 tiles with 8 waves (tools/probe/gemm_lab.cpp)
            float4 ra2[NLA], rb2[NLB];
            const int last = nfast - 1;
            if (nfast > 0) {
                fetch_set(ra, rb, 0);
                stash_set(ra, rb, 0);
                fetch_set(ra, rb, 1 < nfast ? 1 : last);                   // A <- tile 1
            }
            __syncthreads();
            for (int kt = 0; kt < nfast; kt += 2) {                        // LDS 0 holds tile kt, set A holds tile kt+1
                fetch_set(ra2, rb2, kt + 2 < nfast ? kt + 2 : last);       // B <- tile kt+2
                mult(0);
                stash_set(ra, rb, 1);                                      // A (tile kt+1) -> LDS 1
                __syncthreads();
                if (kt + 1 < nfast) {
                    fetch_set(ra, rb, kt + 3 < nfast ? kt + 3 : last);     // A <- tile kt+3
                    mult(1);
                    stash_set(ra2, rb2, 0);                                // B (tile kt+2) -> LDS 0
                    __syncthreads();
                }
            }
        } else {
        if (nfast > 0) {
            fetch_fast(0);
            stash_fast(0);
        }
        __syncthreads();
        for (int kt = 0; kt < nfast; ++kt) {     // branch-free body: the last iteration re-fetches its own tile into the idle buffer
            const int cur = kt & 1;
            fetch_fast(kt + 1 < nfast ? kt + 1 : kt);
            mult(cur);
            stash_fast(cur ^ 1);
            __syncthreads();
        }
        }
        if (nkt > nfast) {
            gen_seek(kbeg + nfast * BK);
            fetch();
            stash(nfast & 1);
            __syncthreads();
            mult(nfast & 1);
        }
    } else {
        gen_seek(kbeg);
        if (nkt > 0) {
            fetch();
            stash(0);
        }
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {       // branch-free body: past the last tile every validity bit is 0 (zeros into the idle buffer)
            const int cur = kt & 1;
            fetch();
            mult(cur);
            stash(cur ^ 1);
            __syncthreads();
        }
    }

    gemm_epilogue<TM, TN>(acc, ep, M, N, i0, j0, BM, BN, wm0, wn0, z);
}

// ALLVEC = both operands may be read with 16-byte loads (decided on the host): separate instantiation so the common
// vector kernel does not inherit the register pressure of the element-wise (attention head / stem) path.
// __launch_bounds__ 2nd argument = waves per SIMD the register allocator must leave room for (= resident 256-thread
// blocks per CU): PMC showed the big tiles lose more to the tail round (tiles / resident slots) than to anything in the
// K loop, so they are held to 3 blocks per CU (<= 168 registers incl. 64 accumulators) and the small ones to 4+.
template <int BM, int BN, int WAVES_M, int BK, class LA, bool A_KC, class LB, bool B_KC, bool ALLVEC, int NT = 256, int PF = 1>
__global__ void __launch_bounds__(NT, NT == 512 ? 2 : ((BM * BN >= 128 * 96) ? 3 : 4)) gemm_kernel(LA la, LB lb, GemmEpi ep, int M, int N, int K, int tiles_m, int tiles_n, int kchunk) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + GEMM_PAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + GEMM_PAD];
    gemm_tile<BM, BN, WAVES_M, BK, LA, A_KC, LB, B_KC, ALLVEC, NT, PF>(la, lb, ep, M, N, K, tiles_m, tiles_n, kchunk, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------- pair launch
// TWO independent plain GEMMs in ONE grid (the weight gradient and the input gradient of a layer both read dy and write disjoint outputs):
// blocks [0, n1) run problem 1's tiles, blocks [n1pad, n1pad + n2) problem 2's (n1pad = n1 rounded up to the 8 XCDs; the blocks between exit).
// The second problem's workgroups start as the first one's retire, so the drain of one launch, the dispatch gap and the ramp of the next -
// 4-9 us per 64 x 64-tile launch on this part (profiles/r04_launch_lab.txt) - are filled with MFMA work, without the cross-queue edge a side
// stream costs (~10 us each, DESIGN section 3 "round 4").  64 x 64 tiles, 4 waves, both problems with 16-byte operand loads.
struct PairSide { PlainOp la, lb; GemmEpi ep; int M, N, K, tiles_m, tiles_n, kchunk, gx, gy; };

template <int BK1, bool A1_KC, bool B1_KC, int BK2, bool A2_KC, bool B2_KC>
__global__ void __launch_bounds__(256, 4) gemm_pair_kernel(PairSide s1, PairSide s2, int n1, int n1pad) {
    constexpr int BKM = BK1 > BK2 ? BK1 : BK2;
    __shared__ __attribute__((aligned(16))) float As[2 * BKM * (64 + GEMM_PAD)];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BKM * (64 + GEMM_PAD)];
    int bid = blockIdx.x;
    if (bid < n1) {
        const int by = bid / s1.gx, bx = bid - by * s1.gx;
        gemm_tile<64, 64, 2, BK1, PlainOp, A1_KC, PlainOp, B1_KC, true>(s1.la, s1.lb, s1.ep, s1.M, s1.N, s1.K, s1.tiles_m, s1.tiles_n, s1.kchunk,
                                                                            reinterpret_cast<float (*)[BK1][64 + GEMM_PAD]>(As),
                                                                            reinterpret_cast<float (*)[BK1][64 + GEMM_PAD]>(Bs), bx, by, 0);
    } else if (bid >= n1pad) {
        bid -= n1pad;
        const int by = bid / s2.gx, bx = bid - by * s2.gx;
        gemm_tile<64, 64, 2, BK2, PlainOp, A2_KC, PlainOp, B2_KC, true>(s2.la, s2.lb, s2.ep, s2.M, s2.N, s2.K, s2.tiles_m, s2.tiles_n, s2.kchunk,
                                                                            reinterpret_cast<float (*)[BK2][64 + GEMM_PAD]>(As),
                                                                            reinterpret_cast<float (*)[BK2][64 + GEMM_PAD]>(Bs), bx, by, 0);
    }
}

// ---------------------------------------------------------------- host dispatch
// kind 0: gemm_kernel (register-staged, any loader).  kind >= 1: LDS-DMA kernel configuration of tf_gemm_dma.h (plain vector operands
// only; bm/bn/bk then mirror that configuration's tile for the plan file's readers).
struct GemmPlan { int bm, bn, bk, splitk; int kind = 0; };

// LDS-DMA configurations (tf_gemm_dma.h); the launchers live in gemm_dma_{nt,nn,tn,tt}.cpp (one translation unit per operand layout)
constexpr int kDmaKinds = 8;
struct DmaKindInfo { int bm, bn, bk, nw, occ, lds; };        // tile, waves per workgroup, __launch_bounds__ waves per SIMD, LDS bytes (mirror of tf_gemm_dma_launch.h)
inline DmaKindInfo dma_kind_info(int kind) {
    static const DmaKindInfo t[kDmaKinds + 1] = {{0, 0, 0, 0, 0, 0}, {128, 128, 16, 4, 2, 49152}, {64, 64, 16, 4, 4, 32768}, {128, 64, 16, 4, 3, 36864}, {64, 128, 16, 4, 3, 36864},
                                                 {128, 128, 32, 4, 2, 65536}, {64, 64, 16, 1, 2, 32768}, {64, 128, 16, 2, 2, 36864}, {128, 64, 16, 2, 2, 36864}};   // 6-8: one 64x64 accumulator block per wave (1 / 2 / 2 waves)
    return t[(kind >= 1 && kind <= kDmaKinds) ? kind : 0];
}
template <bool A_KC, bool B_KC>
void launch_dma_plan(int kind, const PlainOp& la, const PlainOp& lb, const GemmEpi& ep, int M, int N, int K, int batch, int splitk, void* stream);
bool dma_eligible(const PlainOp& a, const PlainOp& b);

// two-pass split-K (gemm_fixup.cpp): C (op)= epilogue(sum_s ws[s]) with the slices summed in order; ws slices are [M][ldws] row-major
void launch_splitk_fixup(const float* ws, int nsplit, long sk_stride, int ldws, const GemmEpi& ep, int M, int N, void* stream);
constexpr int kStreamK = 2000000;  // GemmPlan.splitk == kStreamK: stream-K (persistent workgroups over the (tile, k-tile) space; LDS-DMA kinds only, tf_gemm_dma.h)
constexpr int kStreamKMaxBlocks = 2048;
int device_cus();                  // api.cpp: compute units of the current device (cached)
long streamk_count(int add);       // api.cpp: stream-K launches so far (tests assert that a pinned stream-K plan really ran as one)
// workgroups of a stream-K launch of this configuration, 0 = not eligible: single-batch, non-atomic epilogue, caller scratch + flags, at least one
// whole tile of work per workgroup (so a tile is cut at most once) and a tile count that does NOT already divide evenly over the resident slots
inline int streamk_blocks(const GemmEpi& ep, int M, int N, int K, int batch, int bm, int bn, int bk, int nw, int occ, int lds_bytes) {
    if (!ep.sk_ws || !ep.sk_flags || batch != 1 || ep.mode == 2 || K < 8 * bk) return 0;
    int per_cu = occ * 4 / nw;                                 // __launch_bounds__(64 nw, occ): occ waves per SIMD
    const int by_lds = (160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1);
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    const long nt = (long)cdiv(M, bm) * cdiv(N, bn);
    long P = (long)device_cus() * per_cu;
    if (P > kStreamKMaxBlocks) P = kStreamKMaxBlocks;
    if (P > nt) P = nt;
    P &= ~7L;                                                  // XCD-contiguous positions
    if (P < 8 || nt % P == 0) return 0;                        // already balanced: the data-parallel launch is the same thing without the bookkeeping
    if (P * (long)bm * bn > ep.sk_ws_floats) return 0;
    return (int)P;
}
constexpr int kTwoPass = 1000000;  // GemmPlan.splitk >= kTwoPass: two-pass split-K with S = splitk - kTwoPass slices (LDS-DMA kinds only); atomic split counts stay far below
inline int twopass_ldws(int N) { return (N + 3) & ~3; }
// eligibility of the two-pass split for this call: plain row-major single-batch output and enough caller scratch for S slices
inline bool twopass_ok(const GemmEpi& ep, int M, int N, int batch, int S) {
    return ep.sk_ws && batch == 1 && ep.ldcj == 1 && (ep.mode == 0 || ep.mode == 1) && S >= 2 && S <= 4 && (long)S * M * twopass_ldws(N) <= ep.sk_ws_floats;
}

// pair launch (gemm_pair.cpp): between tf_gemm_pair_begin() and tf_gemm_pair_end() up to two eligible GEMMs are held back and launched as ONE grid
bool pair_capturing();
bool pair_hold(const PlainOp& la, const PlainOp& lb, const GemmEpi& ep, int M, int N, int K, bool a_kc, bool b_kc, const GemmPlan& p, const char* what);

// plan cache + autotuner state (api.cpp)
bool plan_lookup(const char* what, int M, int N, int K, int batch, int acc, GemmPlan* out);
void plan_store(const char* what, int M, int N, int K, int batch, int acc, const GemmPlan& p);
bool autotune_enabled();
bool forced_plan(GemmPlan* out);   // tests: pin one tiling for every call

inline int heuristic_splitk(int M, int N, int K, int batch, int bm, int bn, int bk) {
    const long tiles = (long)cdiv(M, bm) * cdiv(N, bn) * batch;
    const int ktiles = cdiv(K, bk);
    if (tiles >= 512 || ktiles * bk < 256) return 1;
    long want = (1024 + tiles - 1) / tiles;
    long maxs = (long)ktiles * bk / 128;
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : (int)want;
}

// Heuristic tile choice (used when no tuned plan exists): smallest BN that covers N with the least padding,
// BM=64 when the grid would not fill the 256 CUs, split-K (atomic epilogue) for skinny outputs with deep reductions.
inline GemmPlan plan_gemm(int M, int N, int K, int batch, bool allow_splitk) {
    GemmPlan p;
    if (N <= 32) p.bn = 32;
    else if (N <= 64) p.bn = 64;
    else if (N <= 96) p.bn = 96;
    else {
        const long w128 = (long)cdiv(N, 128) * 128, w96 = (long)cdiv(N, 96) * 96;
        p.bn = (w96 < w128) ? 96 : 128;
    }
    p.bm = 128;
    p.bk = 16;
    const long t128 = (long)cdiv(M, 128) * cdiv(N, p.bn) * batch;
    if ((p.bn == 128 || p.bn == 64) && t128 < 384 && M > 64) p.bm = 64;
    p.splitk = allow_splitk ? heuristic_splitk(M, N, K, batch, p.bm, p.bn, p.bk) : 1;
    return p;
}

// grid geometry + launch-time epilogue fields (compute precision, tile rasterisation, number of statistic parts) of one register-staged launch
struct CfgGeom { int tiles_m, tiles_n, kchunk, nsplit; GemmEpi epg; };
inline CfgGeom cfg_geom(const GemmEpi& ep, int M, int N, int K, int splitk, int BM, int BN, int BK, int WAVES_M) {
    CfgGeom c;
    c.tiles_m = cdiv(M, BM); c.tiles_n = cdiv(N, BN);
    int kchunk = cdiv(cdiv(K, splitk), BK) * BK;
    if (kchunk < BK) kchunk = BK;
    c.kchunk = kchunk;
    const int nsplit = cdiv(K, kchunk);
    c.nsplit = nsplit > 0 ? nsplit : 1;
    c.epg = ep;
    c.epg.prec = gemm_precision();
    { static const int dbg = [] { const char* e = getenv("TF_GEMM_DBG"); return e ? atoi(e) : 0; }(); c.epg.prec |= dbg << 8; }      // timing diagnosis (tools/pair_lab.py): bit 8 = no epilogue stores, results are garbage
    {
        static const int forced = [] { const char* e = getenv("TF_GROUP_M"); return e ? atoi(e) : 0; }();
        long panel = (long)BM * (kchunk < K ? kchunk : K) * 4;          // bytes of one A panel of this launch
        int g = (int)((2L << 20) / (panel > 0 ? panel : 1));            // as many tile-rows as keep their A panels in ~half the L2
        if (g > 8) g = 8;
        if (forced > 0) g = forced;
        if (g > c.tiles_m) g = c.tiles_m;
        c.epg.group_m = (g >= 2 && c.tiles_n >= 4) ? g : 1;
    }
    if (c.epg.stat_nparts) *c.epg.stat_nparts = c.epg.stat ? cdiv(M, BM / WAVES_M) : 0;
    return c;
}

template <int BM, int BN, int WAVES_M, int BK, class LA, bool A_KC, class LB, bool B_KC, int NT = 256, int PF = 1>
inline void launch_cfg(const LA& la, const LB& lb, const GemmEpi& ep, int M, int N, int K, int batch, int splitk, void* stream) {
    const CfgGeom cg = cfg_geom(ep, M, N, K, splitk, BM, BN, BK, WAVES_M);
    const int tiles_m = cg.tiles_m, tiles_n = cg.tiles_n, kchunk = cg.kchunk;
    dim3 grid(tiles_m * tiles_n, cg.nsplit, batch);
    const GemmEpi& epg = cg.epg;
    if (la.vec && lb.vec)
        TF_LAUNCH((gemm_kernel<BM, BN, WAVES_M, BK, LA, A_KC, LB, B_KC, true, NT, PF>), grid, dim3(NT), stream, la, lb, epg, M, N, K, tiles_m, tiles_n, kchunk);
    else
        TF_LAUNCH((gemm_kernel<BM, BN, WAVES_M, BK, LA, A_KC, LB, B_KC, false, NT, PF>), grid, dim3(NT), stream, la, lb, epg, M, N, K, tiles_m, tiles_n, kchunk);
}

template <class LA, bool A_KC, class LB, bool B_KC>
inline void launch_plan(const GemmPlan& p, const LA& la, const LB& lb, const GemmEpi& ep, int M, int N, int K, int batch, void* stream) {
    if constexpr (std::is_same<LA, PlainOp>::value && std::is_same<LB, PlainOp>::value) {
        if (p.kind >= 1 && p.kind <= kDmaKinds && dma_eligible(la, lb)) {
            launch_dma_plan<A_KC, B_KC>(p.kind, la, lb, ep, M, N, K, batch, p.splitk, stream);
            return;
        }
    }
    const int sk = p.splitk >= kTwoPass ? 1 : p.splitk;       // the two-pass split exists in the LDS-DMA kernels only
#define TF_CFG(BM_, BN_, WM_)                                                                                      \
    do {                                                                                                           \
        if (p.bk == 32) launch_cfg<BM_, BN_, WM_, 32, LA, A_KC, LB, B_KC>(la, lb, ep, M, N, K, batch, sk, stream); \
        else launch_cfg<BM_, BN_, WM_, 16, LA, A_KC, LB, B_KC>(la, lb, ep, M, N, K, batch, sk, stream);       \
    } while (0)
    if (p.bm == 128) {
        if (p.bn == 32) TF_CFG(128, 32, 4);
        else if (p.bn == 64) TF_CFG(128, 64, 2);
        else if (p.bn == 96) TF_CFG(128, 96, 4);
        else TF_CFG(128, 128, 2);   // (an 8-wave / prefetch-distance-2 variant - gemm_tile<..., NT = 512, PF = 2> - reached 102-111 TF/s in
                                    //  tools/probe/gemm_lab.cpp but only 88-93 TF/s inside the engine: not dispatched)
    } else {
        if (p.bn == 64) TF_CFG(64, 64, 2);
        else TF_CFG(64, 128, 1);
    }
#undef TF_CFG
}

#ifndef TF_EMU
// TF_TRACE_TUNE=1: name every autotune candidate before it is launched and wait for it (debugging aid: a GPU fault then names its plan)
inline bool trace_tune() { static const bool on = [] { const char* e = getenv("TF_TRACE_TUNE"); return e && atoi(e) != 0; }(); return on; }
inline void trace_candidate(const GemmPlan& p, int M, int N, int K, int batch, int mode, void* stream, bool before) {
    if (!trace_tune()) return;
    if (before) fprintf(stderr, "[tune] %dx%dx%d batch %d: tile %dx%dx%d splitk %d kind %d mode %d ...", M, N, K, batch, p.bm, p.bn, p.bk, p.splitk, p.kind, mode);
    else { hipError_t e = hipStreamSynchronize((hipStream_t)stream); fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); }
    fflush(stderr);
}
// cudnn.benchmark-style tuning (the reference enables it, train.py:115): time every candidate tiling of this exact
// problem ONCE with HIP events during eager warm-up; the winner goes into the plan cache.  Trial launches of
// accumulating epilogues write into a scratch copy of the destination extent, so the live accumulator is left unchanged.
template <class LA, bool A_KC, class LB, bool B_KC>
inline GemmPlan autotune_gemm(const LA& la, const LB& lb, const GemmEpi& ep, int M, int N, int K, int batch, bool allow_splitk, void* stream) {
    static const int tiles[6][2] = {{128, 128}, {128, 96}, {128, 64}, {128, 32}, {64, 128}, {64, 64}};
    GemmEpi trial = ep;
    float* trial_scratch = nullptr;
    if (ep.mode != 0) {
        // accumulating epilogues (weight gradients into the gradient arena): the trials write to a SCRATCH copy of the destination extent, never
        // to the live accumulator (a trial that multiplied the accumulator by alpha = 0 would turn an Inf / NaN already in it into NaN for good)
        const long zo = (long)((batch - 1) / ep.inner), zi = (long)((batch - 1) % ep.inner);
        const long extent = zo * (ep.sc_outer > 0 ? ep.sc_outer : 0) + zi * (ep.sc_inner > 0 ? ep.sc_inner : 0) + (long)(M - 1) * ep.ldc + (long)(N - 1) * ep.ldcj + 1;
        if (ep.sc_outer >= 0 && ep.sc_inner >= 0 && hipMalloc((void**)&trial_scratch, (size_t)extent * sizeof(float)) == hipSuccess) {
            hipMemsetAsync(trial_scratch, 0, (size_t)extent * sizeof(float), (hipStream_t)stream);
            trial.C = trial_scratch;
        } else {                                     // cannot allocate: fall back to the untimed heuristic plan, the accumulator stays untouched
            (void)hipGetLastError();
            return plan_gemm(M, N, K, batch, allow_splitk);
        }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    GemmPlan best = plan_gemm(M, N, K, batch, allow_splitk);
    float best_ms = 1e30f;
    const int npad = cdiv(N, 32) * 32;
    for (int t = 0; t < 6; ++t) {
        const int bm = tiles[t][0], bn = tiles[t][1];
        if (bn > 32 && bn >= npad + 32) continue;          // a whole extra 32-column strip of padding
        if (bm == 64 && M <= 64 && t != 4 && t != 5) continue;
        for (int bk = 16; bk <= 32; bk += 16) {
            int sks[3] = {1, 0, 0}, nsk = 1;
            if (allow_splitk) {
                const int h = heuristic_splitk(M, N, K, batch, bm, bn, bk);
                if (h > 1) { sks[nsk++] = h; if (h >= 4) sks[nsk++] = h / 2; }
            }
            for (int s = 0; s < nsk; ++s) {
                GemmPlan p{bm, bn, bk, sks[s], 0};
                GemmEpi e = trial;
                if (p.splitk > 1) { if (ep.mode == 0) continue; e.mode = 2; }
                trace_candidate(p, M, N, K, batch, e.mode, stream, true);
                launch_plan<LA, A_KC, LB, B_KC>(p, la, lb, e, M, N, K, batch, stream);   // warm
                trace_candidate(p, M, N, K, batch, e.mode, stream, false);
                float ms = 1e30f;
                for (int pass = 0; pass < 3; ++pass) {     // best of three groups of 4 launches: one noisy group must not decide a plan
                    hipEventRecord(e0, (hipStream_t)stream);
                    for (int r = 0; r < 4; ++r) launch_plan<LA, A_KC, LB, B_KC>(p, la, lb, e, M, N, K, batch, stream);
                    hipEventRecord(e1, (hipStream_t)stream);
                    hipEventSynchronize(e1);
                    float t = 0.f;
                    hipEventElapsedTime(&t, e0, e1);
                    if (t < ms) ms = t;
                }
                if (ms < best_ms) { best_ms = ms; best = p; }
            }
        }
    }
    if constexpr (std::is_same<LA, PlainOp>::value && std::is_same<LB, PlainOp>::value) {
        if (dma_eligible(la, lb)) {
            for (int kind = 1; kind <= kDmaKinds; ++kind) {
                const DmaKindInfo ki = dma_kind_info(kind);
                if (ki.bn > 32 && ki.bn >= npad + 32) continue;
                if (ki.bm > 64 && M <= 64) continue;
                int sks[3] = {1, 0, 0}, nsk = 1;
                if (allow_splitk) {
                    const int h = heuristic_splitk(M, N, K, batch, ki.bm, ki.bn, ki.bk);
                    if (h > 1) { sks[nsk++] = h; if (h >= 4) sks[nsk++] = h / 2; }
                }
                int cand[8], nc = 0;
                for (int s = 0; s < nsk; ++s) cand[nc++] = sks[s];
                if ((long)cdiv(M, ki.bm) * cdiv(N, ki.bn) < 512 && K >= 32 * ki.bk)      // too few tiles for the 256 CUs: deterministic two-pass split
                    for (int S = 2; S <= 4; ++S)
                        if (twopass_ok(ep, M, N, batch, S)) cand[nc++] = kTwoPass + S;
                if (streamk_blocks(ep, M, N, K, batch, ki.bm, ki.bn, ki.bk, ki.nw, ki.occ, ki.lds) > 0) cand[nc++] = kStreamK;      // tile count does not divide over the resident slots
                for (int s = 0; s < nc; ++s) {
                    GemmPlan p{ki.bm, ki.bn, ki.bk, cand[s], kind};
                    GemmEpi e = trial;
                    if (p.splitk > 1 && p.splitk < kTwoPass) { if (ep.mode == 0) continue; e.mode = 2; }
                    if (p.splitk == kStreamK && e.mode == 2) continue;
                    trace_candidate(p, M, N, K, batch, e.mode, stream, true);
                    launch_plan<LA, A_KC, LB, B_KC>(p, la, lb, e, M, N, K, batch, stream);
                    trace_candidate(p, M, N, K, batch, e.mode, stream, false);
                    float ms = 1e30f;
                    for (int pass = 0; pass < 3; ++pass) {
                        hipEventRecord(e0, (hipStream_t)stream);
                        for (int r = 0; r < 4; ++r) launch_plan<LA, A_KC, LB, B_KC>(p, la, lb, e, M, N, K, batch, stream);
                        hipEventRecord(e1, (hipStream_t)stream);
                        hipEventSynchronize(e1);
                        float t = 0.f;
                        hipEventElapsedTime(&t, e0, e1);
                        if (t < ms) ms = t;
                    }
                    if (ms < best_ms) { best_ms = ms; best = p; }
                }
            }
        }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (trial_scratch) { hipStreamSynchronize((hipStream_t)stream); hipFree(trial_scratch); }
    return best;
}
#endif

template <class LA, bool A_KC, class LB, bool B_KC>
inline int launch_gemm(const LA& la, const LB& lb, GemmEpi ep, int M, int N, int K, int batch, bool allow_splitk, void* stream,
                       const char* what) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    // split-K only for pure accumulations (weight gradients into the grad arena): atomic epilogue
    const bool sk_ok = allow_splitk && ep.mode == 1 && !ep.bias && !ep.res && !ep.relu;
    const int acc = (ep.mode != 0 ? (sk_ok ? 2 : 1) : 0) + 4 * (gemm_precision() == 3 ? 1 : gemm_precision());   // plans are tuned per compute precision (fp16 shares the bf16 plans)
    GemmPlan p;
    if (forced_plan(&p)) {
        if (!sk_ok && p.splitk < kTwoPass) p.splitk = 1;
    } else if (!plan_lookup(what, M, N, K, batch, acc, &p)) {
#ifndef TF_EMU
        if (autotune_enabled()) {
            p = autotune_gemm<LA, A_KC, LB, B_KC>(la, lb, ep, M, N, K, batch, sk_ok, stream);
            plan_store(what, M, N, K, batch, acc, p);
        } else
#endif
            p = plan_gemm(M, N, K, batch, sk_ok);
    }
    bool streamk = p.splitk == kStreamK;
    if (streamk) {      // stream-K plan: needs an LDS-DMA kind, this call's scratch + flags and a tile count that does not divide over the slots
        bool ok = false;
        if constexpr (std::is_same<LA, PlainOp>::value && std::is_same<LB, PlainOp>::value) {
            const DmaKindInfo ki = dma_kind_info(p.kind);
            ok = p.kind >= 1 && p.kind <= kDmaKinds && dma_eligible(la, lb) && streamk_blocks(ep, M, N, K, batch, ki.bm, ki.bn, ki.bk, ki.nw, ki.occ, ki.lds) > 0;
        }
        if (!ok) { p.splitk = 1; streamk = false; }
    }
    if (ep.stat && streamk) {
        if (ep.mode != 0 || ep.res || ep.relu || ep.mask) ep.stat = nullptr;      // the owner of a cut tile holds the complete sum: statistics as usual
        if (!ep.stat && ep.stat_nparts) *ep.stat_nparts = 0;
    } else if (ep.stat) {
        // fused output statistics need the whole reduction in one block: no k-split.  A cached two-pass plan keeps its speed and reports
        // "no statistics" (nparts 0): the caller runs its separate reduction for this (rare: <= 256-tile) output
        if (p.splitk >= kTwoPass && p.kind >= 1 && twopass_ok(ep, M, N, batch, p.splitk - kTwoPass)) ep.stat = nullptr;
        else if (p.splitk > 1) p.splitk = 1;
        if (ep.mode != 0 || ep.res || ep.relu || ep.mask) ep.stat = nullptr;
        if (!ep.stat && ep.stat_nparts) *ep.stat_nparts = 0;
    }
    if (streamk) {
    } else if (p.splitk >= kTwoPass) {
        if (p.kind < 1 || !twopass_ok(ep, M, N, batch, p.splitk - kTwoPass)) p.splitk = 1;     // no scratch on this call / not a plain output
    } else {
        if (!sk_ok) p.splitk = 1;
        if (p.splitk > 1) ep.mode = 2;
    }
    if constexpr (std::is_same<LA, PlainOp>::value && std::is_same<LB, PlainOp>::value) {
        // tf_gemm_pair_begin(): a 64 x 64-tile register-staged launch with vector operands is held back so that tf_gemm_pair_end() can put
        // it into one grid with its partner (gemm_pair.cpp); anything else launches right here, as always
        if (pair_capturing() && p.kind == 0 && p.bm == 64 && p.bn == 64 && (p.bk == 16 || p.bk == 32) && p.splitk < kTwoPass && batch == 1 && la.vec && lb.vec &&
            !ep.stat && pair_hold(la, lb, ep, M, N, K, A_KC, B_KC, p, what))
            return 0;
    }
    launch_plan<LA, A_KC, LB, B_KC>(p, la, lb, ep, M, N, K, batch, stream);
    return launch_status(what);
}

}  // namespace tf

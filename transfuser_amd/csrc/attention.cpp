// Fused self-attention of the GPT fusion stages (transfuser.py:510-527): for one (sample, head) and a tile of 32 token rows a workgroup
// computes  S = q k^T / sqrt(hs)  ->  softmax  ->  attn_drop  ->  @ v  without the (B, nh, T, T) score matrix ever leaving the CU
// (before: 2 batched GEMM launches + a softmax launch per layer forward, 5 GEMMs + softmax backward, scores / probabilities
// / dropped probabilities round-tripping HBM).  T <= 192 (the model has T = 5 x 22 + 8 x 8 = 174), hs <= 384 (18 / 54 / 144 / 378).
//
// All three kernels share one skeleton, on a tile of 32 "row" tokens against ALL T "column" tokens:
//   phase A  one or two score-shaped products [32 x 192] contracted over hs: six waves, one 32 x 32 fp32-MFMA accumulator per product each;
//            every lane fetches its own MFMA fragments straight from global memory (8-byte loads, k slots permuted so that a lane half
//            owns consecutive channels), double-buffered in registers - no LDS staging, no barrier (round 3, second version: the first one
//            staged 16-deep operand tiles through LDS with a barrier per chunk and one 106 KB workgroup per CU; it was latency-bound and
//            no faster than the five batched GEMMs + softmax it replaced);
//   middle   the score rows live in LDS ([32][193] floats): softmax / its backward / dropout mask regenerated from the counter RNG;
//   phase B  one or two [32 x hs] products contracted over T with the LDS-resident scores as the A operand, B fragments from global memory.
// forward:   rows = queries: S = Q K^T; P = softmax(S); Y = drop(P) V; saves only L_i = max_i + log(sum_i) per row (T floats per head).
// backward dq:   rows = queries: recomputes P = exp(S - L), dPd = dY V^T, D_i = sum_j dP_ij P_ij, dS = P (dP - D); dQ = dS K / sqrt(hs).
// backward dkv:  rows = keys: the transposed products K Q^T, V dY^T; dS^T with L_i, D_i per COLUMN; dK = dS^T Q / sqrt(hs), dV = Pd^T dY.
// The dropout mask of element (query i, key j) is the one tf_dropout_f32 draws for the flat index ((b nh + h) T + i) Tp + j of the
// (B nh, T, Tp) probability tensor (Tp = T rounded up to 4), i.e. the unfused path's mask - parity tests share their masks.
// Exact fp32 MFMA in every precision mode: the contraction is 1.5 % of the step's FLOPs, its cost was launches and HBM round trips.
#include "tf_common.h"
#include <stdlib.h>
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int kR = 32;             // row tokens per workgroup
constexpr int kNC = 192;           // column tokens (T padded)
constexpr int kNW = 6;             // waves: one 32-column tile of the scores each
constexpr int kNT = 64 * kNW;
constexpr int kKH = 8;             // phase A: contraction elements per lane half and chunk (chunk = 16 channels, 8 MFMA k-steps)
constexpr int kSP = kNC + 1;       // score matrix pitch [32][193]: conflict-free as MFMA A operand (lane = row) and as row-wise softmax input
constexpr int kHS = 384;           // max head size
constexpr int kTB = 16;            // phase B: k-steps (token pairs) per register chunk

struct AtGeom {
    int B, nh, T, hs, C, Tp;
    float alpha, keep_scale;
    uint32_t site, thresh;
    int dbg;                       // TF_ATT_DBG (timing diagnosis, results are garbage): 1 no phase A, 2 no phase B, 4 no softmax / dS middle
};

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- phase A: acc[p] (32 x 32 per wave) = sum_k R_p[row0 + i][k] * C_p[wave * 32 + j][k], p < NP; both operands row-major, k contiguous.
// No LDS and no barrier: the operands are tiny (<= 192 x 384) and every wave multiplies ONE 32 x 32 tile, so each lane fetches its own MFMA
// fragments straight from global memory (L2 / L1 resident after the first touch).  The MFMA's k slots are permuted so that lane half `hi`
// owns kKH CONSECUTIVE channels of a 16-channel chunk: a fragment row is four 8-byte loads (head offsets h * hs are only 8-byte aligned:
// hs = 18 / 54 / 378), the next chunk's loads are in flight while the current one is multiplied.  hs must be even (pairs are all-or-nothing).
template <int NP>
__device__ __forceinline__ void phase_a(f32x16 (&acc)[NP], const float* const (&rsrc)[NP], const long (&rld)[NP], const float* const (&csrc)[NP],
                                        const long (&cld)[NP], int row0, int T, int hs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int rrow = row0 + l31, crow = wave * 32 + l31;
    const bool rok = rrow < T, cok = crow < T;
    const float* rp[NP];
    const float* cp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        rp[p] = rsrc[p] + (long)(rok ? rrow : 0) * rld[p];
        cp[p] = csrc[p] + (long)(cok ? crow : 0) * cld[p];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    }
    // Loads must be UNCONDITIONAL for the prefetch to overlap: with predicated loads (or a select / factor applied at fetch time) hipcc waits
    // for every load before the MFMA group, i.e. the next chunk's latency is exposed once per chunk (first version: no faster than the
    // LDS-staged kernel).  So: out-of-range rows fetch row 0 (finite garbage - their score rows / columns are masked by every consumer and
    // never stored), full 16-channel chunks are fetched bare, and only the ragged last chunk zeroes its dead channels (one exposed wait).
    float2 fa[2][NP][kKH / 2], fb[2][NP][kKH / 2];
    auto fetch = [&](int buf, int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < kKH / 2; ++q) {
                const int k = k0 + kKH * hi + 2 * q;
                fa[buf][p][q] = *reinterpret_cast<const float2*>(rp[p] + k);
                fb[buf][p][q] = *reinterpret_cast<const float2*>(cp[p] + k);
            }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < kKH / 2; ++q) {
                mfma_32x32x2(fa[buf][p][q].x, fb[buf][p][q].x, acc[p]);
                mfma_32x32x2(fa[buf][p][q].y, fb[buf][p][q].y, acc[p]);
            }
    };
    // straight-line trips (every fetch unconditional, the chunk index clamped instead): hipcc then waits with counted vmcnt(N), i.e. only for
    // the buffer it is about to multiply - a conditional fetch makes the wait a vmcnt(0) at the join
    const int nfull = hs / (2 * kKH), npair = nfull >> 1;
    if (nfull > 0) fetch(0, 0);
    for (int i = 0; i < npair; ++i) {
        fetch(1, (2 * i + 1) * 2 * kKH);
        mma(0);
        const int nx = 2 * i + 2 < nfull ? 2 * i + 2 : nfull - 1;
        fetch(0, nx * 2 * kKH);
        mma(1);
    }
    if (nfull & 1) mma(0);                                // buffer 0 holds chunk nfull - 1
    if (hs - nfull * 2 * kKH > 0) {                       // ragged tail: dead channels read channel 0 and are zeroed on the row operand
        const int k0 = nfull * 2 * kKH;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < kKH / 2; ++q) {
                const int k = k0 + kKH * hi + 2 * q;
                const bool kok = k < hs;
                const int kc = kok ? k : 0;
                const float2 va = *reinterpret_cast<const float2*>(rp[p] + kc);
                const float m = kok ? 1.f : 0.f;
                fa[0][p][q] = make_float2(va.x * m, va.y * m);
                fb[0][p][q] = *reinterpret_cast<const float2*>(cp[p] + kc);
            }
        mma(0);
    }
}

// ---- phase A with the ROW operand staged in LDS (round 6).  The 32 x hs row tile is the same for all six waves - fetched per lane it is half of the
// phase's load instructions, six times over, and with the column operand's 6 x 32 rows it overflows the 32 KB vector L1 between two chunks of a row's
// cache line (PMC: the TCP waits on pending misses for half of the kernel, MFMA busy 28 %).  Here the workgroup copies the tile once (8-byte coalesced
// loads, rows >= T and the channels of the ragged last chunk zero), one barrier, and the A fragments are ds_read_b64 (pitch 386: pitch / 2 odd - the 32
// rows of a half wave fall into different bank pairs); only the column operand's fragments still come from global memory, double-buffered as before.
constexpr int kRP = 386;           // row-tile pitch in LDS (floats)
template <int NP>
__device__ __forceinline__ void phase_a_lds(f32x16 (&acc)[NP], float* Rl, const float* const (&rsrc)[NP], const long (&rld)[NP], const float* const (&csrc)[NP],
                                            const long (&cld)[NP], int row0, int T, int hs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int crow = wave * 32 + l31;
    const bool cok = crow < T;
    const float* cp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        cp[p] = csrc[p] + (long)(cok ? crow : 0) * cld[p];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    }
    float2 fb[2][NP][kKH / 2];
    auto fetch = [&](int buf, int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < kKH / 2; ++q) fb[buf][p][q] = *reinterpret_cast<const float2*>(cp[p] + k0 + kKH * hi + 2 * q);
    };
    const int nfull = hs / (2 * kKH), npair = nfull >> 1, h2 = hs >> 1, hp2 = (hs + 2 * kKH - 1) / (2 * kKH) * kKH;       // pairs per row incl. the zero padding
    if (nfull > 0) fetch(0, 0);                                   // the first column fragments travel while the row tile is staged
    {
        constexpr int NQ = (kHS / 2 + 63) / 64;                   // float2 per lane and row
#pragma unroll
        for (int p = 0; p < NP; ++p)
            for (int i = wave; i < kR; i += kNW) {                // a wave copies whole rows: no index division
                const int row = row0 + i;
                const float2* src = reinterpret_cast<const float2*>(rsrc[p] + (long)(row < T ? row : 0) * rld[p]);
                float2 v[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) { const int c = lane + 64 * q; v[q] = src[c < h2 ? c : 0]; }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = lane + 64 * q;
                    if (c < hp2) *reinterpret_cast<float2*>(Rl + (p * kR + i) * kRP + 2 * c) = (row < T && c < h2) ? v[q] : make_float2(0.f, 0.f);
                }
            }
    }
    __syncthreads();
    const float* ap = Rl + l31 * kRP + kKH * hi;
    auto mma = [&](int buf, int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < kKH / 2; ++q) {
                const float2 a = *reinterpret_cast<const float2*>(ap + p * kR * kRP + k0 + 2 * q);
                mfma_32x32x2(a.x, fb[buf][p][q].x, acc[p]);
                mfma_32x32x2(a.y, fb[buf][p][q].y, acc[p]);
            }
    };
    for (int i = 0; i < npair; ++i) {
        fetch(1, (2 * i + 1) * 2 * kKH);
        mma(0, 2 * i * 2 * kKH);
        const int nx = 2 * i + 2 < nfull ? 2 * i + 2 : nfull - 1;
        fetch(0, nx * 2 * kKH);
        mma(1, (2 * i + 1) * 2 * kKH);
    }
    if (nfull & 1) mma(0, (nfull - 1) * 2 * kKH);             // buffer 0 holds chunk nfull - 1
    if (hs - nfull * 2 * kKH > 0) {                               // ragged tail: the row tile holds zeros there; the column operand's dead channels read channel 0
        const int k0 = nfull * 2 * kKH;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < kKH / 2; ++q) {
                const int k = k0 + kKH * hi + 2 * q;
                fb[0][p][q] = *reinterpret_cast<const float2*>(cp[p] + (k < hs ? k : 0));
            }
        mma(0, k0);
    }
}

// the wave's accumulator tile -> score matrix Sm[i][wave * 32 + j] (x scale)
__device__ __forceinline__ void put_scores(float* Sm, const f32x16& acc, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) Sm[acc_row(r, hi) * kSP + wave * 32 + l31] = acc[r] * scale;
}

// ---- phase B: out[row0 + i][c] = scale * sum_t Sm[i][t] * bsrc[t][c], c < hs; wave w owns the column tiles w and w + 6.  The A operand is
// the LDS-resident score tile; the B fragments (lane = output column, one token per k slot) come straight from global memory: a half wave
// reads one 128-byte line of a token row per k-step, kTB k-steps in flight ahead of the MFMAs.  No barrier inside.
__device__ __forceinline__ void phase_b(const float* Sm, const float* bsrc, long bld, int T, int hs, float* out, long old, int row0, float scale) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int ntile = (hs + 31) >> 5;
    for (int u = 0; u < 2; ++u) {
        const int tile = wave + u * kNW;
        if (tile >= ntile) break;                               // wave-uniform
        const int col = tile * 32 + l31;
        const bool colok = col < hs;
        const float* bp = bsrc + (colok ? col : 0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float fb[2][kTB];
        auto fetch = [&](int buf, int t0) {            // unconditional (see phase_a): tokens >= T read token T - 1; their score column is exactly 0
#pragma unroll
            for (int q = 0; q < kTB; ++q) {
                const int t = t0 + 2 * q + hi;
                fb[buf][q] = bp[(long)(t < T ? t : T - 1) * bld];
            }
        };
        const int nch = (T + 2 * kTB - 1) / (2 * kTB), npair = nch >> 1;
        const float* ap = Sm + l31 * kSP + hi;
        auto mma = [&](int buf, int c) {
#pragma unroll
            for (int q = 0; q < kTB; ++q) mfma_32x32x2(ap[c * 2 * kTB + 2 * q], fb[buf][q], acc);
        };
        fetch(0, 0);
        for (int i = 0; i < npair; ++i) {                       // straight-line trips, see phase_a
            fetch(1, (2 * i + 1) * 2 * kTB);
            mma(0, 2 * i);
            const int nx = 2 * i + 2 < nch ? 2 * i + 2 : nch - 1;
            fetch(0, nx * 2 * kTB);
            mma(1, 2 * i + 1);
        }
        if (nch & 1) mma(0, nch - 1);
        if (colok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + acc_row(r, hi);
                if (row < T) out[(long)row * old + col] = acc[r] * scale;
            }
        }
    }
}

// qkv: (B*T, 3C) = [key | query | value] (transfuser.py:500-502 order); head h owns columns h*hs .. of each third
template <bool RL>
__global__ void __launch_bounds__(kNT, 1) attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ y, float* __restrict__ lse, AtGeom g,
                                                               const uint32_t* __restrict__ seed) {
    __shared__ float Sm[kR * kSP];          // 24.7 KB: several workgroups per CU
    const int bh = blockIdx.y, b = bh / g.nh, h = bh - b * g.nh, row0 = blockIdx.x * kR;
    const long ld3 = 3L * g.C;
    const float* kp = qkv + (long)b * g.T * ld3 + (long)h * g.hs;
    const float* qp = kp + g.C;
    const float* vp = kp + 2 * g.C;
    f32x16 acc[1];
    const float* const rs[1] = {qp};
    const float* const cs[1] = {kp};
    const long rl[1] = {ld3}, cl[1] = {ld3};
    __shared__ float Rl[RL ? kR * kRP : 4];
    if (g.dbg & 1) { for (int r = 0; r < 16; ++r) acc[0][r] = 0.f; }
    else if (RL) phase_a_lds<1>(acc, Rl, rs, rl, cs, cl, row0, g.T, g.hs);
    else phase_a<1>(acc, rs, rl, cs, cl, row0, g.T, g.hs);
    put_scores(Sm, acc[0], g.alpha);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t sd = g.thresh ? *seed : 0u;
    // the wave's rows i = wave, wave + 6, ... (six for waves 0 / 1, five otherwise) TOGETHER: the row maxima and the row sums are six independent
    // reductions whose exchange steps are issued back to back (round 6; one row after the other every one of the 12 wave reductions was its own chain
    // of six dependent ds_bpermute round trips: ~5 us of the kernel's 15 us floor)
    constexpr int NR = (kR + kNW - 1) / kNW;
    float v[NR][3], mx[NR], sum[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = wave + r * kNW, ic = i < kR ? i : kR - 1;
        mx[r] = -3.0e38f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = lane + 64 * q;
            v[r][q] = j < g.T ? Sm[ic * kSP + j] : -3.0e38f;
            mx[r] = fmaxf(mx[r], v[r][q]);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int r = 0; r < NR; ++r) mx[r] = fmaxf(mx[r], shfl_xor(mx[r], m));
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        sum[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = lane + 64 * q;
            v[r][q] = j < g.T ? expf(v[r][q] - mx[r]) : 0.f;
            sum[r] += v[r][q];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int r = 0; r < NR; ++r) sum[r] += shfl_xor(sum[r], m);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = wave + r * kNW;
        if (i < kR) {                                  // wave-uniform
            float* srow = Sm + i * kSP;
            const float inv = 1.0f / sum[r];
            const int row = row0 + i;
            const uint32_t base = (uint32_t)(((long)bh * g.T + row) * g.Tp);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int j = lane + 64 * q;
                float p = v[r][q] * inv;
                if (g.thresh) p = dropout_keep(sd, g.site, base + (uint32_t)j, g.thresh) ? p * g.keep_scale : 0.f;
                srow[j] = j < g.T ? p : 0.f;
            }
            if (lane == 0 && row < g.T) lse[(long)bh * g.T + row] = mx[r] + logf(sum[r]);
        }
    }
    __syncthreads();
    if (!(g.dbg & 2)) phase_b(Sm, vp, ld3, g.T, g.hs, y + (long)b * g.T * g.C + (long)h * g.hs, g.C, row0, 1.f);
}

// rows = queries: dQ, and D_i = sum_j dP_ij P_ij for the dkv kernel.  HAVE_D: D_i comes in (attention_dsum_kernel), no reduction and nothing written to dsum
template <bool HAVE_D, bool RL>
__device__ __forceinline__ void bwd_dq_body(float* S0, float* S1, float* Rl, const float* __restrict__ qkv, const float* __restrict__ dy, const float* __restrict__ lse,
                                            float* __restrict__ dqkv, float* __restrict__ dsum, const AtGeom& g, const uint32_t* __restrict__ seed) {
    const int bh = blockIdx.y, b = bh / g.nh, h = bh - b * g.nh, row0 = blockIdx.x * kR;
    const long ld3 = 3L * g.C;
    const float* kp = qkv + (long)b * g.T * ld3 + (long)h * g.hs;
    const float* qp = kp + g.C;
    const float* vp = kp + 2 * g.C;
    const float* dyp = dy + (long)b * g.T * g.C + (long)h * g.hs;
    f32x16 acc[2];
    const float* const rs[2] = {qp, dyp};
    const float* const cs[2] = {kp, vp};
    const long rl[2] = {ld3, (long)g.C}, cl[2] = {ld3, ld3};
    if (g.dbg & 1) { for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f; }
    else if (RL) phase_a_lds<2>(acc, Rl, rs, rl, cs, cl, row0, g.T, g.hs);       // S = Q K^T, dPd = dY V^T
    else phase_a<2>(acc, rs, rl, cs, cl, row0, g.T, g.hs);
    put_scores(S0, acc[0], g.alpha);
    put_scores(S1, acc[1], 1.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t sd = g.thresh ? *seed : 0u;
    constexpr int NR = (kR + kNW - 1) / kNW;              // the wave's rows together, as in the forward kernel: six independent dot-product reductions
    float p[NR][3], dp[NR][3], dot[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = wave + r * kNW, ic = i < kR ? i : kR - 1, row = row0 + ic;
        const float L = row < g.T ? lse[(long)bh * g.T + row] : 0.f;
        const uint32_t base = (uint32_t)(((long)bh * g.T + row) * g.Tp);
        dot[r] = (HAVE_D && row < g.T) ? dsum[(long)bh * g.T + row] : 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = lane + 64 * q;
            const bool ok = j < g.T;
            p[r][q] = ok ? expf(S0[ic * kSP + j] - L) : 0.f;
            dp[r][q] = ok ? S1[ic * kSP + j] : 0.f;
            if (g.thresh) dp[r][q] = (ok && dropout_keep(sd, g.site, base + (uint32_t)j, g.thresh)) ? dp[r][q] * g.keep_scale : 0.f;
            if (!HAVE_D) dot[r] += p[r][q] * dp[r][q];
        }
    }
    if (!HAVE_D) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
            for (int r = 0; r < NR; ++r) dot[r] += shfl_xor(dot[r], m);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = wave + r * kNW, row = row0 + i;
        if (i < kR) {                                  // wave-uniform
#pragma unroll
            for (int q = 0; q < 3; ++q) S0[i * kSP + lane + 64 * q] = p[r][q] * (dp[r][q] - dot[r]);
            if (!HAVE_D && lane == 0 && row < g.T) dsum[(long)bh * g.T + row] = dot[r];
        }
    }
    __syncthreads();
    if (!(g.dbg & 2)) phase_b(S0, kp, ld3, g.T, g.hs, dqkv + (long)b * g.T * ld3 + g.C + (long)h * g.hs, ld3, row0, g.alpha);      // dQ = dS K / sqrt(hs)
}
template <bool RL>
__global__ void __launch_bounds__(kNT, 1) attention_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dy, const float* __restrict__ lse,
                                                                  float* __restrict__ dqkv, float* __restrict__ dsum, AtGeom g,
                                                                  const uint32_t* __restrict__ seed) {
    __shared__ float S0[kR * kSP];
    __shared__ float S1[kR * kSP];
    __shared__ float Rl[RL ? 2 * kR * kRP : 4];
    bwd_dq_body<false, RL>(S0, S1, Rl, qkv, dy, lse, dqkv, dsum, g, seed);
}

// rows = keys: dK, dV
template <bool RL>
__device__ __forceinline__ void bwd_dkv_body(float* S0, float* S1, float* Rl, const float* __restrict__ qkv, const float* __restrict__ dy, const float* __restrict__ lse,
                                             const float* __restrict__ dsum, float* __restrict__ dqkv, const AtGeom& g, const uint32_t* __restrict__ seed) {
    const int bh = blockIdx.y, b = bh / g.nh, h = bh - b * g.nh, row0 = blockIdx.x * kR;
    const long ld3 = 3L * g.C;
    const float* kp = qkv + (long)b * g.T * ld3 + (long)h * g.hs;
    const float* qp = kp + g.C;
    const float* vp = kp + 2 * g.C;
    const float* dyp = dy + (long)b * g.T * g.C + (long)h * g.hs;
    f32x16 acc[2];
    const float* const rs[2] = {kp, vp};
    const float* const cs[2] = {qp, dyp};
    const long rl[2] = {ld3, ld3}, cl[2] = {ld3, (long)g.C};
    if (g.dbg & 1) { for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f; }
    else if (RL) phase_a_lds<2>(acc, Rl, rs, rl, cs, cl, row0, g.T, g.hs);       // S^T = K Q^T, dPd^T = V dY^T
    else phase_a<2>(acc, rs, rl, cs, cl, row0, g.T, g.hs);
    put_scores(S0, acc[0], g.alpha);
    put_scores(S1, acc[1], 1.f);
    __syncthreads();
    const uint32_t sd = g.thresh ? *seed : 0u;
    const float* Lp = lse + (long)bh * g.T;
    const float* Dp = dsum + (long)bh * g.T;
    for (int idx = threadIdx.x; idx < kR * kNC; idx += kNT) {
        const int jj = idx / kNC, i = idx - jj * kNC;           // key row0 + jj (tile row), query i (column)
        float ds = 0.f, pd = 0.f;
        if (i < g.T) {
            const float p = expf(S0[jj * kSP + i] - Lp[i]);
            float dp = S1[jj * kSP + i];
            pd = p;
            if (g.thresh) {
                const bool keep = dropout_keep(sd, g.site, (uint32_t)(((long)bh * g.T + i) * g.Tp + row0 + jj), g.thresh);
                dp = keep ? dp * g.keep_scale : 0.f;
                pd = keep ? p * g.keep_scale : 0.f;
            }
            ds = p * (dp - Dp[i]);
        }
        S0[jj * kSP + i] = ds;
        S1[jj * kSP + i] = pd;
    }
    __syncthreads();
    float* dk = dqkv + (long)b * g.T * ld3 + (long)h * g.hs;
    if (g.dbg & 2) return;
    phase_b(S0, qp, ld3, g.T, g.hs, dk, ld3, row0, g.alpha);                 // dK = dS^T Q / sqrt(hs)
    phase_b(S1, dyp, g.C, g.T, g.hs, dk + 2 * g.C, ld3, row0, 1.f);          // dV = Pd^T dY
}
template <bool RL>
__global__ void __launch_bounds__(kNT, 1) attention_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dy, const float* __restrict__ lse,
                                                                   const float* __restrict__ dsum, float* __restrict__ dqkv, AtGeom g,
                                                                   const uint32_t* __restrict__ seed) {
    __shared__ float S0[kR * kSP];
    __shared__ float S1[kR * kSP];
    __shared__ float Rl[RL ? 2 * kR * kRP : 4];
    bwd_dkv_body<RL>(S0, S1, Rl, qkv, dy, lse, dsum, dqkv, g, seed);
}

// ---- round 6: both backward kernels in ONE grid.  D_i = sum_j dP_ij P_ij = dY_i . Y_i (the rows of the forward output; holds with attention dropout:
// sum_j Pd_ij (dY_i . V_j) = dY_i . sum_j Pd_ij V_j), so it needs no score matrix: attention_dsum_kernel forms it from dy and y (one wave per row and
// head), and the dK / dV half no longer waits for the dQ half.  blockIdx.z = 0: keys (the longer half starts first), 1: queries.  A launch of one
// half is 240 six-wave workgroups on 256 CUs - one per CU, two SIMDs with two waves and two with one; the merged grid puts two workgroups on a CU.
__global__ void __launch_bounds__(256) attention_dsum_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dsum, int B, int T,
                                                             int C, int nh) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int total = B * T * nh, hs = C / nh;
    const bool live = wid < total;
    const int w = live ? wid : 0, h = w % nh, bt = w / nh;
    const float2* a = reinterpret_cast<const float2*>(dy + (long)bt * C + (long)h * hs);
    const float2* c = reinterpret_cast<const float2*>(y + (long)bt * C + (long)h * hs);
    const int n2 = hs >> 1;
    float2 u[kHS / 128], v[kHS / 128];
#pragma unroll
    for (int q = 0; q < kHS / 128; ++q) {
        const int i = lane + 64 * q, ic = i < n2 ? i : n2 - 1;
        u[q] = a[ic]; v[q] = c[ic];
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < kHS / 128; ++q) s += (lane + 64 * q < n2) ? u[q].x * v[q].x + u[q].y * v[q].y : 0.f;
    s = wave_sum(s);
    if (live && lane == 0) {
        const int b = bt / T, t = bt - b * T;
        dsum[((long)b * nh + h) * T + t] = s;
    }
}
template <bool RL>
__global__ void __launch_bounds__(kNT, 1) attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dy, const float* __restrict__ lse,
                                                               float* __restrict__ dsum, float* __restrict__ dqkv, AtGeom g, const uint32_t* __restrict__ seed) {
    __shared__ float S0[kR * kSP];
    __shared__ float S1[kR * kSP];
    __shared__ float Rl[RL ? 2 * kR * kRP : 4];
    if (blockIdx.z == 0) bwd_dkv_body<RL>(S0, S1, Rl, qkv, dy, lse, dsum, dqkv, g, seed);
    else bwd_dq_body<true, RL>(S0, S1, Rl, qkv, dy, lse, dqkv, dsum, g, seed);
}

// TF_ATT_ROWLDS: 0 never, 1 (default) the backward kernels at head sizes >= 256, 2 everywhere.  Measured in-graph (profiles/r06_attention_lab_row_lds.txt):
// backward at hs = 378 170.5 -> 145.5 us; at hs <= 144 the staging prologue and its barrier cost 5-7 us per backward and 1-2 us per forward, and the
// forward kernel does not gain at any size (its phase A is one product).
inline int row_lds_mode() { static const int m = [] { const char* e = getenv("TF_ATT_ROWLDS"); return e ? atoi(e) : 1; }(); return m; }
inline bool row_lds(int hs, bool backward) { const int m = row_lds_mode(); return m >= 2 || (m == 1 && backward && hs >= 256); }
inline int make_geom(AtGeom& g, int B, int T, int C, int nh, uint32_t site, float pdrop, const char* who) {
    TF_REQUIRE(B > 0 && T > 0 && T <= kNC && nh > 0 && C % nh == 0 && C / nh <= kHS && (C / nh) % 2 == 0, "%s: needs T <= %d and an even head size <= %d (got T=%d, hs=%d)",
               who, kNC, kHS, T, nh > 0 ? C / nh : -1);
    TF_REQUIRE(pdrop >= 0.f && pdrop < 1.f && (long)B * nh * T * ((T + 3) / 4 * 4) < (1L << 32), "%s: bad dropout probability / index range", who);
    g.B = B; g.nh = nh; g.T = T; g.hs = C / nh; g.C = C; g.Tp = (T + 3) / 4 * 4;
    g.alpha = 1.0f / sqrtf((float)g.hs);
    g.site = site;
    g.thresh = (uint32_t)((double)pdrop * 4294967296.0);
    g.keep_scale = 1.f / (1.f - pdrop);
    static const int dbg = [] { const char* e = getenv("TF_ATT_DBG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    return 0;
}

}  // namespace

extern "C" int tf_attention_supported(int T, int C, int nh) { return (T > 0 && T <= kNC && nh > 0 && C % nh == 0 && C / nh <= kHS && (C / nh) % 2 == 0) ? 1 : 0; }

extern "C" int tf_attention_fwd_f32(const float* qkv, float* y, float* lse, int B, int T, int C, int nh, const uint32_t* seed_dev, uint32_t site, float pdrop,
                                    void* stream) {
    TF_REQUIRE(qkv && y && lse && (pdrop == 0.f || seed_dev), "tf_attention_fwd_f32: null argument");
    AtGeom g;
    if (int e = make_geom(g, B, T, C, nh, site, pdrop, "tf_attention_fwd_f32")) return e;
    if (row_lds(g.hs, false)) TF_LAUNCH(attention_fwd_kernel<true>, dim3(cdiv(T, kR), B * nh), dim3(kNT), stream, qkv, y, lse, g, seed_dev);
    else TF_LAUNCH(attention_fwd_kernel<false>, dim3(cdiv(T, kR), B * nh), dim3(kNT), stream, qkv, y, lse, g, seed_dev);
    return launch_status("tf_attention_fwd_f32");
}

extern "C" int tf_attention_bwd_y_f32(const float* qkv, const float* dy, const float* y, const float* lse, float* dqkv, float* dsum, int B, int T, int C, int nh,
                                      const uint32_t* seed_dev, uint32_t site, float pdrop, void* stream) {
    TF_REQUIRE(qkv && dy && y && lse && dqkv && dsum && (pdrop == 0.f || seed_dev), "tf_attention_bwd_y_f32: null argument");
    AtGeom g;
    if (int e = make_geom(g, B, T, C, nh, site, pdrop, "tf_attention_bwd_y_f32")) return e;
    TF_REQUIRE((((uintptr_t)dy | (uintptr_t)y) & 7) == 0, "tf_attention_bwd_y_f32: dy / y must be 8-byte aligned");
    TF_LAUNCH(attention_dsum_kernel, dim3(cdiv((long)B * T * nh, 4)), dim3(256), stream, dy, y, dsum, B, T, C, nh);
    if (row_lds(g.hs, true)) TF_LAUNCH(attention_bwd_kernel<true>, dim3(cdiv(T, kR), B * nh, 2), dim3(kNT), stream, qkv, dy, lse, dsum, dqkv, g, seed_dev);
    else TF_LAUNCH(attention_bwd_kernel<false>, dim3(cdiv(T, kR), B * nh, 2), dim3(kNT), stream, qkv, dy, lse, dsum, dqkv, g, seed_dev);
    return launch_status("tf_attention_bwd_y_f32");
}

extern "C" int tf_attention_bwd_f32(const float* qkv, const float* dy, const float* lse, float* dqkv, float* dsum, int B, int T, int C, int nh,
                                    const uint32_t* seed_dev, uint32_t site, float pdrop, void* stream) {
    TF_REQUIRE(qkv && dy && lse && dqkv && dsum && (pdrop == 0.f || seed_dev), "tf_attention_bwd_f32: null argument");
    AtGeom g;
    if (int e = make_geom(g, B, T, C, nh, site, pdrop, "tf_attention_bwd_f32")) return e;
    if (row_lds(g.hs, true)) {
        TF_LAUNCH(attention_bwd_dq_kernel<true>, dim3(cdiv(T, kR), B * nh), dim3(kNT), stream, qkv, dy, lse, dqkv, dsum, g, seed_dev);
        TF_LAUNCH(attention_bwd_dkv_kernel<true>, dim3(cdiv(T, kR), B * nh), dim3(kNT), stream, qkv, dy, lse, (const float*)dsum, dqkv, g, seed_dev);
    } else {
        TF_LAUNCH(attention_bwd_dq_kernel<false>, dim3(cdiv(T, kR), B * nh), dim3(kNT), stream, qkv, dy, lse, dqkv, dsum, g, seed_dev);
        TF_LAUNCH(attention_bwd_dkv_kernel<false>, dim3(cdiv(T, kR), B * nh), dim3(kNT), stream, qkv, dy, lse, (const float*)dsum, dqkv, g, seed_dev);
    }
    return launch_status("tf_attention_bwd_f32");
}

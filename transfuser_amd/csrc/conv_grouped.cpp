// Direct (LDS-tiled) GROUPED 3x3 / stride 1 / pad 1 convolutions for the RegNetY bottlenecks (timm regnety_032: group width 24, i.e.
// C / 24 groups of a 24 -> 24 channel 3x3 convolution; transfuser.py:380,442 via timm) and their two gradients.
//
// Through the implicit-GEMM engine these launches ran at 18-30 TFLOP/s (profiles/r01_engine_census_eager_step.txt): a 24-wide group is a
// K = 216, N = 24 problem - 13.5 short k-tiles per 128 x 32 tile, every im2col element gathered with per-element index arithmetic and the
// input re-read 9x through L2.  Here a block owns ONE group: it stages the group's 9 x 24 x 24 weight panel in LDS once, then walks its
// share of 128-pixel tiles: the (TH+2) x (TW+2) x 24 input patch is staged once per tile and all 9 taps are read from LDS
// (v_mfma_f32_32x32x2_f32, one 32-pixel x 32-channel accumulator per wave, 108 MFMAs per tile and wave; the 24 channels occupy 24 of the
// 32 MFMA columns = the 75 % ceiling of any MFMA mapping of a 24-wide group).  Tiles are 8 x 16 or 4 x 32 pixels (picked per map
// width: 44 -> 3 x 16, 88 -> 3 x 32, 16 / 32 / 64 exact).  dgrad = the same kernel on dY with the flipped, transposed panel.  wgrad keeps
// 9 accumulators (one per tap, 24 x 24 used of 32 x 32) per wave with the tile's pixels as the K dimension, reduced over the block's
// waves through LDS and over a group's blocks by a second tiny kernel (deterministic, no atomics).
#include "tf_common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/transfuser_hip.h"

using namespace tf;
namespace tf { int gemm_precision(); }   // api.cpp (tf_set_precision): 1 = bf16-MFMA contractions, 2 = bf16x3 split (fp32-accurate) on the bf16 MFMA

namespace {

constexpr int CG = 24;              // channels per group (in and out)
constexpr int PP = 25;              // floats per patch pixel: 24 channels + 1 (odd pitch: conflict-free MFMA A fetch)
constexpr int WP = 36;              // pitch of a weight row (32 output columns + 4)

struct GcGeom { int B, H, W, C, G, tiles_h, tiles_w, ntiles, nb, f16, dbg; };   // f16: the bf16 paths use IEEE-half operands instead (tf_set_precision(3))   // C = total channels (pixel stride), nb = blocks per group

// an integer the optimiser may not treat as loop-invariant: index arithmetic derived from it is RE-COMPUTED where it is used instead of being hoisted
// out of the tile loop into a dozen long-lived registers (round 6: the seven-wave weight gradient needs <= 80 VGPRs without spills)
__device__ __forceinline__ int opaque(int v) {
#ifndef TF_EMU
    asm volatile("" : "+v"(v));
#endif
    return v;
}

template <int TW> struct Tile {
    static constexpr int RW = 32 / TW;            // image rows per wave
    static constexpr int TH = 4 * RW;             // tile rows (4 waves)
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = PH * PW;
    static constexpr int NV = (NPIX * 6 + 255) / 256;   // float4 patch slots per thread (6 per pixel)
};

template <int TW> struct Tile2 {
    static constexpr int RW = 32 / TW, TH = 4 * RW;
    static constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1, NPIX = PH * PW;
};

// group panel -> LDS as wl[tap * CG + k][n] (n < 32 zero padded):  fwd   wl[tap][ci][co] = W[g*CG + co][tap][ci]
//                                                                   dgrad wl[tap][co][ci] = W[g*CG + co][8 - tap][ci]
// The group's panel is 24 x 9 x 24 = 5184 CONTIGUOUS floats: every thread issues its 21 (clamped, unconditional) loads back to back, then the
// first patch's loads, and only then scatters the panel into LDS - one memory round trip in front of the first tile.  (Round 3 gathered one element per loop trip under a predicate: hipcc put a
// vmcnt(0) into every one of the 27 trips, ~20 us of dependent latency in front of every forward / input-gradient launch.)
constexpr int WNE = 9 * CG * CG, WNL = (WNE + 255) / 256;
struct GroupWeightRegs { float v[WNL]; };
__device__ __forceinline__ void issue_group_weights(GroupWeightRegs& r, const float* __restrict__ w) {
#pragma unroll
    for (int p = 0; p < WNL; ++p) { const int e = threadIdx.x + p * 256; r.v[p] = w[e < WNE ? e : WNE - 1]; }
}
__device__ __forceinline__ void scatter_group_weights(float (*wl)[WP], const GroupWeightRegs& r, int dgrad) {
    for (int i = threadIdx.x; i < 9 * CG * 8; i += 256) wl[i >> 3][CG + (i & 7)] = 0.f;      // the 8 padding columns of the 32-wide MFMA operand
#pragma unroll
    for (int p = 0; p < WNL; ++p) {
        const int e = threadIdx.x + p * 256;
        const int co = e / (9 * CG), q = e - co * (9 * CG), tap = q / CG, ci = q - tap * CG;
        if (e < WNE) {
            if (dgrad) wl[(8 - tap) * CG + co][ci] = r.v[p]; else wl[tap * CG + ci][co] = r.v[p];
        }
    }
}

template <int TW>
__device__ __forceinline__ float4 load_patch_slot(const float* __restrict__ x, const GcGeom& g, int coff, int b, int h0, int w0, int s, bool* valid = nullptr) {
    typedef Tile<TW> T;
    const int pix = s / 6, c = (s - pix * 6) * 4;
    const int ph = pix / T::PW, pw = pix - ph * T::PW;
    const int h = h0 - 1 + ph, w = w0 - 1 + pw;
    const bool ok = pix < T::NPIX && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
    const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = w < 0 ? 0 : (w >= g.W ? g.W - 1 : w);      // clamped address: the load itself is unconditional
    const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * g.H + hc) * g.W + wc) * g.C + coff + c);
    if (valid) *valid = ok;
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int TW>
__device__ __forceinline__ void store_patch_slot(float* patch, int s, const float4 v) {
    const int pix = s / 6, c = (s - pix * 6) * 4;
    if (pix < Tile<TW>::NPIX) { float* q = patch + pix * PP + c; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
}
// BatchNorm apply of the PRODUCER folded into this consumer (timm ConvBnAct conv1 -> conv2 of a RegNetY bottleneck): the kernel reads the raw
// convolution output and stages max(x sc + sh, 0) - the normalised activation is never written to memory.  cf = [scale (24) | shift (24)] of the
// block's group in LDS; zero padding stays zero (the transform is applied to pixels inside the map only).
__device__ __forceinline__ float4 bnrelu4(const float4 v, const float* cf, int c, bool ok) {
    if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    return make_float4(fmaxf(v.x * cf[c] + cf[CG + c], 0.f), fmaxf(v.y * cf[c + 1] + cf[CG + c + 1], 0.f), fmaxf(v.z * cf[c + 2] + cf[CG + c + 2], 0.f),
                       fmaxf(v.w * cf[c + 3] + cf[CG + c + 3], 0.f));
}
__device__ __forceinline__ void stage_group_coef(float* cf, const float* __restrict__ in_coef, int coff, int C) {
    if (in_coef && threadIdx.x < 2 * CG) cf[threadIdx.x] = in_coef[threadIdx.x < CG ? coff + threadIdx.x : C + coff + threadIdx.x - CG];
}

// y[.., g*24 + co] = sum_{tap, ci} x[.. + tap, g*24 + ci] * W  (+ bias) (relu) (+= when accumulate); dgrad != 0: x is dY, y is dX
// X3 = the bf16x3-split instantiation (tf_set_precision(2)); the default binary holds the fp32 path and the runtime-selected bf16 path
// STAT: the forward also produces the BatchNorm statistics of its output (timm ConvBnAct: conv2 is followed by BatchNormAct2d): every wave
// keeps a running Welford triple (n, mean, M2) per channel over the tiles it computes, the block's 4 waves are merged through LDS at
// the end and ONE triple per (block, channel) goes to stat[(sub * 3 + {0,1,2}) * C + channel] - plain stores, nb parts per channel.
// F32T (round 6) = the exact-fp32 instantiation with the TRANSPOSED accumulator: the MFMA takes the weight panel as its A operand (i = output channel) and
// the patch as B (j = pixel), so a lane ends up with 12 output channels of ONE pixel in three groups of four consecutive channels - three
// 16-byte stores per lane and tile instead of sixteen 4-byte ones.  (tools/grouped_lab.py with TF_GC_DBG: at (10, 16, 44, 576) the 16 scalar stores were
// 8 us of a 37.7 us launch, MFMA 22, everything else 8 - and nothing overlaps, the blocks of a CU run in lockstep.)  The statistics of that layout: every
// lane of a half keeps the running Welford triple of its 12 channels, fed per tile with the count / mean / M2 over the wave's 32 pixels (half_sum).
template <int TW, bool X3 = false, bool STAT = false, bool F32T = false>
__global__ void __launch_bounds__(256, 2) conv3x3_grouped_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, GcGeom g, int dgrad, int relu, int accumulate, int prec,
                                                                 float* __restrict__ stat = nullptr, const float* __restrict__ in_coef = nullptr) {
    typedef Tile<TW> T;
    __shared__ float patch[T::NPIX * PP];
    __shared__ float wl[9 * CG][WP];
    __shared__ float cf[2 * CG];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int grp = blockIdx.x / g.nb, sub = blockIdx.x - grp * g.nb, coff = grp * CG;
    GroupWeightRegs wreg;
    issue_group_weights(wreg, w + (long)grp * CG * 9 * CG);          // 21 loads in flight; the first patch joins them before anything waits
    float4 pre[T::NV];
    unsigned okm = 0;                                                 // validity of the pre[] slots (in_coef: the transform must leave the zero padding zero)
    auto fetch = [&](int t) {
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        okm = 0;
#pragma unroll
        for (int p = 0; p < T::NV; ++p) {
            bool ok;
            pre[p] = load_patch_slot<TW>(x, g, coff, b, h0, w0, tid + p * 256, &ok);
            okm |= ok ? 1u << p : 0u;
        }
    };
    int tile = sub;
    if (tile < g.ntiles) fetch(tile);
    stage_group_coef(cf, in_coef, coff, g.C);
    scatter_group_weights(wl, wreg, dgrad);
    // this lane's pixel inside the wave's 32: (row, col) of the tile
    const int prow = wave * T::RW + l31 / TW, pcol = l31 % TW;
    float st_n = 0.f, st_mean = 0.f, st_m2 = 0.f;      // STAT: running triple of channel l31 over this wave's pixels (same in both lane halves)
    float tn = 0.f, tmean[F32T && STAT ? 12 : 1], tm2[F32T && STAT ? 12 : 1];      // F32T + STAT: this lane's triples of channels (e & 3) + 8 (e >> 2) + 4 hi, e < 12
#pragma unroll
    for (int e = 0; e < (F32T && STAT ? 12 : 1); ++e) { tmean[e] = 0.f; tm2[e] = 0.f; }
    for (; tile < g.ntiles; tile += g.nb) {
        __syncthreads();                           // previous tile's MFMAs are done with the patch (and the weights / coefficients are staged)
        if (in_coef) {
#pragma unroll
            for (int p = 0; p < T::NV; ++p) store_patch_slot<TW>(patch, tid + p * 256, bnrelu4(pre[p], cf, ((tid + p * 256) % 6) * 4, (okm >> p) & 1u));
        } else {
#pragma unroll
            for (int p = 0; p < T::NV; ++p) store_patch_slot<TW>(patch, tid + p * 256, pre[p]);
        }
        __syncthreads();
        const int nxt = tile + g.nb;
        if (nxt < g.ntiles && !(g.dbg & 4)) fetch(nxt);            // next patch travels while this one is multiplied
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!F32T && (X3 || prec)) {
            // bf16 MFMA (tf_set_precision(1); X3: bf16x3 split): per tap two 16-deep groups over the 24 (zero-padded to 32) input channels; lane half hi
            // owns channels 16 q + 8 hi .. + 7, so the upper half of the second group is all zeros
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const float* pa = patch + ((prow + kh) * T::PW + pcol + kw) * PP;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float a[8], b[8];
                    const bool live = (q == 0) || (hi == 0);
                    const int k0 = 16 * q + 8 * hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        a[j] = live ? pa[k0 + j] : 0.f;
                        b[j] = live ? wl[tap * CG + k0 + j][l31] : 0.f;
                    }
                    if constexpr (X3) mfma_32x32x16_x3(a, b, acc); else mfma_32x32x16_lp(a, b, acc, prec);
                }
            }
        } else if constexpr (!X3) {
            if (!(g.dbg & 1))
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const float* pa = patch + ((prow + kh) * T::PW + pcol + kw) * PP + hi;
                const float* pb = &wl[tap * CG + hi][l31];
#pragma unroll
                for (int kk = 0; kk < CG / 2; ++kk) {
                    if constexpr (F32T) mfma_32x32x2(pb[2 * kk * WP], pa[2 * kk], acc);      // D[i = co][j = pixel]
                    else mfma_32x32x2(pa[2 * kk], pb[2 * kk * WP], acc);                     // D[i = pixel][j = co]
                }
            }
        }
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        if constexpr (F32T) {
            const int hh = h0 + prow, ww = w0 + pcol;                    // this lane's pixel
            const bool pok = hh < g.H && ww < g.W;
            if (pok && !(g.dbg & 2)) {
                float* dst = y + (((long)b * g.H + hh) * g.W + ww) * g.C + coff + 4 * hi;
#pragma unroll
                for (int q = 0; q < 3; ++q) {                            // channels 8 q + 4 hi .. + 3 = accumulator elements 4 q .. 4 q + 3
                    float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                    if (bias) { const float4 bj = *reinterpret_cast<const float4*>(bias + coff + 8 * q + 4 * hi); v.x += bj.x; v.y += bj.y; v.z += bj.z; v.w += bj.w; }
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float4* d4 = reinterpret_cast<float4*>(dst + 8 * q);
                    if (accumulate) { const float4 o = *d4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *d4 = v;
                }
            }
            if constexpr (STAT) {
                // the wave's 32 pixels of this tile as ONE sample group per channel (as in the pixel-major kernel): count, mean and M2 about that mean over the
                // 32 lanes of the half (half_sum: four DPP adds + one 16-lane exchange per value), Chan-merged into the running triple every lane of the half keeps identically
                const float cnt = half_sum(pok ? 1.f : 0.f);
                if (cnt > 0.f) {                                         // (uniform over the half)
                    const float inv = 1.f / cnt, n_new = tn + cnt, wb = cnt / n_new, wab = tn * cnt / n_new;
#pragma unroll
                    for (int e = 0; e < 12; ++e) {
                        const float m_t = half_sum(pok ? acc[e] : 0.f) * inv, d0 = acc[e] - m_t;
                        const float q = half_sum(pok ? d0 * d0 : 0.f);
                        const float delta = m_t - tmean[e];
                        tmean[e] += delta * wb;
                        tm2[e] += q + delta * delta * wab;
                    }
                    tn = n_new;
                }
            }
        } else {
        if (l31 < CG && !(g.dbg & 2)) {
                const float bj = bias ? bias[coff + l31] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;           // pixel index inside the wave's 32
                    const int hh = h0 + wave * T::RW + i / TW, ww = w0 + i % TW;
                    if (hh < g.H && ww < g.W) {
                        float* dst = y + (((long)b * g.H + hh) * g.W + ww) * g.C + coff + l31;
                        float v = acc[e] + bj;
                        if (relu) v = fmaxf(v, 0.f);
                        *dst = accumulate ? *dst + v : v;
                    }
                }
            }
        if constexpr (STAT) {       // all 64 lanes take part in the shuffles; the padding columns (l31 >= 24) carry zeros and are never stored
            float s = 0.f, cnt = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
                const bool ok = (h0 + wave * T::RW + i / TW) < g.H && (w0 + i % TW) < g.W;
                s += ok ? acc[e] : 0.f;
                cnt += ok ? 1.f : 0.f;
            }
            s += shfl_xor(s, 32);
            cnt += shfl_xor(cnt, 32);
            if (cnt > 0.f) {                                             // wave-uniform (pixel validity does not depend on the channel)
                const float m_t = s / cnt;
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    const bool ok = (h0 + wave * T::RW + i / TW) < g.H && (w0 + i % TW) < g.W;
                    const float d = acc[e] - m_t;
                    q += ok ? d * d : 0.f;
                }
                q += shfl_xor(q, 32);
                const float n_new = st_n + cnt, delta = m_t - st_mean;       // Chan's pairwise merge of (st_n, st_mean, st_m2) and (cnt, m_t, q)
                st_mean += delta * (cnt / n_new);
                st_m2 += q + delta * delta * (st_n * cnt / n_new);
                st_n = n_new;
            }
        }
        }   // (F32T / plain epilogue)
    }
    if constexpr (STAT) {
        __syncthreads();                                                 // every wave is done with the patch: its memory carries the 4 triples
        float* red = patch;                                              // [wave][3][32]
        if constexpr (F32T) {
            // every lane of a half holds the wave's running triple of its 12 channels
            if (l31 == 0) {
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    const int co = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    red[(wave * 3 + 0) * 32 + co] = tn; red[(wave * 3 + 1) * 32 + co] = tmean[e]; red[(wave * 3 + 2) * 32 + co] = tm2[e];
                }
            }
        } else
        if (hi == 0) { red[(wave * 3 + 0) * 32 + l31] = st_n; red[(wave * 3 + 1) * 32 + l31] = st_mean; red[(wave * 3 + 2) * 32 + l31] = st_m2; }
        __syncthreads();
        if (tid < CG) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const float nb_ = red[(wv * 3 + 0) * 32 + tid], mb = red[(wv * 3 + 1) * 32 + tid], qb = red[(wv * 3 + 2) * 32 + tid];
                if (nb_ > 0.f) {
                    const float n_new = n + nb_, delta = mb - mean;
                    mean += delta * (nb_ / n_new);
                    m2 += qb + delta * delta * (n * nb_ / n_new);
                    n = n_new;
                }
            }
            float* o = stat + (long)sub * 3 * g.C + coff + tid;
            o[0] = n; o[g.C] = mean; o[2 * (long)g.C] = m2;
        }
    }
}

// dW[g*24 + co][tap][ci] (+)= sum_pixels dY[p][g*24 + co] * X[p + tap][g*24 + ci].  THREE waves per block, wave kh owns the taps (kh, 0..2): every
// wave walks ALL 128 pixels of the tile as the K dimension (A[i = co][k = pixel] = the staged dY tile, shared by the three waves; B[k][j = ci] = the
// patch shifted by the tap), so a wave's three accumulators ARE three finished taps of the block's partial panel - no cross-wave reduction,
// no barrier after the last tile, the panel rows go straight from the accumulator registers to part[block][tap][co (32)][ci (32)] and
// conv3x3_grouped_wgrad_reduce_kernel sums a group's panels in a fixed order (deterministic, no atomics).  31-33 KB of LDS and ~100 VGPRs
// per block: four to five blocks per CU (round 3: four waves splitting the PIXELS, nine accumulators each, an 18-barrier reduction through
// LDS at the end and one block per CU - 62 us per launch against a 22 us MFMA bound).  Operand rows / columns 24..31 of the 32-wide MFMA
// carry whatever the neighbouring LDS words hold: they only reach accumulator rows / columns >= 24, which are never read.
template <int TW, int PREC>
__global__ void __launch_bounds__(192) conv3x3_grouped_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, GcGeom g,
                                                                    const float* __restrict__ in_coef = nullptr) {
    typedef Tile<TW> T;
    constexpr int NT = 192;
    constexpr int NVP = (T::NPIX * 6 + NT - 1) / NT, ND = (128 * 6 + NT - 1) / NT;     // float4 slots per thread: patch, dY (6 per pixel)
    __shared__ float lds[T::NPIX * PP + 128 * PP + 8];
    __shared__ float cf[2 * CG];
    float* patch = lds;
    float* dyt = lds + T::NPIX * PP;               // [pixel][co]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int grp = blockIdx.x / g.nb, sub = blockIdx.x - grp * g.nb, coff = grp * CG;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 pre[NVP], dpre[ND];
    unsigned okm = 0;              // validity of the pre[] slots (in_coef, see bnrelu4)
    auto fetch = [&](int t) {      // unconditional loads from clamped addresses; the zero padding is applied to the VALUE
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        okm = 0;
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const int ph = pix / T::PW, pw = pix - ph * T::PW, h = h0 - 1 + ph, w = w0 - 1 + pw;
            const bool ok = pix < T::NPIX && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
            const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = w < 0 ? 0 : (w >= g.W ? g.W - 1 : w);
            const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * g.H + hc) * g.W + wc) * g.C + coff + c);
            pre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            okm |= ok ? 1u << p : 0u;
        }
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const int h = h0 + pix / TW, w = w0 + pix % TW;
            const bool ok = pix < 128 && h < g.H && w < g.W;
            const int hc = h >= g.H ? g.H - 1 : h, wc = w >= g.W ? g.W - 1 : w;
            const float4 v = *reinterpret_cast<const float4*>(dy + (((long)b * g.H + hc) * g.W + wc) * g.C + coff + c);
            dpre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int tile = sub;
    if (tile < g.ntiles) fetch(tile);
    stage_group_coef(cf, in_coef, coff, g.C);
    for (; tile < g.ntiles; tile += g.nb) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const float4 v = in_coef ? bnrelu4(pre[p], cf, c, (okm >> p) & 1u) : pre[p];
            if (pix < T::NPIX) { float* q = patch + pix * PP + c; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
        }
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            if (pix < 128) { float* q = dyt + pix * PP + c; q[0] = dpre[p].x; q[1] = dpre[p].y; q[2] = dpre[p].z; q[3] = dpre[p].w; }
        }
        __syncthreads();
        if (tile + g.nb < g.ntiles) fetch(tile + g.nb);
        if constexpr (PREC != 0) { // bf16 MFMA (PREC 1: rounded operands, 2: bf16x3 split): the tile's 128 pixels = eight 16-deep K groups; lane half hi owns pixels 16 q + 8 hi .. + 7
#pragma unroll 2
            for (int q = 0; q < 8; ++q) {
                float a[8], b[3][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pi = 16 * q + 8 * hi + j, prow = pi / TW, pcol = pi % TW;
                    a[j] = dyt[pi * PP + l31];
                    const float* pb = patch + ((prow + wave) * T::PW + pcol) * PP + l31;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) b[kw][j] = pb[kw * PP];
                }
                if constexpr (PREC == 2) {      // the dY fragment is split once and shared by the wave's 3 taps
                    const Bf16x3 fa = split_bf16x3(a);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) mfma_x3_presplit(fa, split_bf16x3(b[kw]), acc[kw]);
                } else {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) mfma_32x32x16_lp(a, b[kw], acc[kw], g.f16 ? 3 : 1);
                }
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < 64; ++kk) {      // k = pixel 2 kk + hi of the tile
                const int pi = 2 * kk + hi, prow = pi / TW, pcol = pi % TW;
                const float a = dyt[pi * PP + l31];
                const float* pb = patch + ((prow + wave) * T::PW + pcol) * PP + l31;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) mfma_32x32x2(a, pb[kw * PP], acc[kw]);
            }
        }
    }
    // the wave's three taps of the block's partial panel: rows co < 24 only (accumulator elements 12..15 are rows 24..31)
    float* pp = part + ((long)blockIdx.x * 9 + 3 * wave) * 1024 + l31;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
            pp[kw * 1024 + i * 32] = acc[kw][e];
        }
}

// ---- round 6: the fp32 weight gradient with the NINE TAPS IN THE MFMA's N dimension.  dW of a group is the (24 x 216) product dY^T . im2col(X):
// im2col column c = tap * 24 + ci, and c is also the offset of dW[co][tap][ci] inside the row of output channel co - so the group's gradient is ONE
// row-major (24 x 216) matrix.  SEVEN waves per block, wave j owns the columns [32 j, 32 j + 32): 216 of 224 MFMA columns carry work (the kernel
// above multiplied 3 x 3 blocks of 24 x 24 inside 32 x 32 tiles: 56 % of the MFMA; this one 24 / 32 x 216 / 224 = 72 %, i.e. 448 instead of 576 MFMAs
// per 128-pixel tile).  All waves walk the tile's 128 pixels as the K dimension: A[i = co][k = pixel] is the staged dY tile (shared by the seven waves),
// B[k][j = column] is gathered from the staged patch - a lane's column fixes its tap and input channel, i.e. one constant LDS offset, and every
// k-step adds a compile-time pixel offset.  One accumulator (16 registers) per wave, ~40 VGPRs, 31 KB of LDS: four blocks = 28 waves per CU.
// The block's (24 x 216) panel is ADDED to dW with fp32 atomics (64 lanes = two rows x 32 consecutive floats per instruction): no partial-panel
// workspace (36 KB written and read back per block before: 3.45x the algorithmic traffic), no reduce launch.  Like the k-split GEMM weight
// gradients the sum order over a group's blocks is not fixed (run-to-run differences at fp32 round-off); accumulate = 0 zero-fills dW first.
// S2: the stride-2 convolution of a stage's first block - tiles over the OUTPUT grid (Ho x Wo, g.H / g.W = the input extent), the staged patch is Tile2's
// (2 TH + 1) x (2 TW + 1) input pixels and output pixel (i, j), tap (kh, kw) reads patch pixel (2 i + kh, 2 j + kw): every pixel offset below doubles.
template <int TW, bool S2 = false>
__global__ void __launch_bounds__(448, S2 ? 3 : 6) conv3x3_grouped_wgrad7_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, GcGeom g,
                                                                              const float* __restrict__ in_coef, int Ho, int Wo) {
    typedef typename std::conditional<S2, Tile2<TW>, Tile<TW> >::type T;
    constexpr int SS = S2 ? 2 : 1;
    constexpr int NT = 448;
    constexpr int NVP = (T::NPIX * 6 + NT - 1) / NT, ND = (128 * 6 + NT - 1) / NT;     // float4 slots per thread: patch, dY (6 per pixel)
    __shared__ float lds[T::NPIX * PP + 128 * PP + 8];
    __shared__ float cf[2 * CG];
    float* patch = lds;
    float* dyt = lds + T::NPIX * PP;               // [pixel][co]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int grp = blockIdx.x / g.nb, sub = blockIdx.x - grp * g.nb, coff = grp * CG;
    // this lane's im2col column: tap (kh, kw) and input channel; the eight columns 216..223 of wave 6 read tap 8 again (in-bounds) and are never stored
    const int col = 32 * wave + l31, tapc = col / CG < 9 ? col / CG : 8, ci = col - (col / CG) * CG;
    const int boff = ((tapc / 3) * T::PW + (tapc % 3)) * PP + ci + hi * SS * PP;       // + the k-step's pixel (2 kk + hi): hi moves one output pixel to the right (TW is even)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 pre[NVP], dpre[ND];
    unsigned okm = 0;              // validity of the pre[] slots (in_coef, see bnrelu4)
    auto fetch = [&](int t) {      // unconditional loads from clamped addresses; the zero padding is applied to the VALUE
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        const int tq = opaque(tid);
        okm = 0;
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int s = tq + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const int ph = pix / T::PW, pw = pix - ph * T::PW, h = SS * h0 - 1 + ph, w = SS * w0 - 1 + pw;
            const bool ok = pix < T::NPIX && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
            const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = w < 0 ? 0 : (w >= g.W ? g.W - 1 : w);
            const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * g.H + hc) * g.W + wc) * g.C + coff + c);
            pre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            okm |= ok ? 1u << p : 0u;
        }
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int s = tq + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const int h = h0 + pix / TW, w = w0 + pix % TW;
            const bool ok = pix < 128 && h < Ho && w < Wo;
            const int hc = h >= Ho ? Ho - 1 : h, wc = w >= Wo ? Wo - 1 : w;
            const float4 v = *reinterpret_cast<const float4*>(dy + (((long)b * Ho + hc) * Wo + wc) * g.C + coff + c);
            dpre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // the block's tiles: a contiguous, balanced range of the group's tiles (neighbouring tiles share halo rows in L2)
    const int t0 = (int)((long)sub * g.ntiles / g.nb), t1 = (int)((long)(sub + 1) * g.ntiles / g.nb);
    int tile = t0;
    if (tile < t1) fetch(tile);
    stage_group_coef(cf, in_coef, coff, g.C);
    for (; tile < t1; ++tile) {
        __syncthreads();
        const int ts = opaque(tid);
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int s = ts + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const float4 v = in_coef ? bnrelu4(pre[p], cf, c, (okm >> p) & 1u) : pre[p];
            if (pix < T::NPIX) { float* q = patch + pix * PP + c; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
        }
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int s = ts + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            if (pix < 128) { float* q = dyt + pix * PP + c; q[0] = dpre[p].x; q[1] = dpre[p].y; q[2] = dpre[p].z; q[3] = dpre[p].w; }
        }
        __syncthreads();
        if (tile + 1 < t1) fetch(tile + 1);
        const float* pa = dyt + hi * PP + l31;
        const float* pb = patch + boff;
        // the tile's 64 k-steps (k = pixel 2 kk + hi) in eight chunks of eight: the NEXT chunk's sixteen operand reads are issued before the current
        // chunk's eight MFMAs, so a wave never sits out an LDS round trip between two MFMAs (one accumulator chain per wave: nothing else to issue)
        auto kofs_a = [](int kk) { return 2 * kk * PP; };
        auto kofs_b = [](int kk) { return (((2 * kk) / TW) * T::PW + (2 * kk) % TW) * SS * PP; };
        float a0[8], b0[8], a1[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0[u] = pa[kofs_a(u)]; b0[u] = pb[kofs_b(u)]; }
#pragma unroll
        for (int ch = 0; ch < 8; ch += 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a1[u] = pa[kofs_a(8 * ch + 8 + u)]; b1[u] = pb[kofs_b(8 * ch + 8 + u)]; }
            TF_SCHED_FENCE();           // (hipcc's scheduler otherwise sinks every read to just in front of its MFMA: read, wait, two MFMAs, read, wait, ...)
#pragma unroll
            for (int u = 0; u < 8; ++u) mfma_32x32x2(a0[u], b0[u], acc);
            TF_SCHED_FENCE();
            if (ch + 2 < 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a0[u] = pa[kofs_a(8 * ch + 16 + u)]; b0[u] = pb[kofs_b(8 * ch + 16 + u)]; }
            }
            TF_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u) mfma_32x32x2(a1[u], b1[u], acc);
            TF_SCHED_FENCE();
        }
    }
    // rows co < 24 only (accumulator elements 12..15 are rows 24..31); columns < 216
    if (col < 9 * CG && t0 < t1) {
        float* d = dw + (long)grp * CG * 9 * CG + col;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
            atomicAdd(d + i * 9 * CG, acc[e]);
        }
    }
}

__global__ void __launch_bounds__(256) conv3x3_grouped_wgrad_reduce_kernel(const float* __restrict__ part, int nb, float* __restrict__ dw, int accumulate) {
    const int grp = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;   // (tap, co, ci) over 9 x 32 x 32
    if (i >= 9 * 1024) return;
    const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
    if (co >= CG || ci >= CG) return;
    float s = 0.f;
    for (int b0 = 0; b0 < nb; b0 += 8) {                 // 8 partial panels per trip (clamped index, out-of-range ones dropped after the load)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((long)(grp * nb + (b0 + u < nb ? b0 + u : nb - 1))) * 9216 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += b0 + u < nb ? v[u] : 0.f;
    }
    float* d = dw + (((long)grp * CG + co) * 9 + tap) * CG + ci;
    *d = accumulate ? *d + s : s;
}

// ---- input gradient of the STRIDE-2 grouped 3x3 convolution (first block of every RegNetY stage) by sub-pixel decomposition.
// dX[2i+a][2j+b] only receives the taps kh = a+1 (mod 2), kw = b+1 (mod 2): one tap for (a, b) = (0, 0), two for (0, 1) / (1, 0), four for (1, 1) -
// nine tap products per dY pixel in all, each reading dY at (i + dr, j + dc), dr, dc in {0, 1}.  The implicit-GEMM path gathers nine MASKED taps for
// every pixel of the 4x larger dX grid (6-12 TFLOP/s measured).  Here a wave owns 32 dY pixels and four accumulators (one per parity class);
// the (TH+1) x (TW+1) dY patch and the group's weight panel are staged as in conv3x3_grouped_kernel.  Tiles run over the dY grid.
template <int TW>
__global__ void __launch_bounds__(256, 2) conv3x3_grouped_s2_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, GcGeom g,
                                                                          int Hi, int Wi, int accumulate, int prec) {
    constexpr int RW = 32 / TW, TH = 4 * RW, PH = TH + 1, PW = TW + 1, NPIX = PH * PW;
    constexpr int NV = (NPIX * 6 + 255) / 256;
    __shared__ float patch[NPIX * PP];
    __shared__ float wl[9 * CG][WP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int grp = blockIdx.x / g.nb, sub = blockIdx.x - grp * g.nb, coff = grp * CG;
    GroupWeightRegs wreg;
    issue_group_weights(wreg, w + (long)grp * CG * 9 * CG);
    float4 pre[NV];
    auto fetch = [&](int t) {      // dY patch rows h0 .. h0 + TH, columns w0 .. w0 + TW (zero outside the dY map); g.H / g.W = the dY extent
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * TH, w0 = (r % g.tiles_w) * TW;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int s = tid + p * 256, pix = s / 6, c = (s - pix * 6) * 4;
            const int ph = pix / PW, pw = pix - ph * PW, h = h0 + ph, ww = w0 + pw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < NPIX && h < g.H && ww < g.W) v = *reinterpret_cast<const float4*>(dy + (((long)b * g.H + h) * g.W + ww) * g.C + coff + c);
            pre[p] = v;
        }
    };
    int tile = sub;
    if (tile < g.ntiles) fetch(tile);
    scatter_group_weights(wl, wreg, 1);      // wl[tap'][co][ci] = W[co][8 - tap'][ci]: tap (kh, kw) sits at tap' = 8 - (3 kh + kw)
    const int prow = wave * RW + l31 / TW, pcol = l31 % TW;
    for (; tile < g.ntiles; tile += g.nb) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int s = tid + p * 256, pix = s / 6, c = (s - pix * 6) * 4;
            if (pix < NPIX) { float* q = patch + pix * PP + c; q[0] = pre[p].x; q[1] = pre[p].y; q[2] = pre[p].z; q[3] = pre[p].w; }
        }
        __syncthreads();
        if (tile + g.nb < g.ntiles) fetch(tile + g.nb);
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int a = (kh + 1) & 1, b2 = (kw + 1) & 1;            // parity class fed by this tap
                const int dr = (a + 1 - kh) / 2, dc = (b2 + 1 - kw) / 2;    // dY offset: (2i + a + 1 - kh) / 2 = i + dr
                if (prec == 1 || prec == 3) {      // bf16 / fp16 compute modes: operands rounded in registers like conv3x3_grouped_kernel (two 16-deep groups over 24 -> 32 channels)
                    const float* pa = patch + ((prow + dr) * PW + pcol + dc) * PP;
                    const int tp = (8 - (3 * kh + kw)) * CG;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float av[8], bv[8];
                        const bool live = (q == 0) || (hi == 0);
                        const int k0 = 16 * q + 8 * hi;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            av[j] = live ? pa[k0 + j] : 0.f;
                            bv[j] = live ? wl[tp + k0 + j][l31] : 0.f;
                        }
                        mfma_32x32x16_lp(av, bv, acc[a][b2], prec);
                    }
                } else {
                    const float* pa = patch + ((prow + dr) * PW + pcol + dc) * PP + hi;
                    const float* pb = &wl[(8 - (3 * kh + kw)) * CG + hi][l31];
#pragma unroll
                    for (int kk = 0; kk < CG / 2; ++kk) mfma_32x32x2(pa[2 * kk], pb[2 * kk * WP], acc[a][b2]);
                }
            }
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * TH, w0 = (r % g.tiles_w) * TW;
        if (l31 < CG) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;           // dY pixel inside the wave's 32
                        const int hh = 2 * (h0 + wave * RW + i / TW) + a, ww = 2 * (w0 + i % TW) + b2;
                        if (hh < Hi && ww < Wi) {
                            float* dst = dx + (((long)b * Hi + hh) * Wi + ww) * g.C + coff + l31;
                            *dst = accumulate ? *dst + acc[a][b2][e] : acc[a][b2][e];
                        }
                    }
        }
    }
}

// ---- the STRIDE-2 grouped 3x3 convolution (first block of every RegNetY stage) as direct kernels: forward (+ output statistics, + the producer's
// BatchNorm apply folded in like bnrelu4) and weight gradient.  Tiles run over the OUTPUT grid (128 output pixels), the staged input patch is
// (2 TH + 1) x (2 TW + 1) pixels (56 KB; with the weight panel 87 KB of LDS: one block per CU, the next patch travels in registers meanwhile);
// output pixel (i, j), tap (kh, kw) reads patch pixel (2 i + kh, 2 j + kw).  Through the implicit-GEMM engine these 16 launches per step ran at
// 6-34 TFLOP/s (profiles/r04_census_fp32_final.txt).  OPT-IN (ops: TF_GROUPED_S2=1) until measured on the MI355X.

template <int TW, bool STAT>
__global__ void __launch_bounds__(256, 1) conv3x3_grouped_s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, GcGeom g, int Ho,
                                                                        int Wo, int prec, float* __restrict__ stat, const float* __restrict__ in_coef) {
    typedef Tile2<TW> T;
    constexpr int NV = (T::NPIX * 6 + 255) / 256;
    __shared__ float patch[T::NPIX * PP];
    __shared__ float wl[9 * CG][WP];
    __shared__ float cf[2 * CG];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int grp = blockIdx.x / g.nb, sub = blockIdx.x - grp * g.nb, coff = grp * CG;
    GroupWeightRegs wreg;
    issue_group_weights(wreg, w + (long)grp * CG * 9 * CG);
    float4 pre[NV];
    unsigned okm = 0;
    auto fetch = [&](int t) {      // g.H / g.W = the INPUT extent, tiles over the output grid
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        okm = 0;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int s = tid + p * 256, pix = s / 6, c = (s - pix * 6) * 4;
            const int ph = pix / T::PW, pw = pix - ph * T::PW, h = 2 * h0 - 1 + ph, ww = 2 * w0 - 1 + pw;
            const bool ok = pix < T::NPIX && (unsigned)h < (unsigned)g.H && (unsigned)ww < (unsigned)g.W;
            const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = ww < 0 ? 0 : (ww >= g.W ? g.W - 1 : ww);
            const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * g.H + hc) * g.W + wc) * g.C + coff + c);
            pre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            okm |= ok ? 1u << p : 0u;
        }
    };
    int tile = sub;
    if (tile < g.ntiles) fetch(tile);
    stage_group_coef(cf, in_coef, coff, g.C);
    scatter_group_weights(wl, wreg, 0);
    const int prow = wave * T::RW + l31 / TW, pcol = l31 % TW;
    float st_n = 0.f, st_mean = 0.f, st_m2 = 0.f;
    for (; tile < g.ntiles; tile += g.nb) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int s = tid + p * 256, pix = s / 6, c = (s - pix * 6) * 4;
            const float4 v = in_coef ? bnrelu4(pre[p], cf, c, (okm >> p) & 1u) : pre[p];
            if (pix < T::NPIX) { float* q = patch + pix * PP + c; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
        }
        __syncthreads();
        if (tile + g.nb < g.ntiles) fetch(tile + g.nb);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (prec == 1 || prec == 3) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const float* pa = patch + ((2 * prow + kh) * T::PW + 2 * pcol + kw) * PP;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float a[8], b[8];
                    const bool live = (q == 0) || (hi == 0);
                    const int k0 = 16 * q + 8 * hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        a[j] = live ? pa[k0 + j] : 0.f;
                        b[j] = live ? wl[tap * CG + k0 + j][l31] : 0.f;
                    }
                    mfma_32x32x16_lp(a, b, acc, prec);
                }
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const float* pa = patch + ((2 * prow + kh) * T::PW + 2 * pcol + kw) * PP + hi;
                const float* pb = &wl[tap * CG + hi][l31];
#pragma unroll
                for (int kk = 0; kk < CG / 2; ++kk) mfma_32x32x2(pa[2 * kk], pb[2 * kk * WP], acc);
            }
        }
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        if (l31 < CG) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
                const int hh = h0 + wave * T::RW + i / TW, ww = w0 + i % TW;
                if (hh < Ho && ww < Wo) y[(((long)b * Ho + hh) * Wo + ww) * g.C + coff + l31] = acc[e];
            }
        }
        if constexpr (STAT) {       // as conv3x3_grouped_kernel: a running Welford triple per (wave, channel), merged per block at the end
            float s = 0.f, cnt = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
                const bool ok = (h0 + wave * T::RW + i / TW) < Ho && (w0 + i % TW) < Wo;
                s += ok ? acc[e] : 0.f;
                cnt += ok ? 1.f : 0.f;
            }
            s += shfl_xor(s, 32);
            cnt += shfl_xor(cnt, 32);
            if (cnt > 0.f) {
                const float m_t = s / cnt;
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    const bool ok = (h0 + wave * T::RW + i / TW) < Ho && (w0 + i % TW) < Wo;
                    const float d = acc[e] - m_t;
                    q += ok ? d * d : 0.f;
                }
                q += shfl_xor(q, 32);
                const float n_new = st_n + cnt, delta = m_t - st_mean;
                st_mean += delta * (cnt / n_new);
                st_m2 += q + delta * delta * (st_n * cnt / n_new);
                st_n = n_new;
            }
        }
    }
    if constexpr (STAT) {
        __syncthreads();
        float* red = patch;                                              // [wave][3][32]
        if (hi == 0) { red[(wave * 3 + 0) * 32 + l31] = st_n; red[(wave * 3 + 1) * 32 + l31] = st_mean; red[(wave * 3 + 2) * 32 + l31] = st_m2; }
        __syncthreads();
        if (tid < CG) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const float nb_ = red[(wv * 3 + 0) * 32 + tid], mb = red[(wv * 3 + 1) * 32 + tid], qb = red[(wv * 3 + 2) * 32 + tid];
                if (nb_ > 0.f) {
                    const float n_new = n + nb_, delta = mb - mean;
                    mean += delta * (nb_ / n_new);
                    m2 += qb + delta * delta * (n * nb_ / n_new);
                    n = n_new;
                }
            }
            float* o = stat + (long)sub * 3 * g.C + coff + tid;
            o[0] = n; o[g.C] = mean; o[2 * (long)g.C] = m2;
        }
    }
}

// weight gradient of the stride-2 convolution: conv3x3_grouped_wgrad_kernel with the input patch of Tile2 (wave kh owns the taps (kh, 0..2), K = the
// tile's 128 OUTPUT pixels, B operand = patch pixel (2 i + kh, 2 j + kw)); partial panels reduced by conv3x3_grouped_wgrad_reduce_kernel
template <int TW, int PREC>
__global__ void __launch_bounds__(192) conv3x3_grouped_s2_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, GcGeom g,
                                                                       int Ho, int Wo, const float* __restrict__ in_coef) {
    typedef Tile2<TW> T;
    constexpr int NT = 192;
    constexpr int NVP = (T::NPIX * 6 + NT - 1) / NT, ND = (128 * 6 + NT - 1) / NT;
    __shared__ float lds[T::NPIX * PP + 128 * PP + 8];
    __shared__ float cf[2 * CG];
    float* patch = lds;
    float* dyt = lds + T::NPIX * PP;               // [output pixel][co]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int grp = blockIdx.x / g.nb, sub = blockIdx.x - grp * g.nb, coff = grp * CG;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 pre[NVP], dpre[ND];
    unsigned okm = 0;
    auto fetch = [&](int t) {
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * T::TH, w0 = (r % g.tiles_w) * TW;
        okm = 0;
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const int ph = pix / T::PW, pw = pix - ph * T::PW, h = 2 * h0 - 1 + ph, w = 2 * w0 - 1 + pw;
            const bool ok = pix < T::NPIX && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
            const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = w < 0 ? 0 : (w >= g.W ? g.W - 1 : w);
            const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * g.H + hc) * g.W + wc) * g.C + coff + c);
            pre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            okm |= ok ? 1u << p : 0u;
        }
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const int h = h0 + pix / TW, w = w0 + pix % TW;
            const bool ok = pix < 128 && h < Ho && w < Wo;
            const int hc = h >= Ho ? Ho - 1 : h, wc = w >= Wo ? Wo - 1 : w;
            const float4 v = *reinterpret_cast<const float4*>(dy + (((long)b * Ho + hc) * Wo + wc) * g.C + coff + c);
            dpre[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int tile = sub;
    if (tile < g.ntiles) fetch(tile);
    stage_group_coef(cf, in_coef, coff, g.C);
    for (; tile < g.ntiles; tile += g.nb) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            const float4 v = in_coef ? bnrelu4(pre[p], cf, c, (okm >> p) & 1u) : pre[p];
            if (pix < T::NPIX) { float* q = patch + pix * PP + c; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
        }
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int s = tid + p * NT, pix = s / 6, c = (s - pix * 6) * 4;
            if (pix < 128) { float* q = dyt + pix * PP + c; q[0] = dpre[p].x; q[1] = dpre[p].y; q[2] = dpre[p].z; q[3] = dpre[p].w; }
        }
        __syncthreads();
        if (tile + g.nb < g.ntiles) fetch(tile + g.nb);
        if constexpr (PREC != 0) {
#pragma unroll 2
            for (int q = 0; q < 8; ++q) {
                float a[8], b[3][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pi = 16 * q + 8 * hi + j, prow = pi / TW, pcol = pi % TW;
                    a[j] = dyt[pi * PP + l31];
                    const float* pb = patch + ((2 * prow + wave) * T::PW + 2 * pcol) * PP + l31;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) b[kw][j] = pb[kw * PP];
                }
                if constexpr (PREC == 2) {
                    const Bf16x3 fa = split_bf16x3(a);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) mfma_x3_presplit(fa, split_bf16x3(b[kw]), acc[kw]);
                } else {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) mfma_32x32x16_lp(a, b[kw], acc[kw], g.f16 ? 3 : 1);
                }
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < 64; ++kk) {
                const int pi = 2 * kk + hi, prow = pi / TW, pcol = pi % TW;
                const float a = dyt[pi * PP + l31];
                const float* pb = patch + ((2 * prow + wave) * T::PW + 2 * pcol) * PP + l31;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) mfma_32x32x2(a, pb[kw * PP], acc[kw]);
            }
        }
    }
    float* pp = part + ((long)blockIdx.x * 9 + 3 * wave) * 1024 + l31;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
            pp[kw * 1024 + i * 32] = acc[kw][e];
        }
}

__global__ void __launch_bounds__(256) fill_f32_kernel(float* __restrict__ p, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i + k < n) p[i + k] = 0.f;
}

constexpr int kMaxBlocks = 768;     // persistent grid: up to 3 blocks per CU

// compute precision of these kernels; TF_X3_DIRECT=0 keeps them on the exact fp32 MFMA in f32x3 mode (A/B switch)
inline int direct_prec() {
    static const bool x3 = [] { const char* e = getenv("TF_X3_DIRECT"); return !e || atoi(e) != 0; }();
    const int p = tf::gemm_precision();
    return (p == 2 && !x3) ? 0 : p;
}

// f32x3 mode: the forward / input-gradient kernels hold ONE 32x32 tile per wave, where the in-register split is VALU-bound and measured slower
// than the exact fp32 MFMA (46 vs 41 us at 16x44x576, 78 vs 69 us at 64x176x72); only the weight gradient (a dY fragment shared by 9 taps)
// gains (43 vs 60, 86 vs 115 us).  TF_X3_GROUPED_FWD=1 re-enables the X3 instantiation for A/B runs.
inline int fwd_prec() {
    static const bool x3 = [] { const char* e = getenv("TF_X3_GROUPED_FWD"); return e && atoi(e) != 0; }();
    const int p = direct_prec();
    return (p == 2 && !x3) ? 0 : p;
}

inline bool f32t_on() {        // A/B switch: TF_GROUPED_F32T=0 keeps the fp32 forward / input gradient on the pixel-major accumulator (16 scalar stores per lane)
    static const bool on = [] { const char* e = getenv("TF_GROUPED_F32T"); return !e || atoi(e) != 0; }();
    return on;
}
inline GcGeom make_geom(int B, int H, int W, int C, int TW, int max_blocks = kMaxBlocks) {
    GcGeom g; g.B = B; g.H = H; g.W = W; g.C = C; g.G = C / CG; g.f16 = tf::gemm_precision() == 3 ? 1 : 0;
    static const int dbg = [] { const char* e = getenv("TF_GC_DBG"); return e ? atoi(e) : 0; }();      // timing diagnosis only (tools/grouped_lab.py): phases of the forward kernel switched off, results are garbage
    g.dbg = dbg;
    const int th = TW == 16 ? 8 : 4;
    g.tiles_h = cdiv(H, th); g.tiles_w = cdiv(W, TW); g.ntiles = B * g.tiles_h * g.tiles_w;
    int nb = max_blocks / g.G;
    if (nb < 1) nb = 1;
    if (nb > g.ntiles) nb = g.ntiles;
    g.nb = nb;
    return g;
}
// tile width with the least padding (ties -> 32: longer contiguous rows)
inline int pick_tw(int H, int W) {
    const long p16 = (long)cdiv(H, 8) * 8 * cdiv(W, 16) * 16, p32 = (long)cdiv(H, 4) * 4 * cdiv(W, 32) * 32;
    return p16 < p32 ? 16 : 32;
}
inline bool args_ok(const void* a, const void* b, const void* c, int B, int H, int W, int C) {
    return a && b && c && B > 0 && H > 0 && W > 0 && C > 0 && C % CG == 0 && aligned16(a) && aligned16(c);
}

}  // namespace

extern "C" int tf_conv3x3_grouped_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int relu, void* stream) {
    TF_REQUIRE(args_ok(x, w, y, B, H, W, C), "tf_conv3x3_grouped_fwd_f32: needs NHWC tensors with C %% 24 == 0 (group width 24), 16-byte aligned");
    const int tw = pick_tw(H, W);
    GcGeom g = make_geom(B, H, W, C, tw);
    const int prec = fwd_prec();
    if (prec == 2) {
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, bias, y, g, 0, relu, 0, 2, (float*)nullptr);
        else TF_LAUNCH((conv3x3_grouped_kernel<32, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, bias, y, g, 0, relu, 0, 2, (float*)nullptr);
    } else if (prec == 0 && f32t_on()) {        // exact fp32: the transposed-accumulator instantiation (16-byte stores)
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, false, false, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, bias, y, g, 0, relu, 0, 0, (float*)nullptr);
        else TF_LAUNCH((conv3x3_grouped_kernel<32, false, false, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, bias, y, g, 0, relu, 0, 0, (float*)nullptr);
    } else if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, false>), dim3(g.G * g.nb), dim3(256), stream, x, w, bias, y, g, 0, relu, 0, prec, (float*)nullptr);
    else TF_LAUNCH((conv3x3_grouped_kernel<32, false>), dim3(g.G * g.nb), dim3(256), stream, x, w, bias, y, g, 0, relu, 0, prec, (float*)nullptr);
    return launch_status("tf_conv3x3_grouped_fwd_f32");
}

extern "C" int tf_conv3x3_grouped_colstat_parts(void) { return kMaxBlocks; }

static int grouped_fwd_colstat(const char* what, const float* x, const float* in_coef, const float* w, float* y, int B, int H, int W, int C, float* colstat,
                               int* colstat_nparts, void* stream) {
    const int tw = pick_tw(H, W);
    GcGeom g = make_geom(B, H, W, C, tw);
    const int prec = fwd_prec();
    const float* nob = nullptr;
    *colstat_nparts = g.nb;
    if (prec == 2) {
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, true, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, nob, y, g, 0, 0, 0, 2, colstat, in_coef);
        else TF_LAUNCH((conv3x3_grouped_kernel<32, true, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, nob, y, g, 0, 0, 0, 2, colstat, in_coef);
    } else if (prec == 0 && f32t_on()) {
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, false, true, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, nob, y, g, 0, 0, 0, 0, colstat, in_coef);
        else TF_LAUNCH((conv3x3_grouped_kernel<32, false, true, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, nob, y, g, 0, 0, 0, 0, colstat, in_coef);
    } else if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, false, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, nob, y, g, 0, 0, 0, prec, colstat, in_coef);
    else TF_LAUNCH((conv3x3_grouped_kernel<32, false, true>), dim3(g.G * g.nb), dim3(256), stream, x, w, nob, y, g, 0, 0, 0, prec, colstat, in_coef);
    return launch_status(what);
}
extern "C" int tf_conv3x3_grouped_fwd_colstat_f32(const float* x, const float* w, float* y, int B, int H, int W, int C, float* colstat, int* colstat_nparts,
                                                  void* stream) {
    TF_REQUIRE(args_ok(x, w, y, B, H, W, C) && colstat && colstat_nparts, "tf_conv3x3_grouped_fwd_colstat_f32: needs NHWC tensors with C %% 24 == 0, colstat and colstat_nparts");
    return grouped_fwd_colstat("tf_conv3x3_grouped_fwd_colstat_f32", x, nullptr, w, y, B, H, W, C, colstat, colstat_nparts, stream);
}
extern "C" int tf_conv3x3_grouped_bnrelu_fwd_colstat_f32(const float* x, const float* in_coef, const float* w, float* y, int B, int H, int W, int C, float* colstat,
                                                         int* colstat_nparts, void* stream) {
    TF_REQUIRE(args_ok(x, w, y, B, H, W, C) && in_coef && colstat && colstat_nparts,
               "tf_conv3x3_grouped_bnrelu_fwd_colstat_f32: needs NHWC tensors with C %% 24 == 0, in_coef = [scale | shift] (2 C), colstat and colstat_nparts");
    return grouped_fwd_colstat("tf_conv3x3_grouped_bnrelu_fwd_colstat_f32", x, in_coef, w, y, B, H, W, C, colstat, colstat_nparts, stream);
}

extern "C" int tf_conv3x3_grouped_dgrad_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int accumulate, void* stream) {
    TF_REQUIRE(args_ok(dy, w, dx, B, H, W, C), "tf_conv3x3_grouped_dgrad_f32: needs NHWC tensors with C %% 24 == 0 (group width 24), 16-byte aligned");
    const int tw = pick_tw(H, W);
    GcGeom g = make_geom(B, H, W, C, tw);
    const int prec = fwd_prec();
    const float* nob = nullptr;
    if (prec == 2) {
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, true>), dim3(g.G * g.nb), dim3(256), stream, dy, w, nob, dx, g, 1, 0, accumulate, 2, (float*)nullptr);
        else TF_LAUNCH((conv3x3_grouped_kernel<32, true>), dim3(g.G * g.nb), dim3(256), stream, dy, w, nob, dx, g, 1, 0, accumulate, 2, (float*)nullptr);
    } else if (prec == 0 && f32t_on()) {
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, false, false, true>), dim3(g.G * g.nb), dim3(256), stream, dy, w, nob, dx, g, 1, 0, accumulate, 0, (float*)nullptr);
        else TF_LAUNCH((conv3x3_grouped_kernel<32, false, false, true>), dim3(g.G * g.nb), dim3(256), stream, dy, w, nob, dx, g, 1, 0, accumulate, 0, (float*)nullptr);
    } else if (tw == 16) TF_LAUNCH((conv3x3_grouped_kernel<16, false>), dim3(g.G * g.nb), dim3(256), stream, dy, w, nob, dx, g, 1, 0, accumulate, prec, (float*)nullptr);
    else TF_LAUNCH((conv3x3_grouped_kernel<32, false>), dim3(g.G * g.nb), dim3(256), stream, dy, w, nob, dx, g, 1, 0, accumulate, prec, (float*)nullptr);
    return launch_status("tf_conv3x3_grouped_dgrad_f32");
}

constexpr int kWgradMaxBlocks = 1536;   // partial panels in the workspace (36 KB each)
extern "C" long tf_conv3x3_grouped_wgrad_ws_floats(void) { return (long)kWgradMaxBlocks * 9216; }

static int grouped_wgrad(const char* what, const float* dy, const float* x, const float* in_coef, float* dw, int B, int H, int W, int C, int accumulate, float* ws,
                         void* stream) {
    const int tw = pick_tw(H, W);
    // three-wave blocks, four to five per CU: ~700 blocks put two waves on every SIMD; a block takes at least two tiles when there are enough
    // (every block ends with a 27 KB partial panel that the reduce kernel reads back)
    GcGeom g = make_geom(B, H, W, C, tw, 1024);
    int tpb = cdiv(g.ntiles, g.nb);                     // tiles per block, balanced: every block takes tpb or tpb - 1 tiles
    if (tpb < 2 && g.ntiles >= 2) tpb = 2;
    g.nb = cdiv(g.ntiles, tpb);
    if ((long)g.G * g.nb > kWgradMaxBlocks) g.nb = kWgradMaxBlocks / g.G;
    TF_REQUIRE(g.nb >= 1, "%s: %d groups exceed the workspace", what, g.G);
    const int prec = direct_prec();
    static const int v7 = [] { const char* e = getenv("TF_GROUPED_WGRAD7"); return e ? atoi(e) : 1; }();     // A/B switch: 0 = the three-wave kernel + partial panels
    if (prec == 0 && v7) {
        // seven-wave blocks (conv3x3_grouped_wgrad7_kernel), 31 KB of LDS: up to four per CU.  Every block ends with 5184 atomics on its group's dW, so a
        // block takes at least two tiles when there are enough; ~3 blocks per CU otherwise (TF_GROUPED_WGRAD7 > 1: that many blocks in all, for the lab)
        GcGeom g7 = make_geom(B, H, W, C, tw, v7 > 1 ? v7 : 768);
        int tpb7 = cdiv(g7.ntiles, g7.nb);
        if (tpb7 < 2 && g7.ntiles >= 2) tpb7 = 2;
        g7.nb = cdiv(g7.ntiles, tpb7);
        if (!accumulate) TF_LAUNCH(fill_f32_kernel, dim3(cdiv((long)C * 9 * CG, 1024)), dim3(256), stream, dw, (long)C * 9 * CG);
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_wgrad7_kernel<16>), dim3(g7.G * g7.nb), dim3(448), stream, x, dy, dw, g7, in_coef, H, W);
        else TF_LAUNCH((conv3x3_grouped_wgrad7_kernel<32>), dim3(g7.G * g7.nb), dim3(448), stream, x, dy, dw, g7, in_coef, H, W);
        return launch_status(what);
    }
#define TF_GW(TW_, P_) TF_LAUNCH((conv3x3_grouped_wgrad_kernel<TW_, P_>), dim3(g.G * g.nb), dim3(192), stream, x, dy, ws, g, in_coef)
    if (tw == 16) { if (prec == 2) TF_GW(16, 2); else if (prec == 1 || prec == 3) TF_GW(16, 1); else TF_GW(16, 0); }
    else { if (prec == 2) TF_GW(32, 2); else if (prec == 1 || prec == 3) TF_GW(32, 1); else TF_GW(32, 0); }
#undef TF_GW
    TF_LAUNCH(conv3x3_grouped_wgrad_reduce_kernel, dim3(36, g.G), dim3(256), stream, (const float*)ws, g.nb, dw, accumulate);
    return launch_status(what);
}
extern "C" int tf_conv3x3_grouped_wgrad_f32(const float* dy, const float* x, float* dw, int B, int H, int W, int C, int accumulate, float* ws, void* stream) {
    TF_REQUIRE(args_ok(dy, x, dw, B, H, W, C) && ws && aligned16(dy), "tf_conv3x3_grouped_wgrad_f32: needs C %% 24 == 0 and ws of tf_conv3x3_grouped_wgrad_ws_floats() floats");
    return grouped_wgrad("tf_conv3x3_grouped_wgrad_f32", dy, x, nullptr, dw, B, H, W, C, accumulate, ws, stream);
}
// the weight gradient against max(x sc + sh, 0) of a raw producer output x (BatchNorm apply folded into the consumer, see bnrelu4)
extern "C" int tf_conv3x3_grouped_bnrelu_wgrad_f32(const float* dy, const float* x, const float* in_coef, float* dw, int B, int H, int W, int C, int accumulate,
                                                   float* ws, void* stream) {
    TF_REQUIRE(args_ok(dy, x, dw, B, H, W, C) && in_coef && ws && aligned16(dy),
               "tf_conv3x3_grouped_bnrelu_wgrad_f32: needs C %% 24 == 0, in_coef = [scale | shift] (2 C) and ws of tf_conv3x3_grouped_wgrad_ws_floats() floats");
    return grouped_wgrad("tf_conv3x3_grouped_bnrelu_wgrad_f32", dy, x, in_coef, dw, B, H, W, C, accumulate, ws, stream);
}

// stride-2 forward (+ output statistics when colstat != NULL, + the producer's BatchNorm apply when in_coef != NULL) and weight gradient; Hi x Wi = input extent
extern "C" int tf_conv3x3_grouped_s2_fwd_f32(const float* x, const float* in_coef, const float* w, float* y, int B, int Hi, int Wi, int C, float* colstat,
                                             int* colstat_nparts, void* stream) {
    TF_REQUIRE(args_ok(x, w, y, B, Hi, Wi, C) && (!colstat || colstat_nparts), "tf_conv3x3_grouped_s2_fwd_f32: needs NHWC tensors with C %% 24 == 0 (group width 24), 16-byte aligned");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const int tw = pick_tw(Ho, Wo);
    GcGeom g = make_geom(B, Ho, Wo, C, tw, 256);            // one 87 KB block per CU
    g.H = Hi; g.W = Wi;
    int prec = tf::gemm_precision();
    if (prec == 2) prec = 0;                                 // f32x3: the exact fp32 MFMA (at least as accurate)
    if (colstat_nparts) *colstat_nparts = colstat ? g.nb : 0;
#define TF_S2F(TW_, ST_) TF_LAUNCH((conv3x3_grouped_s2_fwd_kernel<TW_, ST_>), dim3(g.G * g.nb), dim3(256), stream, x, w, y, g, Ho, Wo, prec, colstat, in_coef)
    if (tw == 16) { if (colstat) TF_S2F(16, true); else TF_S2F(16, false); }
    else { if (colstat) TF_S2F(32, true); else TF_S2F(32, false); }
#undef TF_S2F
    return launch_status("tf_conv3x3_grouped_s2_fwd_f32");
}
extern "C" int tf_conv3x3_grouped_s2_wgrad_f32(const float* dy, const float* x, const float* in_coef, float* dw, int B, int Hi, int Wi, int C, int accumulate, float* ws,
                                               void* stream) {
    TF_REQUIRE(args_ok(dy, x, dw, B, Hi, Wi, C) && ws && aligned16(dy), "tf_conv3x3_grouped_s2_wgrad_f32: needs C %% 24 == 0 and ws of tf_conv3x3_grouped_wgrad_ws_floats() floats");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const int tw = pick_tw(Ho, Wo);
    GcGeom g = make_geom(B, Ho, Wo, C, tw, 512);            // 69 KB of LDS: two blocks per CU
    int tpb = cdiv(g.ntiles, g.nb);
    if (tpb < 2 && g.ntiles >= 2) tpb = 2;
    g.nb = cdiv(g.ntiles, tpb);
    if ((long)g.G * g.nb > kWgradMaxBlocks) g.nb = kWgradMaxBlocks / g.G;
    TF_REQUIRE(g.nb >= 1, "tf_conv3x3_grouped_s2_wgrad_f32: %d groups exceed the workspace", g.G);
    g.H = Hi; g.W = Wi;
    const int prec = direct_prec();
    static const int v7 = [] { const char* e = getenv("TF_GROUPED_WGRAD7"); return e ? atoi(e) : 1; }();
    if (prec == 0 && v7) {      // seven-wave form (69 - 72 KB of LDS: two blocks per CU), atomically accumulated: see grouped_wgrad
        GcGeom g7 = make_geom(B, Ho, Wo, C, tw, 512);
        int tpb7 = cdiv(g7.ntiles, g7.nb);
        if (tpb7 < 2 && g7.ntiles >= 2) tpb7 = 2;
        g7.nb = cdiv(g7.ntiles, tpb7);
        g7.H = Hi; g7.W = Wi;
        if (!accumulate) TF_LAUNCH(fill_f32_kernel, dim3(cdiv((long)C * 9 * CG, 1024)), dim3(256), stream, dw, (long)C * 9 * CG);
        if (tw == 16) TF_LAUNCH((conv3x3_grouped_wgrad7_kernel<16, true>), dim3(g7.G * g7.nb), dim3(448), stream, x, dy, dw, g7, in_coef, Ho, Wo);
        else TF_LAUNCH((conv3x3_grouped_wgrad7_kernel<32, true>), dim3(g7.G * g7.nb), dim3(448), stream, x, dy, dw, g7, in_coef, Ho, Wo);
        return launch_status("tf_conv3x3_grouped_s2_wgrad_f32");
    }
#define TF_GW2(TW_, P_) TF_LAUNCH((conv3x3_grouped_s2_wgrad_kernel<TW_, P_>), dim3(g.G * g.nb), dim3(192), stream, x, dy, ws, g, Ho, Wo, in_coef)
    if (tw == 16) { if (prec == 2) TF_GW2(16, 2); else if (prec == 1 || prec == 3) TF_GW2(16, 1); else TF_GW2(16, 0); }
    else { if (prec == 2) TF_GW2(32, 2); else if (prec == 1 || prec == 3) TF_GW2(32, 1); else TF_GW2(32, 0); }
#undef TF_GW2
    TF_LAUNCH(conv3x3_grouped_wgrad_reduce_kernel, dim3(36, g.G), dim3(256), stream, (const float*)ws, g.nb, dw, accumulate);
    return launch_status("tf_conv3x3_grouped_s2_wgrad_f32");
}

extern "C" int tf_conv3x3_grouped_s2_dgrad_f32(const float* dy, const float* w, float* dx, int B, int Hi, int Wi, int C, int accumulate, void* stream) {
    TF_REQUIRE(args_ok(dy, w, dx, B, Hi, Wi, C), "tf_conv3x3_grouped_s2_dgrad_f32: needs NHWC tensors with C %% 24 == 0 (group width 24), 16-byte aligned");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;        // 3x3 / stride 2 / pad 1
    const int tw = pick_tw(Ho, Wo);
    GcGeom g = make_geom(B, Ho, Wo, C, tw);                        // tiles over the dY grid
    const int prec = tf::gemm_precision();      // 0 / 2 (f32x3: the exact fp32 MFMA is at least as accurate): fp32 path; 1 / 3: bf16 / fp16 operands
    if (tw == 16) TF_LAUNCH((conv3x3_grouped_s2_dgrad_kernel<16>), dim3(g.G * g.nb), dim3(256), stream, dy, w, dx, g, Hi, Wi, accumulate, prec);
    else TF_LAUNCH((conv3x3_grouped_s2_dgrad_kernel<32>), dim3(g.G * g.nb), dim3(256), stream, dy, w, dx, g, Hi, Wi, accumulate, prec);
    return launch_status("tf_conv3x3_grouped_s2_dgrad_f32");
}

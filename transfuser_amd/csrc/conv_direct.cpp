// Direct (LDS-tiled) 3x3 / stride 1 / pad 1 convolutions for SMALL channel counts at LARGE resolution: the last layers of the
// segmentation / depth decoders (transfuser.py:232-237,267-272: 32 -> 32, 32 -> 7, 32 -> 1 at 256 x 704, B = 10) and their gradients.
//
// Through the implicit-GEMM engine every input pixel is re-read 9x through L2 (one im2col row per tap): those launches are L2-bandwidth
// bound (~5 TB/s, 0.4-0.8 ms each, 12 per step).  Here a block stages a (4+2) x (32+2) pixel patch (<= 32 channels) in LDS ONCE and
// every tap reads it from there: HBM/L2 traffic drops to one read of x and one write of y.  The contraction still runs on the fp32
// MFMA (v_mfma_f32_32x32x2_f32): per wave one row of 32 output pixels x 32 output channels (channel counts are zero-padded to 32).
// Blocks are persistent over tiles so the 9 x 32 x 32 weight panel is staged once per block.
#include "tf_common.h"
#include <stdlib.h>
#include "../../include/transfuser_hip.h"

using namespace tf;
namespace tf { int gemm_precision(); }   // api.cpp (tf_set_precision): 1 = bf16-MFMA contractions, 2 = bf16x3 split (fp32-accurate) on the bf16 MFMA

namespace {

constexpr int TH = 4, TW = 32;                 // output tile: 4 rows x 32 columns = 4 waves x 32 pixels
constexpr int PH = TH + 2, PW = TW + 2;        // input patch with halo
constexpr int PP = 33;                         // floats per patch pixel (32 channels + 1: conflict-free stride for the MFMA A fetch)
constexpr int WP = 36;                         // pitch of a weight row (32 output channels + 4)
constexpr int NV = (PH * PW * 8 + 255) / 256;  // float4 patch slots per thread

struct DcGeom { int B, H, W, Ci, Co, tiles_h, tiles_w, ntiles, f16; };   // f16: the PREC 1 paths use IEEE-half operands (tf_set_precision(3))

// stage W into LDS as wl[(tap, k)][n]: fwd  wl[tap][ci][co] = Wt[co][tap][ci];  dgrad  wl[tap][co][ci] = Wt[co][8 - tap][ci].
// Wt is CoW x 9 x CiW <= 9216 CONTIGUOUS floats: a thread issues its (up to 36) loads back to back from clamped addresses, the first patch's
// loads follow, and only then is the panel scattered into LDS (round 3: one predicated element per loop trip = 36 dependent round trips,
// ~30 us in front of every launch).
constexpr int WNL = 9 * 32 * 32 / 256;
struct WeightRegs { float v[WNL]; };
__device__ __forceinline__ void issue_weights(WeightRegs& r, const float* __restrict__ w, int CoW, int CiW) {
    const int ne = CoW * 9 * CiW;
#pragma unroll
    for (int p = 0; p < WNL; ++p) { const int e = threadIdx.x + p * 256; r.v[p] = w[e < ne ? e : ne - 1]; }
}
__device__ __forceinline__ void scatter_weights(float (*wl)[WP], const WeightRegs& r, int CoW, int CiW, int dgrad) {
    const int ne = CoW * 9 * CiW, row = 9 * CiW;
    for (int i = threadIdx.x; i < 9 * 32 * 32; i += 256) {         // the zero padding of a panel narrower than 32 x 32 (slots the scatter never writes)
        const int n = i & 31, k = (i >> 5) & 31, tap = i >> 10;
        const bool used = dgrad ? (k < CoW && n < CiW) : (n < CoW && k < CiW);
        if (!used) wl[tap * 32 + k][n] = 0.f;
    }
#pragma unroll
    for (int p = 0; p < WNL; ++p) {
        const int e = threadIdx.x + p * 256;
        const int co = e / row, q = e - co * row, tap = q / CiW, ci = q - tap * CiW;
        if (e < ne) {
            if (dgrad) wl[(8 - tap) * 32 + co][ci] = r.v[p]; else wl[tap * 32 + ci][co] = r.v[p];
        }
    }
}

// patch slot s of this thread -> (pixel, channel quad); loads 4 channels of one patch pixel (zero outside the image / beyond Ci)
template <bool VEC>
__device__ __forceinline__ float4 load_patch_slot(const float* __restrict__ x, const DcGeom& g, int b, int h0, int w0, int s) {
    const int pix = s >> 3, c = (s & 7) * 4;
    const int ph = pix / PW, pw = pix - ph * PW;
    const int h = h0 - 1 + ph, w = w0 - 1 + pw;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pix < PH * PW && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W && c < g.Ci) {
        const float* p = x + (((long)b * g.H + h) * g.W + w) * g.Ci + c;
        if (VEC) v = *reinterpret_cast<const float4*>(p);
        else { v.x = p[0]; if (c + 1 < g.Ci) v.y = p[1]; if (c + 2 < g.Ci) v.z = p[2]; if (c + 3 < g.Ci) v.w = p[3]; }
    }
    return v;
}
__device__ __forceinline__ void store_patch_slot(float* patch, int s, const float4 v) {
    const int pix = s >> 3, c = (s & 7) * 4;
    if (pix < PH * PW) { float* q = patch + pix * PP + c; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
}

// y = conv3x3(x, W) (+bias) (relu) (+= when accumulate).  dgrad != 0: x is dY (Ci = W's Cout), y is dX (Co = W's Cin).
// TRN (round 6, PREC 0 only): the weight panel is the MFMA's A operand and the patch its B operand, i.e. the wave computes the TRANSPOSED tile (rows = output
// channels, columns = its 32 pixels): a lane then owns 16 channels of ONE pixel as four groups of four consecutive channels - four 16-byte stores per
// lane and tile instead of sixteen 4-byte ones (in the grouped kernels that form was 8 of 38 us, csrc/conv_grouped.cpp).  Needs Co % 4 == 0 and 16-byte aligned y / bias.
template <bool VEC, int PREC, bool TRN = false>
__global__ void __launch_bounds__(256, 2) conv3x3_small_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, DcGeom g, int CoW, int CiW, int dgrad, int relu, int accumulate) {
    __shared__ float patch[PH * PW * PP];
    __shared__ float wl[9 * 32][WP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    WeightRegs wreg;
    issue_weights(wreg, w, CoW, CiW);
    const int kpairs = (g.Ci + 1) >> 1;          // MFMA k-steps per tap (channels padded to even with the zero-filled LDS columns)
    int tile = blockIdx.x;
    float4 pre[NV];
    auto fetch = [&](int t) {
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * TH, w0 = (r % g.tiles_w) * TW;
#pragma unroll
        for (int p = 0; p < NV; ++p) pre[p] = load_patch_slot<VEC>(x, g, b, h0, w0, tid + p * 256);
    };
    if (tile < g.ntiles) fetch(tile);
    scatter_weights(wl, wreg, CoW, CiW, dgrad);
    for (; tile < g.ntiles; tile += gridDim.x) {
        __syncthreads();                           // previous tile's MFMAs are done with the patch (and the weights are staged)
#pragma unroll
        for (int p = 0; p < NV; ++p) store_patch_slot(patch, tid + p * 256, pre[p]);
        __syncthreads();
        const int nxt = tile + gridDim.x;
        if (nxt < g.ntiles) fetch(nxt);            // next patch travels while this one is multiplied
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (PREC != 0) { // bf16 MFMA (PREC 1: rounded operands, 2: bf16x3 split): channels are zero-padded to 32 in both LDS tiles -> two 16-deep groups per tap
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const float* pa = patch + ((wave + kh) * PW + l31 + kw) * PP;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (16 * q >= g.Ci) break;
                    float a[8], b[8];
                    const int k0 = 16 * q + 8 * hi;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a[j] = pa[k0 + j]; b[j] = wl[tap * 32 + k0 + j][l31]; }
                    if constexpr (PREC == 2) mfma_32x32x16_x3(a, b, acc); else mfma_32x32x16_lp(a, b, acc, g.f16 ? 3 : 1);
                }
            }
        } else {
        for (int tap = 0; tap < 9; ++tap) {        // (explicit operand prefetch + scheduling fences measured slower here: 463 vs 366 us)
            const int kh = tap / 3, kw = tap - kh * 3;
            const float* pa = patch + ((wave + kh) * PW + l31 + kw) * PP + hi;
            const float* pb = &wl[tap * 32 + hi][l31];
            for (int kk = 0; kk < kpairs; ++kk) {
                if constexpr (TRN) mfma_32x32x2(pb[2 * kk * WP], pa[2 * kk], acc);      // D[i = channel][j = pixel]
                else mfma_32x32x2(pa[2 * kk], pb[2 * kk * WP], acc);
            }
        }
        }
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h = (r / g.tiles_w) * TH + wave, w0 = (r % g.tiles_w) * TW;
        if constexpr (TRN) {
            if (h < g.H && w0 + l31 < g.W) {                       // this lane's pixel; channels 8 q + 4 hi .. + 3 = accumulator elements 4 q .. 4 q + 3
                float* dst = y + (((long)b * g.H + h) * g.W + w0 + l31) * g.Co + 4 * hi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (8 * q + 4 * hi < g.Co) {
                        float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                        if (bias) { const float4 bj = *reinterpret_cast<const float4*>(bias + 8 * q + 4 * hi); v.x += bj.x; v.y += bj.y; v.z += bj.z; v.w += bj.w; }
                        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        float4* d4 = reinterpret_cast<float4*>(dst + 8 * q);
                        if (accumulate) { const float4 o = *d4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                        *d4 = v;
                    }
                }
            }
        } else
        if (h < g.H && l31 < g.Co) {
            const float bj = bias ? bias[l31] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int wv = w0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (wv < g.W) {
                    float* dst = y + (((long)b * g.H + h) * g.W + wv) * g.Co + l31;
                    float v = acc[e] + bj;
                    if (relu) v = fmaxf(v, 0.f);
                    *dst = accumulate ? *dst + v : v;
                }
            }
        }
    }
}

// ---- f32x3 (bf16x3 split) forward / dgrad with PRE-SPLIT operands: the in-register split of conv3x3_small_kernel<VEC, 2> re-splits every patch
// element once per tap (9x) and every weight once per tile, which makes a single 32x32 tile VALU-bound (288 VALU vs 192 MFMA cycles per
// 16-deep group).  Here both operands are split ONCE on their way into LDS and live there as three packed-bf16 planes (h, m, l):
//   patch planes  xp[plane][pixel][20 dwords]  (16 dwords = 32 channels, 80-byte pitch: conflict-free ds_read_b128 of 8 channels)
//   weight planes wq[plane][(tap, q, hi)][n = 32][4 dwords]   (k = 16 q + 8 hi + 2 d + {0, 1} in dword d; lane-contiguous b128 reads)
// so the tap loop is 6 x ds_read_b128 + 6 x v_mfma_f32_32x32x16_bf16 per 16-deep group and nothing else.  8 waves / block (8 x 32 output
// pixels, one row per wave), 137 KB of LDS -> one block per CU, persistent over tiles with register prefetch of the next patch.
constexpr int XH = 8, XPH = XH + 2;            // tile rows / patch rows (tile and patch width = TW, PW)
constexpr int XPL = 20;                        // dwords per patch pixel per plane
constexpr int XNV = (XPH * PW * 8 + 511) / 512;
constexpr int XPLANE_P = XPH * PW * XPL;       // dwords per patch plane
constexpr int XPLANE_W = 9 * 2 * 2 * 32 * 4;   // dwords per weight plane

__global__ void __launch_bounds__(512, 1) conv3x3_small_x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ y, DcGeom g, int CoW, int CiW, int dgrad, int relu, int accumulate) {
    __shared__ __attribute__((aligned(16))) uint32_t xp[3 * XPLANE_P];
    __shared__ __attribute__((aligned(16))) uint32_t wq[3 * XPLANE_W];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // weights: dword (tap, q, hi, n, d) holds k = 16 q + 8 hi + 2 d and k + 1 of column n (same (tap, k, n) convention as load_weights)
    for (int i = tid; i < XPLANE_W; i += 512) {
        const int d = i & 3, n = (i >> 2) & 31, grp = i >> 7, tap = grp >> 2, k = 16 * ((grp >> 1) & 1) + 8 * (grp & 1) + 2 * d;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int kk = k + e;
            v[e] = 0.f;
            if (!dgrad) { if (n < CoW && kk < CiW) v[e] = w[((long)n * 9 + tap) * CiW + kk]; }
            else { if (kk < CoW && n < CiW) v[e] = w[((long)kk * 9 + (8 - tap)) * CiW + n]; }
        }
        split_pair_bf16x3(v[0], v[1], wq[i], wq[XPLANE_W + i], wq[2 * XPLANE_W + i]);
    }
    const int tiles_h = (g.H + XH - 1) / XH, ntiles = g.B * tiles_h * g.tiles_w;
    float4 pre[XNV];
    auto fetch = [&](int t) {
        const int b = t / (tiles_h * g.tiles_w), r = t - b * (tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * XH, w0 = (r % g.tiles_w) * TW;
#pragma unroll
        for (int p = 0; p < XNV; ++p) {
            const int s = tid + p * 512, pix = s >> 3, c = (s & 7) * 4;
            const int ph = pix / PW, pw = pix - ph * PW;
            const int h = h0 - 1 + ph, ww = w0 - 1 + pw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < XPH * PW && (unsigned)h < (unsigned)g.H && (unsigned)ww < (unsigned)g.W && c < g.Ci)
                v = *reinterpret_cast<const float4*>(x + (((long)b * g.H + h) * g.W + ww) * g.Ci + c);
            pre[p] = v;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                           // previous tile's MFMAs are done with the patch planes (and the weights are staged)
#pragma unroll
        for (int p = 0; p < XNV; ++p) {
            const int s = tid + p * 512, pix = s >> 3, c = (s & 7) * 4;
            if (pix < XPH * PW) {
                uint32_t* q = xp + pix * XPL + (c >> 1);
                split_pair_bf16x3(pre[p].x, pre[p].y, q[0], q[XPLANE_P], q[2 * XPLANE_P]);
                split_pair_bf16x3(pre[p].z, pre[p].w, q[1], q[XPLANE_P + 1], q[2 * XPLANE_P + 1]);
            }
        }
        __syncthreads();
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) fetch(nxt);              // next patch travels while this one is multiplied
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const uint32_t* pa = xp + ((wave + kh) * PW + l31 + kw) * XPL + 4 * hi;
            const uint32_t* pb = wq + ((tap * 4 + hi) * 32 + l31) * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (16 * q >= g.Ci) break;
                const Bf16x3 fa = frag_from_planes(pa + 8 * q, pa + 8 * q + XPLANE_P, pa + 8 * q + 2 * XPLANE_P);
                const Bf16x3 fb = frag_from_planes(pb + q * 256, pb + q * 256 + XPLANE_W, pb + q * 256 + 2 * XPLANE_W);
                mfma_x3_presplit(fa, fb, acc);
            }
        }
        const int b = tile / (tiles_h * g.tiles_w), r = tile - b * (tiles_h * g.tiles_w);
        const int h = (r / g.tiles_w) * XH + wave, w0 = (r % g.tiles_w) * TW;
        if (h < g.H && l31 < g.Co) {
            const float bj = bias ? bias[l31] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int wv = w0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (wv < g.W) {
                    float* dst = y + (((long)b * g.H + h) * g.W + wv) * g.Co + l31;
                    float v = acc[e] + bj;
                    if (relu) v = fmaxf(v, 0.f);
                    *dst = accumulate ? *dst + v : v;
                }
            }
        }
    }
}

// dW[co][tap][ci] (+)= sum_pixels dY[p][co] * X[p + tap][ci]: per wave a row of 32 pixels as the K dimension, 9 accumulators (one per
// tap, 32 co x 32 ci); blocks are persistent, their partial panels are summed by conv3x3_small_wgrad_reduce_kernel.
template <bool VEC, int PREC>
__global__ void __launch_bounds__(256, 1) conv3x3_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                                     DcGeom g) {
    __shared__ float patch[PH * PW * PP];
    __shared__ float dyt[TH * TW * PP];            // [pixel][co], co padded to 32 (+1)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // register prefetch of the next tile (patch + dY rows) while the current one is multiplied
    constexpr int ND = TH * TW * 32 / 256;         // dY elements per thread (co fastest: thread -> fixed co, ND pixels)
    float4 pre[NV];
    float dpre[ND];
    auto fetch = [&](int t) {
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * TH, w0 = (r % g.tiles_w) * TW;
#pragma unroll
        for (int p = 0; p < NV; ++p) pre[p] = load_patch_slot<VEC>(x, g, b, h0, w0, tid + p * 256);
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const int i = tid + p * 256, co = i & 31, pix = i >> 5, ph = pix / TW, pw = pix - ph * TW;
            const int h = h0 + ph, w = w0 + pw;
            const bool ok = co < g.Co && h < g.H && w < g.W;
            dpre[p] = ok ? dy[(((long)b * g.H + h) * g.W + w) * g.Co + co] : 0.f;
        }
    };
    int tile = blockIdx.x;
    if (tile < g.ntiles) fetch(tile);
    for (; tile < g.ntiles; tile += gridDim.x) {
        __syncthreads();                           // the previous tile's MFMAs are done with both LDS tiles
#pragma unroll
        for (int p = 0; p < NV; ++p) store_patch_slot(patch, tid + p * 256, pre[p]);
#pragma unroll
        for (int p = 0; p < ND; ++p) { const int i = tid + p * 256; dyt[(i >> 5) * PP + (i & 31)] = dpre[p]; }
        __syncthreads();
        if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);
        // K = the 32 pixels of this wave's row: A[i = co][k = pixel] = dyt (shared by the 9 taps), B[k = pixel][j = ci] = patch shifted by
        // the tap -> 9 independent accumulator chains per k step
        const float* pa = dyt + (wave * TW + hi) * PP + l31;
        const float* pb = patch + (wave * PW + hi) * PP + l31;
        if constexpr (PREC != 0) { // bf16 MFMA: the row's 32 pixels = two 16-deep K groups; lane half hi owns pixels 16 q + 8 hi .. + 7
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float a[8], b[9][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pi = 16 * q + 8 * hi + j;
                    a[j] = dyt[(wave * TW + pi) * PP + l31];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) b[tap][j] = patch[((wave + tap / 3) * PW + pi + tap % 3) * PP + l31];
                }
                if constexpr (PREC == 2) {      // the dY fragment is split once and shared by the 9 taps
                    const Bf16x3 fa = split_bf16x3(a);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) mfma_x3_presplit(fa, split_bf16x3(b[tap]), acc[tap]);
                } else {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) mfma_32x32x16_lp(a, b[tap], acc[tap], g.f16 ? 3 : 1);
                }
            }
        } else {
#pragma unroll 2
        for (int kk = 0; kk < TW / 2; ++kk) {
            const float a = pa[2 * kk * PP];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                mfma_32x32x2(a, pb[((kh * PW + kw) + 2 * kk) * PP], acc[tap]);
            }
        }
        }
    }
    // reduce the 4 waves through LDS (patch + dyt are free now), then one partial panel per block: part[block][tap][co][ci]
    __syncthreads();
    float* red = patch;                            // 6732 floats >= 32*32 per pass
    for (int tap = 0; tap < 9; ++tap) {
        for (int wv = 0; wv < 4; ++wv) {
            if (wave == wv) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
                    float* q = red + i * 32 + l31;
                    *q = (wv == 0) ? acc[tap][e] : *q + acc[tap][e];
                }
            }
            __syncthreads();
        }
        for (int i = tid; i < 1024; i += 256) part[((long)blockIdx.x * 9 + tap) * 1024 + i] = red[i];
        __syncthreads();
    }
}

// one WAVE per (tap, co, ci): the lanes stride over the blocks' panels, then a shuffle tree (a serial loop over 256 panels per thread measured 60 us)
__global__ void __launch_bounds__(256) conv3x3_small_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dw, int Co, int Ci,
                                                                         int accumulate) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // (tap, co, ci) over 9 x 32 x 32
    const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
    const bool live = i < 9 * 1024 && co < Co && ci < Ci;      // wave-uniform
    float s = 0.f;
    if (live)
        for (int b = lane; b < nblocks; b += 64) s += part[(long)b * 9216 + i];
    s = wave_sum(s);
    if (live && lane == 0) {
        float* d = dw + ((long)co * 9 + tap) * Ci + ci;
        *d = accumulate ? *d + s : s;
    }
}

// compute precision of these kernels; TF_X3_DIRECT=0 keeps them on the exact fp32 MFMA in f32x3 mode (A/B switch)
inline int direct_prec() {
    static const bool x3 = [] { const char* e = getenv("TF_X3_DIRECT"); return !e || atoi(e) != 0; }();
    const int p = tf::gemm_precision();
    return (p == 2 && !x3) ? 0 : (p == 3 ? 1 : p);      // fp16 mode runs the PREC 1 instantiations with DcGeom.f16 set
}

// transposed-tile form of the fp32 forward / input gradient (conv3x3_small_kernel<.., TRN>): TF_SMALL_TRN=0 switches it off (A/B)
inline bool trn_ok(int Co, const float* y, const float* bias) {
    static const bool on = [] { const char* e = getenv("TF_SMALL_TRN"); return !e || atoi(e) != 0; }();
    return on && Co % 4 == 0 && aligned16(y) && (!bias || aligned16(bias));
}

inline bool presplit_enabled() { static const bool on = [] { const char* e = getenv("TF_X3_PRESPLIT"); return !e || atoi(e) != 0; }(); return on; }

inline DcGeom make_geom(int B, int H, int W, int Ci, int Co) {
    DcGeom g; g.B = B; g.H = H; g.W = W; g.Ci = Ci; g.Co = Co; g.f16 = tf::gemm_precision() == 3 ? 1 : 0;
    g.tiles_h = cdiv(H, TH); g.tiles_w = cdiv(W, TW); g.ntiles = B * g.tiles_h * g.tiles_w;
    return g;
}

}  // namespace

extern "C" int tf_conv3x3_small_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int relu,
                                        void* stream) {
    TF_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 32 && Cout > 0 && Cout <= 32, "tf_conv3x3_small_fwd_f32: needs Cin, Cout <= 32");
    DcGeom g = make_geom(B, H, W, Cin, Cout);
    const int grid = g.ntiles < 512 ? g.ntiles : 512;
    const bool vec = Cin % 4 == 0 && aligned16(x);
    const int prec = direct_prec();
    if (prec == 2 && vec && presplit_enabled()) {      // f32x3 with pre-split LDS planes: 8-row tiles, one 512-thread block per CU
        const int nt = B * cdiv(H, XH) * g.tiles_w;
        TF_LAUNCH(conv3x3_small_x3_kernel, dim3(nt < 256 ? nt : 256), dim3(512), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0);
        return launch_status("tf_conv3x3_small_fwd_f32[x3]");
    }
    if (prec == 2) { if (vec) TF_LAUNCH((conv3x3_small_kernel<true, 2>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0); else TF_LAUNCH((conv3x3_small_kernel<false, 2>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0); }
    else if (prec == 1) { if (vec) TF_LAUNCH((conv3x3_small_kernel<true, 1>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0); else TF_LAUNCH((conv3x3_small_kernel<false, 1>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0); }
    else if (vec && trn_ok(Cout, y, bias)) TF_LAUNCH((conv3x3_small_kernel<true, 0, true>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0);
    else { if (vec) TF_LAUNCH((conv3x3_small_kernel<true, 0>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0); else TF_LAUNCH((conv3x3_small_kernel<false, 0>), dim3(grid), dim3(256), stream, x, w, bias, y, g, Cout, Cin, 0, relu, 0); }
    return launch_status("tf_conv3x3_small_fwd_f32");
}

extern "C" int tf_conv3x3_small_dgrad_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout, int accumulate, void* stream) {
    TF_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 32 && Cout > 0 && Cout <= 32, "tf_conv3x3_small_dgrad_f32: needs Cin, Cout <= 32");
    DcGeom g = make_geom(B, H, W, Cout, Cin);      // the "input" of this pass is dY (Cout channels), the output dX (Cin channels)
    const int grid = g.ntiles < 512 ? g.ntiles : 512;
    const bool vec = Cout % 4 == 0 && aligned16(dy);
    int prec = direct_prec();
    if (prec == 2 && Cout < 16) prec = 0;     // few input channels (the 32 -> 7 / 32 -> 1 layers' gradients): the fp32 kernel multiplies only ceil(Cout / 2) k-steps per tap
                                              // and beats a zero-padded 16-deep bf16x3 group (measured 238 / 122 us vs 271 / 210)
    const float* nob = nullptr;
    if (prec == 2 && vec && presplit_enabled()) {
        const int nt = B * cdiv(H, XH) * g.tiles_w;
        TF_LAUNCH(conv3x3_small_x3_kernel, dim3(nt < 256 ? nt : 256), dim3(512), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate);
        return launch_status("tf_conv3x3_small_dgrad_f32[x3]");
    }
    if (prec == 2) { if (vec) TF_LAUNCH((conv3x3_small_kernel<true, 2>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate); else TF_LAUNCH((conv3x3_small_kernel<false, 2>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate); }
    else if (prec == 1) { if (vec) TF_LAUNCH((conv3x3_small_kernel<true, 1>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate); else TF_LAUNCH((conv3x3_small_kernel<false, 1>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate); }
    else if (vec && trn_ok(Cin, dx, nob)) TF_LAUNCH((conv3x3_small_kernel<true, 0, true>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate);
    else { if (vec) TF_LAUNCH((conv3x3_small_kernel<true, 0>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate); else TF_LAUNCH((conv3x3_small_kernel<false, 0>), dim3(grid), dim3(256), stream, dy, w, nob, dx, g, Cout, Cin, 1, 0, accumulate); }
    return launch_status("tf_conv3x3_small_dgrad_f32");
}

extern "C" long tf_conv3x3_small_wgrad_ws_floats(void) { return 256L * 9216; }

extern "C" int tf_conv3x3_small_wgrad_f32(const float* dy, const float* x, float* dw, int B, int H, int W, int Cin, int Cout, int accumulate, float* ws,
                                          void* stream) {
    TF_REQUIRE(dy && x && dw && ws && B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 32 && Cout > 0 && Cout <= 32,
               "tf_conv3x3_small_wgrad_f32: needs Cin, Cout <= 32 and ws of tf_conv3x3_small_wgrad_ws_floats() floats");
    DcGeom g = make_geom(B, H, W, Cin, Cout);
    const int grid = g.ntiles < 256 ? g.ntiles : 256;
    const bool vec = Cin % 4 == 0 && aligned16(x);
    const int prec = direct_prec();
    if (prec == 2) { if (vec) TF_LAUNCH((conv3x3_small_wgrad_kernel<true, 2>), dim3(grid), dim3(256), stream, x, dy, ws, g); else TF_LAUNCH((conv3x3_small_wgrad_kernel<false, 2>), dim3(grid), dim3(256), stream, x, dy, ws, g); }
    else if (prec == 1) { if (vec) TF_LAUNCH((conv3x3_small_wgrad_kernel<true, 1>), dim3(grid), dim3(256), stream, x, dy, ws, g); else TF_LAUNCH((conv3x3_small_wgrad_kernel<false, 1>), dim3(grid), dim3(256), stream, x, dy, ws, g); }
    else { if (vec) TF_LAUNCH((conv3x3_small_wgrad_kernel<true, 0>), dim3(grid), dim3(256), stream, x, dy, ws, g); else TF_LAUNCH((conv3x3_small_wgrad_kernel<false, 0>), dim3(grid), dim3(256), stream, x, dy, ws, g); }
    TF_LAUNCH(conv3x3_small_wgrad_reduce_kernel, dim3(9 * 1024 / 4), dim3(256), stream, (const float*)ws, grid, dw, Cout, Cin, accumulate);
    return launch_status("tf_conv3x3_small_wgrad_f32");
}

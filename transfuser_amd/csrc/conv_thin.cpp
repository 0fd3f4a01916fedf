// 3x3 / stride 1 / pad 1 convolutions with a THIN output (Cout <= 7) on a 32-channel input at full image resolution: the last layer of
// the segmentation / depth decoders (transfuser.py:232-237,267-272: 32 -> 7 and 32 -> 1 at 256 x 704, B = 10) and its two gradients.
//
// These layers are HBM-bound by nature (230 MB of input for 7 or 1 output channels), but a direct MFMA mapping pads the output channels
// to a 32-column tile and spends 9 taps x 16 k-steps = 144 MFMAs per 32 pixels - exactly the cost of a 32 -> 32 layer (349 us measured for
// either Cout; 46 us of HBM time).  Here the 9 taps are folded into the GEMM's N / K dimension instead:
//   forward   P[pixel][(tap, co)] = sum_ci x[pixel][ci] w[co][tap][ci]   - ONE 1x1-convolution GEMM with N = 9 Cout (63 -> two 32-column
//             tiles: 32 MFMAs per 32 pixels, 16 for Cout = 1), P parked in LDS for an (TH+2) x 32 pixel patch, then
//             y[p][co] = bias[co] + sum_tap P[p + tap][(tap, co)]  (9 LDS reads per output);
//   dgrad     dx[q][ci] = sum_{(tap, co)} dy[q - tap][co] w[co][tap][ci]: K = 9 Cout (32 k-steps; 5 for Cout = 1) with the A fragments
//             gathered from a thin dY patch in LDS; the ReLU mask of the PRECEDING layer (its output = this layer's input) is applied in
//             the epilogue, which removes that layer's separate mask pass over 230 MB;
//   wgrad     dW[(tap, co)][ci] = sum_q dy[q - tap][co] x[q][ci]: M = 9 Cout (two 32-row tiles), K = pixels, x streamed straight from
//             global memory into the B fragments (one 128-byte line per half wave), 32 MFMAs per 32 pixels; per-block partial panels +
//             a reduce kernel (deterministic); the bias gradient rides along (the dY patch is already in LDS).
// The input pixels' 32 channels are the K dimension: lane half `hi` of an A fragment owns channels 16 hi .. 16 hi + 15 (the MFMA's k slots
// are permuted identically on the weight side), so a lane fetches its operand with four 16-byte loads and x never passes through LDS.
// Exact fp32 MFMA in every precision mode (the layers are bandwidth-bound; nothing to gain from bf16 operands).
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int CI = 32;             // input channels (deconv_channel_num_3)
constexpr int TWF = 30;            // forward: output tile width (patch width TWF + 2 = 32 = one MFMA row block)
constexpr int PWD = 34;            // dgrad / wgrad: dY patch width (32 + 2)
constexpr int THD = 8;             // dgrad / wgrad: tile rows (2 per wave)

struct ThGeom { int B, H, W, Co, tiles_h, tiles_w, ntiles; };

__device__ __forceinline__ int acc_row(int e, int hi) { return (e & 3) + 8 * (e >> 2) + 4 * hi; }

// ------------------------------------------------------------------------------------------------ forward
template <int NT, int TH, int CO_T, int PMAX>      // PMAX >= (9 Co) | 1: floats per patch pixel of P
__global__ void __launch_bounds__(256, 2) conv3x3_thin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ y, ThGeom g) {
    constexpr int PR = TH + 2, RPW = PR / 4;
    static_assert(PR % 4 == 0, "patch rows must split over the 4 waves");
    const int Co = CO_T ? CO_T : g.Co, NC = 9 * Co, pitch = NC | 1;
    __shared__ float P[PR * 32 * PMAX];                                // [PR * 32][pitch]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // weight fragments: column n = nt * 32 + l31 = (tap, co); k slot (kk, hi) = channel 16 hi + kk
    float bw[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + l31;
        const bool live = n < NC;
        const int tap = live ? n / Co : 0, co = live ? n - tap * Co : 0;
        const float* wp = w + ((long)co * 9 + tap) * CI + 16 * hi;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) bw[nt][kk] = wp[kk];              // dead columns fetch column 0 (all loads unconditional, zeroed below)
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float m = (nt * 32 + l31 < NC) ? 1.f : 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) bw[nt][kk] *= m;
    }
    // operand fetch: UNCONDITIONAL 16-byte loads (out-of-image pixels read a clamped in-image address; their P rows are zeroed when P is
    // stored) - predicated loads make hipcc wait for every load separately; the next tile's rows travel during the tap sums
    float4 pre[RPW][4];
    auto fetch = [&](int t) {
        const int b = t / (g.tiles_h * g.tiles_w), r = t - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * TH, w0 = (r % g.tiles_w) * TWF;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            int h = h0 - 1 + wave * RPW + rr, ww = w0 - 1 + l31;
            h = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h);
            ww = ww < 0 ? 0 : (ww >= g.W ? g.W - 1 : ww);
            const float4* p = reinterpret_cast<const float4*>(x + (((long)b * g.H + h) * g.W + ww) * CI + 16 * hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) pre[rr][q] = p[q];
        }
    };
    int tile = blockIdx.x;
    if (tile < g.ntiles) fetch(tile);
    for (; tile < g.ntiles; tile += gridDim.x) {
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * TH, w0 = (r % g.tiles_w) * TWF;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int pr = wave * RPW + rr;
            const bool hok = (unsigned)(h0 - 1 + pr) < (unsigned)g.H;
            float a[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[4 * q] = pre[rr][q].x; a[4 * q + 1] = pre[rr][q].y; a[4 * q + 2] = pre[rr][q].z; a[4 * q + 3] = pre[rr][q].w; }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) mfma_32x32x2(a[kk], bw[nt][kk], acc);
                const int n = nt * 32 + l31;
                if (n < NC) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int pc = acc_row(e, hi);
                        P[(pr * 32 + pc) * pitch + n] = (hok && (unsigned)(w0 - 1 + pc) < (unsigned)g.W) ? acc[e] : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);
        for (int idx = tid; idx < TH * TWF; idx += 256) {
            const int oh = idx / TWF, ow = idx - oh * TWF;
            const int h = h0 + oh, ww = w0 + ow;
            if (h < g.H && ww < g.W) {
                float* dst = y + (((long)b * g.H + h) * g.W + ww) * Co;
                const float* p0 = P + (oh * 32 + ow) * pitch;
                for (int co = 0; co < Co; ++co) {
                    float s = bias ? bias[co] : 0.f;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) s += p0[((tap / 3) * 32 + (tap % 3)) * pitch + tap * Co + co];
                    dst[co] = s;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ shared by dgrad / wgrad: the dY patch
// dp[(r * PWD + c) * Co + co] = dY[b][h0 - 1 + r][w0 - 1 + c][co], r < THD + 2, c < PWD (zero outside the image).  Staged through registers
// (NDP slots per thread, all loads issued before the first use, unconditional with a clamped address) so that the next tile's patch can
// travel while the current one is multiplied; slot i covers element tid + 256 i of the (THD + 2) x (PWD Co) patch, its (row, column)
// pair advances without divisions.
constexpr int NDP = ((THD + 2) * PWD * 7 + 255) / 256;
struct DyPatch { float v[NDP]; uint32_t ok; };     // ok: bit i = slot i is inside the image (applied when the patch is stored: no select at the loads)
__device__ __forceinline__ void fetch_dy_patch(DyPatch& d, const float* __restrict__ dy, const ThGeom& g, int Co, int tile) {
    const int b = tile / (g.tiles_h * g.tiles_w), rt = tile - b * (g.tiles_h * g.tiles_w);
    const int h0 = (rt / g.tiles_w) * THD, w0 = (rt % g.tiles_w) * 32;
    const int rowlen = PWD * Co, total = (THD + 2) * rowlen;
    const int jlo = w0 == 0 ? Co : 0, jhi = (g.W - w0 + 1 < PWD ? g.W - w0 + 1 : PWD) * Co;      // in-image element range of a patch row
    int r = threadIdx.x / rowlen, j = threadIdx.x - r * rowlen;
    const int dr = 256 / rowlen, dj = 256 - dr * rowlen;
    const long last = (long)g.B * g.H * g.W * Co - 1;
    d.ok = 0u;
#pragma unroll
    for (int i = 0; i < NDP; ++i) {
        const int h = h0 - 1 + r;
        const bool ok = threadIdx.x + 256 * i < total && (unsigned)h < (unsigned)g.H && j >= jlo && j < jhi;
        long o = (((long)b * g.H + h) * g.W + w0 - 1) * Co + j;
        o = o < 0 ? 0 : (o > last ? last : o);
        d.v[i] = dy[o];
        d.ok |= ok ? (1u << i) : 0u;
        r += dr; j += dj;
        if (j >= rowlen) { j -= rowlen; ++r; }
    }
}
__device__ __forceinline__ void store_dy_patch(float* dp, const DyPatch& d, int Co) {
    const int total = (THD + 2) * PWD * Co;
#pragma unroll
    for (int i = 0; i < NDP; ++i)
        if (threadIdx.x + 256 * i < total) dp[threadIdx.x + 256 * i] = ((d.ok >> i) & 1u) ? d.v[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------ dgrad (+ ReLU mask of the previous layer)
template <int KS>      // k-step pairs: ceil(9 Co / 2) <= KS
__global__ void __launch_bounds__(256, 2) conv3x3_thin_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ mask,
                                                                    float* __restrict__ dx, ThGeom g, int accumulate) {
    __shared__ float dp[(THD + 2) * PWD * 7];
    const int Co = g.Co, NC = 9 * Co;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // k = 2 kk + hi = (tap, co): B[k][ci = l31] = w[co][tap][ci]; A[q][k] = dY patch at (row + 2 - kh, col + 2 - kw)
    float bw[KS];
    int koff[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int k = 2 * kk + hi;
        const bool live = k < NC;
        const int tap = live ? k / Co : 0, co = live ? k - tap * Co : 0, kh = tap / 3, kw = tap - kh * 3;
        bw[kk] = w[((long)co * 9 + tap) * CI + l31];                      // unconditional (dead slots fetch element 0), zeroed below
        koff[kk] = live ? ((2 - kh) * PWD + (2 - kw)) * Co + co : 0;      // dead k slots: any finite patch value x a zero weight
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bw[kk] *= (2 * kk + hi < NC) ? 1.f : 0.f;
    DyPatch dpre;
    int tile = blockIdx.x;
    if (tile < g.ntiles) fetch_dy_patch(dpre, dy, g, Co, tile);
    for (; tile < g.ntiles; tile += gridDim.x) {
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * THD, w0 = (r % g.tiles_w) * 32;
        __syncthreads();                 // the previous tile's fragments are read
        store_dy_patch(dp, dpre, Co);
        __syncthreads();
        if (tile + (int)gridDim.x < g.ntiles) fetch_dy_patch(dpre, dy, g, Co, tile + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < THD / 4; ++rr) {
            const int row = wave * (THD / 4) + rr, h = h0 + row;
            if (h >= g.H) continue;      // wave-uniform
            const float* pa = dp + (row * PWD + l31) * Co;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) mfma_32x32x2(pa[koff[kk]], bw[kk], acc);
            const long obase = (((long)b * g.H + h) * g.W + w0) * CI + l31;
            if (mask && w0 + 32 <= g.W) {            // interior tile: the 16 mask values are fetched together, ahead of the stores
                float mv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) mv[e] = mask[obase + (long)acc_row(e, hi) * CI];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long o = obase + (long)acc_row(e, hi) * CI;
                    const float v = mv[e] > 0.f ? acc[e] : 0.f;
                    dx[o] = accumulate ? dx[o] + v : v;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (w0 + acc_row(e, hi) < g.W) {
                        const long o = obase + (long)acc_row(e, hi) * CI;
                        float v = acc[e];
                        if (mask && !(mask[o] > 0.f)) v = 0.f;
                        dx[o] = accumulate ? dx[o] + v : v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ wgrad (+ bias gradient)
// part[block][m = (tap, co) < 64][ci] + part_b[block][8]
constexpr int kThinPanel = 64 * 32 + 8;

template <int MT>
__global__ void __launch_bounds__(256, 2) conv3x3_thin_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                                    ThGeom g) {
    __shared__ float dp[(THD + 2) * PWD * 7 > 4 * 32 * 33 ? (THD + 2) * PWD * 7 : 4 * 32 * 33];
    const int Co = g.Co, NC = 9 * Co;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    int moff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 32 + l31;
        const bool live = m < NC;
        const int tap = live ? m / Co : 0, co = live ? m - tap * Co : 0, kh = tap / 3, kw = tap - kh * 3;
        moff[mt] = live ? ((2 - kh) * PWD + (2 - kw)) * Co + co : 0;      // dead rows: finite garbage, never read back
    }
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
    float bsum[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) bsum[c] = 0.f;
    DyPatch dpre;
    int tile = blockIdx.x;
    if (tile < g.ntiles) fetch_dy_patch(dpre, dy, g, Co, tile);
    for (; tile < g.ntiles; tile += gridDim.x) {
        const int b = tile / (g.tiles_h * g.tiles_w), r = tile - b * (g.tiles_h * g.tiles_w);
        const int h0 = (r / g.tiles_w) * THD, w0 = (r % g.tiles_w) * 32;
        __syncthreads();
        store_dy_patch(dp, dpre, Co);
        __syncthreads();
        if (tile + (int)gridDim.x < g.ntiles) fetch_dy_patch(dpre, dy, g, Co, tile + gridDim.x);
        {   // bias gradient: thread t owns interior pixel t of the tile (out-of-image pixels hold zeros)
            const float* pb = dp + (((tid >> 5) + 1) * PWD + (tid & 31) + 1) * Co;
#pragma unroll
            for (int c = 0; c < 7; ++c) if (c < Co) bsum[c] += pb[c];
        }
        const bool inner = w0 + 32 <= g.W;      // block-uniform: x loads need no column guard
#pragma unroll
        for (int rr = 0; rr < THD / 4; ++rr) {
            const int row = wave * (THD / 4) + rr, h = h0 + row;
            if (h >= g.H) continue;      // wave-uniform
            const float* xr = x + (((long)b * g.H + h) * g.W + w0) * CI + l31;
            float bx[16];
            if (inner) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) bx[kk] = xr[(long)(2 * kk + hi) * CI];
            } else {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int j = 2 * kk + hi;
                    bx[kk] = (w0 + j < g.W) ? xr[(long)j * CI] : 0.f;
                }
            }
            const float* pa = dp + (row * PWD + hi) * Co;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mfma_32x32x2(pa[moff[mt] + 2 * kk * Co], bx[kk], acc[mt]);
        }
    }
    // cross-wave reduction through LDS (red[wave][32][33]), one panel per block
    __syncthreads();
    float* red = dp;
    float* out = part + (long)blockIdx.x * kThinPanel;
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(wave * 32 + acc_row(e, hi)) * 33 + l31] = acc[mt][e];
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) {
            const int o = (i >> 5) * 33 + (i & 31);
            out[mt * 1024 + i] = (red[o] + red[32 * 33 + o]) + (red[2 * 32 * 33 + o] + red[3 * 32 * 33 + o]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const float s = wave_sum(bsum[c]);
        if (lane == 0) red[wave * 8 + c] = s;
    }
    __syncthreads();
    if (tid < 7) out[64 * 32 + tid] = (red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]);
}

// one WAVE per output element: the lanes stride over the blocks' panels (a serial loop over 512 panels per thread measured 247 us)
__global__ void __launch_bounds__(256) conv3x3_thin_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dw, float* __restrict__ db,
                                                                        int Co, int accumulate) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, NC = 9 * Co;
    const bool isw = i < NC * CI, isb = !isw && db && i - NC * CI < Co;
    const int src = isw ? i : 64 * 32 + (i - NC * CI);
    float s = 0.f;
    if (isw || isb)
        for (int b = lane; b < nblocks; b += 64) s += part[(long)b * kThinPanel + src];
    s = wave_sum(s);
    if (lane != 0) return;
    if (isw) {
        const int m = i >> 5, ci = i & 31, tap = m / Co, co = m - tap * Co;
        float* d = dw + ((long)co * 9 + tap) * CI + ci;
        *d = accumulate ? *d + s : s;
    } else if (isb) {
        db[i - NC * CI] += s;                        // bias gradients always accumulate (functions.bias_grad)
    }
}

inline bool thin_ok(const void* a, const void* b, const void* c, int B, int H, int W, int Cin, int Cout) {
    return a && b && c && B > 0 && H > 0 && W > 0 && Cin == CI && Cout >= 1 && Cout <= 7 && aligned16(a);
}

constexpr int kThinBlocks = 512;

}  // namespace

extern "C" int tf_conv3x3_thin_supported(int Cin, int Cout) { return (Cin == CI && Cout >= 1 && Cout <= 7) ? 1 : 0; }

extern "C" int tf_conv3x3_thin_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, void* stream) {
    TF_REQUIRE(thin_ok(x, w, y, B, H, W, Cin, Cout), "tf_conv3x3_thin_fwd_f32: needs Cin == 32, 1 <= Cout <= 7, 16-byte aligned x");
    ThGeom g; g.B = B; g.H = H; g.W = W; g.Co = Cout; g.tiles_w = cdiv(W, TWF);
    if (Cout == 1) {                      // one column tile, 9 floats per patch pixel: 14-row tiles (16 patch rows)
        g.tiles_h = cdiv(H, 14); g.ntiles = B * g.tiles_h * g.tiles_w;
        TF_LAUNCH((conv3x3_thin_fwd_kernel<1, 14, 1, 9>), dim3(g.ntiles < 1024 ? g.ntiles : 1024), dim3(256), stream, x, w, bias, y, g);
    } else {
        g.tiles_h = cdiv(H, 6); g.ntiles = B * g.tiles_h * g.tiles_w;
        const int grid = g.ntiles < kThinBlocks ? g.ntiles : kThinBlocks;
        if (Cout <= 3) TF_LAUNCH((conv3x3_thin_fwd_kernel<1, 6, 0, 27>), dim3(grid), dim3(256), stream, x, w, bias, y, g);
        else if (Cout == 7) TF_LAUNCH((conv3x3_thin_fwd_kernel<2, 6, 7, 63>), dim3(grid), dim3(256), stream, x, w, bias, y, g);
        else TF_LAUNCH((conv3x3_thin_fwd_kernel<2, 6, 0, 63>), dim3(grid), dim3(256), stream, x, w, bias, y, g);
    }
    return launch_status("tf_conv3x3_thin_fwd_f32");
}

extern "C" int tf_conv3x3_thin_dgrad_f32(const float* dy, const float* w, const float* relu_mask, float* dx, int B, int H, int W, int Cin, int Cout,
                                         int accumulate, void* stream) {
    TF_REQUIRE(thin_ok(dx, w, dy, B, H, W, Cin, Cout), "tf_conv3x3_thin_dgrad_f32: needs Cin == 32, 1 <= Cout <= 7");
    ThGeom g; g.B = B; g.H = H; g.W = W; g.Co = Cout; g.tiles_h = cdiv(H, THD); g.tiles_w = cdiv(W, 32); g.ntiles = B * g.tiles_h * g.tiles_w;
    const int grid = g.ntiles < 1024 ? g.ntiles : 1024;
    const int ks = (9 * Cout + 1) / 2;
    if (ks <= 5) TF_LAUNCH((conv3x3_thin_dgrad_kernel<5>), dim3(grid), dim3(256), stream, dy, w, relu_mask, dx, g, accumulate);
    else if (ks <= 16) TF_LAUNCH((conv3x3_thin_dgrad_kernel<16>), dim3(grid), dim3(256), stream, dy, w, relu_mask, dx, g, accumulate);
    else TF_LAUNCH((conv3x3_thin_dgrad_kernel<32>), dim3(grid), dim3(256), stream, dy, w, relu_mask, dx, g, accumulate);
    return launch_status("tf_conv3x3_thin_dgrad_f32");
}

extern "C" long tf_conv3x3_thin_wgrad_ws_floats(void) { return (long)kThinBlocks * kThinPanel; }

extern "C" int tf_conv3x3_thin_wgrad_f32(const float* dy, const float* x, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout, int accumulate,
                                         float* ws, void* stream) {
    TF_REQUIRE(thin_ok(x, dy, dw, B, H, W, Cin, Cout) && ws, "tf_conv3x3_thin_wgrad_f32: needs Cin == 32, 1 <= Cout <= 7 and ws of tf_conv3x3_thin_wgrad_ws_floats() floats");
    ThGeom g; g.B = B; g.H = H; g.W = W; g.Co = Cout; g.tiles_h = cdiv(H, THD); g.tiles_w = cdiv(W, 32); g.ntiles = B * g.tiles_h * g.tiles_w;
    const int grid = g.ntiles < kThinBlocks ? g.ntiles : kThinBlocks;
    if (9 * Cout <= 32) TF_LAUNCH((conv3x3_thin_wgrad_kernel<1>), dim3(grid), dim3(256), stream, dy, x, ws, g);
    else TF_LAUNCH((conv3x3_thin_wgrad_kernel<2>), dim3(grid), dim3(256), stream, dy, x, ws, g);
    const int nout = 9 * Cout * CI + Cout;
    TF_LAUNCH(conv3x3_thin_wgrad_reduce_kernel, dim3(cdiv(nout, 4)), dim3(256), stream, (const float*)ws, grid, dw, dbias, Cout, accumulate);
    return launch_status("tf_conv3x3_thin_wgrad_f32");
}

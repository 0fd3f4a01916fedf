// Small-M (M = batch rows <= 16) linear kernels: the SE excitation FCs of every RegNetY block, the join MLP and
// other per-sample matmuls (model.py:592-599; timm SEModule fc1/fc2).  A 128-row MFMA tile would be >90 % padding
// and latency-bound (~50 us per call, ~250 calls per step); these stream the weight once with all rows in registers.
#include "tf_common.h"

using namespace tf;

namespace tf {

constexpr int SM_MAXM = 16;

// y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n] (+ res[m][n])); one wave per output column n
__global__ void __launch_bounds__(256) smallm_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w, long ldw,
                                                         const float* __restrict__ bias, const float* __restrict__ res, long ldres,
                                                         float* __restrict__ y, long ldy, int M, int N, int K, int relu) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = n < N;
    const float* wr = w + (long)(live ? n : 0) * ldw;
    float acc[SM_MAXM];
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) acc[m] = 0.f;
    if (live)
        for (int k = lane; k < K; k += 64) {
            const float wv = wr[k];
#pragma unroll
            for (int m = 0; m < SM_MAXM; ++m)
                if (m < M) acc[m] += x[m * ldx + k] * wv;
        }
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m)
        if (m < M) acc[m] = wave_sum(acc[m]);
    if (live && lane == 0) {
        const float b = bias ? bias[n] : 0.f;
        for (int m = 0; m < M; ++m) {
            float v = acc[m] + b;
            if (res) v += res[m * ldres + n];
            if (relu) v = fmaxf(v, 0.f);
            y[m * ldy + n] = v;
        }
    }
}

// dx[m][k] (+)= sum_n dy[m][n] * w[n][k] (+ res): block = 32 k-columns x 8 n-slices of ONE n-chunk (grid.y chunks);
// partial sums are combined through LDS and, across chunks, with fp32 atomics into a destination that the first chunk
// initialises (res / old value / 0) - so the grid has K/32 x N/256 blocks instead of K/32.
__global__ void __launch_bounds__(256) smallm_dgrad_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ w, long ldw,
                                                           float* __restrict__ dx, long lddx, int M, int N, int K, int nchunk) {
    __shared__ float red[8][SM_MAXM][32];
    const int kx = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kx;
    const int n0 = blockIdx.y * nchunk;
    const int n1 = (n0 + nchunk < N) ? n0 + nchunk : N;
    float acc[SM_MAXM];
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) acc[m] = 0.f;
    if (k < K)
        for (int n = n0 + sl; n < n1; n += 8) {
            const float wv = w[(long)n * ldw + k];
#pragma unroll
            for (int m = 0; m < SM_MAXM; ++m)
                if (m < M) acc[m] += dy[m * lddy + n] * wv;
        }
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) red[sl][m][kx] = acc[m];
    __syncthreads();
    for (int m = sl; m < M; m += 8) {
        if (k < K) {
            float v = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) v += red[s][m][kx];
            atomicAdd(dx + m * lddx + k, v);
        }
    }
}
// dx = res (or 0, or unchanged when accumulating) before the atomics of smallm_dgrad_kernel
__global__ void __launch_bounds__(256) smallm_init_kernel(float* __restrict__ dx, long lddx, const float* __restrict__ res, long ldres, int M, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * K) return;
    const int m = i / K, k = i - m * K;
    dx[m * lddx + k] = res ? res[m * ldres + k] : 0.f;
}

// dw[n][k] (+)= sum_m dy[m][n] * x[m][k]; one thread per (n, k), k fastest
__global__ void __launch_bounds__(256) smallm_wgrad_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                           float* __restrict__ dw, long lddw, int M, int N, int K, int accumulate) {
    const long total = (long)N * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        float v = 0.f;
        for (int m = 0; m < M; ++m) v += dy[m * lddy + n] * x[m * ldx + k];
        float* d = dw + (long)n * lddw + k;
        *d = accumulate ? *d + v : v;
    }
}

// ---- "skinny" weight gradient: few output features over MANY rows (the 1x1 head convolutions 64 -> 1 / 2 / 3 / 12 over B * 64 * 64
// pixels, model.py:93-99: 10 MB of operands).  Through the MFMA engine this is one 128 x 32 tile split 160 ways over K = 47 us per head,
// six heads per step; here it is a streaming reduction: a block walks a row chunk, thread (tx, ty) holds dW[0..NO)[4 tx .. 4 tx + 3] for the rows
// ty, ty + RL, ...; the row lanes are combined through LDS and the block adds its partial to dW with fp32 atomics (as the engine's split-K
// weight gradients do).  dw[o][c] += sum_r dy[r][o] * x[r][c];  C % 4 == 0, C <= 256, NO <= 16.
template <int NO>
__global__ void __launch_bounds__(256) skinny_wgrad_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, float* __restrict__ dw,
                                                           long lddw, int rows, int no, int C, int rows_per_block) {
    __shared__ float4 red[256];
    const int cv = C >> 2, RL = 256 / cv;
    const int tx = threadIdx.x % cv, ty = threadIdx.x / cv;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float4 acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ty < RL) {
        int r = r0 + ty;
        for (; r + 3 * RL < r1; r += 4 * RL) {          // four rows in flight per thread
            float4 xv[4];
            float dv[4][NO];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = *reinterpret_cast<const float4*>(x + (long)(r + u * RL) * ldx + 4 * tx);
                const float* g = dy + (long)(r + u * RL) * lddy;
#pragma unroll
                for (int o = 0; o < NO; ++o) dv[u][o] = g[o < no ? o : 0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const float d = dv[u][o];
                    acc[o].x += d * xv[u].x; acc[o].y += d * xv[u].y; acc[o].z += d * xv[u].z; acc[o].w += d * xv[u].w;
                }
        }
        for (; r < r1; r += RL) {
            const float4 xv = *reinterpret_cast<const float4*>(x + (long)r * ldx + 4 * tx);
            const float* g = dy + (long)r * lddy;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const float d = g[o < no ? o : 0];
                acc[o].x += d * xv.x; acc[o].y += d * xv.y; acc[o].z += d * xv.z; acc[o].w += d * xv.w;
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        __syncthreads();
        red[threadIdx.x] = acc[o];
        __syncthreads();
        if (ty == 0 && o < no) {
            float4 t = red[tx];
            for (int j = 1; j < RL; ++j) { const float4 v = red[j * cv + tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            float* d = dw + (long)o * lddw + 4 * tx;
            atomicAdd(d, t.x); atomicAdd(d + 1, t.y); atomicAdd(d + 2, t.z); atomicAdd(d + 3, t.w);
        }
    }
}
bool skinny_wgrad_ok(int no, int C, int rows, long lddy, long ldx, const float* x, int accumulate) {
    return accumulate && no >= 1 && no <= 16 && C % 4 == 0 && C >= 16 && C <= 256 && rows >= 4096 && ldx % 4 == 0 && aligned16(x) && lddy >= no;
}
int skinny_wgrad(const float* dy, long lddy, const float* x, long ldx, float* dw, long lddw, int rows, int no, int C, void* stream) {
    // every block ends with NO x C atomics on the SAME addresses, which the memory side serialises (320 blocks: 36 us, mostly that queue):
    // few, fat blocks
    int blocks = cdiv(rows, 512);
    if (blocks > 96) blocks = 96;
    const int rpb = cdiv(rows, blocks);
    blocks = cdiv(rows, rpb);
#define TF_SK(NO_) TF_LAUNCH(skinny_wgrad_kernel<NO_>, dim3(blocks), dim3(256), stream, dy, lddy, x, ldx, dw, lddw, rows, no, C, rpb)
    if (no <= 1) TF_SK(1); else if (no <= 2) TF_SK(2); else if (no <= 4) TF_SK(4); else if (no <= 8) TF_SK(8); else TF_SK(16);
#undef TF_SK
    return launch_status("tf_gemm_f32[skinny wgrad]");
}

int smallm_fwd(const float* x, long ldx, const float* w, long ldw, const float* bias, const float* res, long ldres, float* y, long ldy, int M, int N,
               int K, int relu, void* stream) {
    TF_LAUNCH(smallm_fwd_kernel, dim3(cdiv(N, 4)), dim3(256), stream, x, ldx, w, ldw, bias, res, ldres, y, ldy, M, N, K, relu);
    return launch_status("tf_gemm_f32[small-m fwd]");
}
int smallm_dgrad(const float* dy, long lddy, const float* w, long ldw, const float* res, long ldres, float* dx, long lddx, int M, int N, int K,
                 int accumulate, void* stream) {
    if (!accumulate) TF_LAUNCH(smallm_init_kernel, dim3(cdiv((long)M * K, 256)), dim3(256), stream, dx, lddx, res, ldres, M, K);
    const int nchunk = 256;
    TF_LAUNCH(smallm_dgrad_kernel, dim3(cdiv(K, 32), cdiv(N, nchunk)), dim3(256), stream, dy, lddy, w, ldw, dx, lddx, M, N, K, nchunk);
    return launch_status("tf_gemm_f32[small-m dgrad]");
}
int smallm_wgrad(const float* dy, long lddy, const float* x, long ldx, float* dw, long lddw, int M, int N, int K, int accumulate, void* stream) {
    long blocks = ((long)N * K + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    TF_LAUNCH(smallm_wgrad_kernel, dim3((unsigned)blocks), dim3(256), stream, dy, lddy, x, ldx, dw, lddw, M, N, K, accumulate);
    return launch_status("tf_gemm_f32[small-m wgrad]");
}

}  // namespace tf

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

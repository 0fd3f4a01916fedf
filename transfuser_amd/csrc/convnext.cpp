// ConvNeXt trunk pieces (timm 0.5.4 ``convnext_*``; the re-labelling branch of the reference's ImageCNN / LidarEncoder,
// transfuser.py:395-416,457-471 - SURVEY.md 8f-4).  What the block needs beyond the existing kernels (LayerNorm over NHWC rows, the
// Linear GEMMs, column sums):
//   depthwise 7x7 / pad 3 convolution (bias) forward, input gradient (the same kernel on the flipped taps) and weight / bias gradient;
//   exact (erf) GELU forward / backward;  y = res + gamma[c] x (layer scale + shortcut) and x += bias[c] (the patchify stem's bias).
// All HBM / L2-bound element-wise or stencil work: one thread per 4 channels of a pixel, the 49-tap weight panel of a 64-channel group in
// LDS ([tap][channel]: the parameter is stored [channel][tap]).  No MFMA: a depthwise convolution has no contraction to put on it.
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int KS = 7, TAPS = 49, CGRP = 64;      // channels per block column group

inline int ew_blocks(long n, int cap = 4096) {
    long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// y[b][h][w][c] = bias[c] + sum_{kh,kw} x[b][h+kh-3][w+kw-3][c] * wt[c][kh][kw]   (flip != 0: taps mirrored = the input gradient, no bias)
__global__ void __launch_bounds__(256) dwconv7_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                      float* __restrict__ y, int B, int H, int W, int C, int flip, int accumulate) {
    __shared__ float ws[TAPS][CGRP];
    const int c0 = blockIdx.y * CGRP, tid = threadIdx.x;
    for (int i = tid; i < TAPS * CGRP; i += 256) {
        const int cc = i % CGRP, t = i / CGRP;
        ws[t][cc] = (c0 + cc < C) ? wt[(long)(c0 + cc) * TAPS + (flip ? TAPS - 1 - t : t)] : 0.f;
    }
    __syncthreads();
    const int cq = tid & 15, c = c0 + cq * 4;          // 16 channel quads x 16 pixels per block trip
    if (c >= C) return;
    const long npix = (long)B * H * W;
    for (long p = (long)blockIdx.x * 16 + (tid >> 4); p < npix; p += (long)gridDim.x * 16) {
        const int w = (int)(p % W);
        const long t = p / W;
        const int h = (int)(t % H);
        const long b = t / H;
        float4 acc = (bias && !flip) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const int hh = h + kh - 3;
            if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int ww = w + kw - 3;
                if ((unsigned)ww >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + hh) * W + ww) * C + c);
                const float* k = &ws[kh * KS + kw][cq * 4];
                acc.x += v.x * k[0]; acc.y += v.y * k[1]; acc.z += v.z * k[2]; acc.w += v.w * k[3];
            }
        }
        float4* dst = reinterpret_cast<float4*>(y + p * C + c);
        if (accumulate) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *dst = acc;
    }
}

// dwt[c][kh][kw] += sum_p dy[p][c] x[p + (kh-3, kw-3)][c];  dbias[c] += sum_p dy[p][c].  One block = one channel group x a slice of the pixels x ONE
// tap row kh (7 accumulators per thread); the block's 16 pixel lanes are combined through LDS, the slices through fp32 atomics.
__global__ void __launch_bounds__(256) dwconv7_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dwt,
                                                            float* __restrict__ dbias, int B, int H, int W, int C) {
    __shared__ float red[16][CGRP][8];
    const int c0 = blockIdx.y * CGRP, tid = threadIdx.x, kh = blockIdx.z;
    const int cq = tid & 15, pl = tid >> 4, c = c0 + cq * 4;
    float acc[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
    const long npix = (long)B * H * W;
    if (c < C)
        for (long p = (long)blockIdx.x * 16 + pl; p < npix; p += (long)gridDim.x * 16) {
            const int w = (int)(p % W);
            const long t = p / W;
            const int h = (int)(t % H);
            const long b = t / H;
            const float4 g = *reinterpret_cast<const float4*>(dy + p * C + c);
            if (kh == 3) { acc[7][0] += g.x; acc[7][1] += g.y; acc[7][2] += g.z; acc[7][3] += g.w; }      // bias gradient rides on the centre row
            const int hh = h + kh - 3;
            if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int ww = w + kw - 3;
                if ((unsigned)ww >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + hh) * W + ww) * C + c);
                acc[kw][0] += g.x * v.x; acc[kw][1] += g.y * v.y; acc[kw][2] += g.z * v.z; acc[kw][3] += g.w * v.w;
            }
        }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[pl][cq * 4 + e][k] = acc[k][e];
    __syncthreads();
    for (int i = tid; i < CGRP * 8; i += 256) {
        const int cc = i >> 3, k = i & 7;
        if (c0 + cc >= C) continue;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j][cc][k];
        if (k < KS) atomicAdd(dwt + (long)(c0 + cc) * TAPS + kh * KS + k, s);
        else if (kh == 3 && dbias) atomicAdd(dbias + c0 + cc, s);
    }
}

__global__ void __launch_bounds__(256) gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}
// dx = dy * d/dx gelu(x) = dy * (Phi(x) + x phi(x))
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}
// y = (res ? res : 0) + gamma[c] * x + (beta ? beta[c] : 0)
__global__ void __launch_bounds__(256) colscale_add_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ res, float* __restrict__ y, long n, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = (gamma ? gamma[c] : 1.f) * x[i];
        if (beta) v += beta[c];
        if (res) v += res[i];
        y[i] = v;
    }
}

}  // namespace

extern "C" int tf_dwconv7_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int flip, int accumulate, void* stream) {
    TF_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(y) && (!bias || aligned16(bias)),
               "tf_dwconv7_fwd_f32: needs NHWC tensors with C %% 4 == 0, 16-byte aligned");
    const long npix = (long)B * H * W;
    long bx = (npix + 15) / 16;
    if (bx > 2048) bx = 2048;
    TF_LAUNCH(dwconv7_kernel, dim3((int)bx, cdiv(C, CGRP)), dim3(256), stream, x, w, bias, y, B, H, W, C, flip, accumulate);
    return launch_status("tf_dwconv7_fwd_f32");
}
extern "C" int tf_dwconv7_wgrad_f32(const float* dy, const float* x, float* dw, float* dbias, int B, int H, int W, int C, void* stream) {
    TF_REQUIRE(dy && x && dw && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(dy), "tf_dwconv7_wgrad_f32: needs C %% 4 == 0, 16-byte aligned");
    const long npix = (long)B * H * W;
    long bx = (npix + 16 * 64 - 1) / (16 * 64);      // >= 64 pixels per pixel lane
    if (bx > 128) bx = 128;
    if (bx < 1) bx = 1;
    TF_LAUNCH(dwconv7_wgrad_kernel, dim3((int)bx, cdiv(C, CGRP), KS), dim3(256), stream, dy, x, dw, dbias, B, H, W, C);
    return launch_status("tf_dwconv7_wgrad_f32");
}
extern "C" int tf_gelu_fwd_f32(const float* x, float* y, int64_t n, void* stream) {
    TF_REQUIRE(x && y && n >= 0, "tf_gelu_fwd_f32: bad arguments");
    if (n) TF_LAUNCH(gelu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), stream, x, y, (long)n);
    return launch_status("tf_gelu_fwd_f32");
}
extern "C" int tf_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    TF_REQUIRE(dy && x && dx && n >= 0, "tf_gelu_bwd_f32: bad arguments");
    if (n) TF_LAUNCH(gelu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), stream, dy, x, dx, (long)n);
    return launch_status("tf_gelu_bwd_f32");
}
extern "C" int tf_colscale_add_f32(const float* x, const float* gamma, const float* beta, const float* res, float* y, int64_t rows, int C, void* stream) {
    TF_REQUIRE(x && y && rows >= 0 && C > 0, "tf_colscale_add_f32: bad arguments");
    const long n = (long)rows * C;
    if (n) TF_LAUNCH(colscale_add_kernel, dim3(ew_blocks(n)), dim3(256), stream, x, gamma, beta, res, y, n, C);
    return launch_status("tf_colscale_add_f32");
}

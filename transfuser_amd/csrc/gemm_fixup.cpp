// Second pass of the deterministic two-pass split-K (tf_gemm_dma.h / tf_gemm_engine.h: kTwoPass): the k-slices of a GEMM whose output has too
// few tiles for the 256 CUs stored raw partial tiles ws[s][M][ldws]; here they are summed IN SLICE ORDER (bitwise reproducible, unlike the
// atomic split-K of the weight gradients) and the GEMM epilogue proper is applied: alpha, bias, residual, ReLU, mask, store / +=.
// HBM-bound: (S + res + mask) reads + one write of the M x N output, float4 where the output allows it.
#include "tf_gemm_engine.h"

namespace tf {
namespace {

template <bool V4>
__global__ void __launch_bounds__(256) splitk_fixup_kernel(const float* __restrict__ ws, int nsplit, long sk_stride, int ldws, GemmEpi ep, int M, int N) {
    constexpr int V = V4 ? 4 : 1;
    const int nq = (N + V - 1) / V;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * nq) return;
    const int i = (int)(idx / nq), j = (int)(idx - (long)i * nq) * V;
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    const float* p = ws + (long)i * ldws + j;
    for (int s = 0; s < nsplit; ++s) {
        if (V4) {
            const float4 t = *reinterpret_cast<const float4*>(p + (long)s * sk_stride);
            v[0] += t.x; v[1 % V] += t.y; v[2 % V] += t.z; v[3 % V] += t.w;
        } else v[0] += p[(long)s * sk_stride];
    }
    float* dst = ep.C + (long)i * ep.ldc + j;
    const uint32_t dseed = ep.drop_seed ? *ep.drop_seed : 0u;
#pragma unroll
    for (int e = 0; e < V; ++e) {
        float x = ep.alpha * v[e] + (ep.bias ? ep.bias[j + e] : 0.f);
        if (ep.drop_seed) x = dropout_keep(dseed, ep.drop_site, (uint32_t)((long)i * N + j + e), ep.drop_thresh) ? x * ep.drop_scale : 0.f;      // see GemmEpi.drop_seed
        if (ep.res) x += ep.res[(long)i * ep.ldres + j + e];
        x = ep.relu ? fmaxf(x, 0.f) : x;
        if (ep.mask) x = (ep.mask[(long)i * ep.ldmask + j + e] > 0.f) ? x : 0.f;
        v[e] = x;
    }
    if (V4) {
        float4 o = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
        if (ep.mode == 1) { const float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
        *reinterpret_cast<float4*>(dst) = o;
    } else {
        *dst = ep.mode == 1 ? *dst + v[0] : v[0];
    }
}

}  // namespace

void launch_splitk_fixup(const float* ws, int nsplit, long sk_stride, int ldws, const GemmEpi& ep, int M, int N, void* stream) {
    const bool v4 = N % 4 == 0 && ep.ldc % 4 == 0 && aligned16(ep.C) && aligned16(ws) && (!ep.bias || aligned16(ep.bias)) &&
                    (!ep.res || (ep.ldres % 4 == 0 && aligned16(ep.res))) && (!ep.mask || (ep.ldmask % 4 == 0 && aligned16(ep.mask)));
    const long n = (long)M * (v4 ? N / 4 : N);
    if (v4) TF_LAUNCH(splitk_fixup_kernel<true>, dim3(cdiv(n, 256)), dim3(256), stream, ws, nsplit, sk_stride, ldws, ep, M, N);
    else TF_LAUNCH(splitk_fixup_kernel<false>, dim3(cdiv(n, 256)), dim3(256), stream, ws, nsplit, sk_stride, ldws, ep, M, N);
}

}  // namespace tf

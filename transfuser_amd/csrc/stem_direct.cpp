// Direct kernels for the RegNet stem convolutions on the NCHW model inputs (transfuser.py:136-143 via timm: 3x3 / stride 2, 3 + 0 or 2 + 1 input
// channels -> 32, no bias; normalize_imagenet folded into the image load, transfuser.py:419-428).
//
// Through the implicit-GEMM engine these two launches per trunk were bound by the im2col gather of an NCHW source (scalar loads with per-element
// index arithmetic and a validity predicate in front of every load): forward 105 + 30 us, weight gradient 179 + 75 us per step for 80 MB of
// traffic.  Here K = ks^2 Cin <= 32 fits ONE MFMA column block, so a wave owns 32 output pixels (forward) or a run of pixels (weight gradient)
// and every lane gathers its own (pixel, k) element with an UNCONDITIONAL load from a clamped address - the padding test is applied to the loaded
// value - so a lane's 14 gathers are in flight together:
//   forward   y[p][co]   = sum_k patch[p][k] W[co][k]   A = patch (lane: pixel, k = 2 kk + hi), B = W held in 16 registers per lane
//   wgrad     dW[co][k] += sum_p dY[p][co] patch[p][k]  A = dY (128-byte rows), B = patch (lane: fixed k, pixel = 2 kk + hi);
//             per-block partial panels + a wave-parallel reduce kernel in a fixed order (deterministic, no atomics)
#include "tf_common.h"
#include <stdlib.h>
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int SK = 32;                 // padded K (and Cout)
constexpr int kWgPix = 256;            // pixels per wave and partial panel of the weight gradient

struct StemSrc {
    const float* s0; const float* s1;
    int C0, C1, Hi, Wi, Ho, Wo, ks, stride, pad, Cin, K, normalize;
    long npix;                         // B * Ho * Wo
    float mean[4], stdv[4];
};

// value of patch element (pixel p, column k) - 0 outside the image / beyond K; the load itself is unconditional (clamped coordinates)
__device__ __forceinline__ float stem_gather(const StemSrc& g, long p, int k) {
    const bool kin = k < g.K && p < g.npix;
    const long pp = p < g.npix ? p : 0;
    const int kk = k < g.K ? k : 0;
    const int ox = (int)(pp % g.Wo);
    const long t = pp / g.Wo;
    const int oy = (int)(t % g.Ho), b = (int)(t / g.Ho);
    const int tap = kk / g.Cin, ci = kk - tap * g.Cin;
    const int kh = tap / g.ks, kw = tap - kh * g.ks;
    const int ih = oy * g.stride - g.pad + kh, iw = ox * g.stride - g.pad + kw;
    const bool ok = kin && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi;
    const int ihc = ih < 0 ? 0 : (ih > g.Hi - 1 ? g.Hi - 1 : ih), iwc = iw < 0 ? 0 : (iw > g.Wi - 1 ? g.Wi - 1 : iw);
    const bool first = ci < g.C0 || !g.s1;
    const float* src = first ? g.s0 : g.s1;
    const int cs = first ? ci : ci - g.C0, Cn = first ? g.C0 : g.C1;
    float v = src[(((long)b * Cn + cs) * g.Hi + ihc) * g.Wi + iwc];
    if (g.normalize) v = ((v / 255.0f) - g.mean[ci & 3]) / g.stdv[ci & 3];
    return ok ? v : 0.f;
}

__global__ void __launch_bounds__(256) stem_direct_fwd_kernel(StemSrc g, const float* __restrict__ w, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const long nwaves = (long)gridDim.x * 4, wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float wf[SK / 2];                  // B fragment: W[co = l31][k = 2 kk + hi]
#pragma unroll
    for (int kk = 0; kk < SK / 2; ++kk) {
        const int k = 2 * kk + hi;
        wf[kk] = k < g.K ? w[(long)l31 * g.K + k] : 0.f;
    }
    const long ntiles = (g.npix + 31) / 32;
    for (long tile = wave0; tile < ntiles; tile += nwaves) {
        const long p0 = tile * 32;
        float a[SK / 2];
#pragma unroll
        for (int kk = 0; kk < SK / 2; ++kk) a[kk] = stem_gather(g, p0 + l31, 2 * kk + hi);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < SK / 2; ++kk) mfma_32x32x2(a[kk], wf[kk], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long p = p0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (p < g.npix) y[p * SK + l31] = acc[r];
        }
    }
}

// partial panel of block-wave w: part[w][co][k] over its kWgPix pixels
__global__ void __launch_bounds__(256) stem_direct_wgrad_kernel(StemSrc g, const float* __restrict__ dy, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long p0 = wv * kWgPix;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int q = 0; q < kWgPix; q += 16) {             // 8 MFMA steps (16 pixels) per batch: 16 independent loads per lane in flight
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long p = p0 + q + 2 * u + hi;
            a[u] = dy[(p < g.npix ? p : 0) * SK + l31];
            if (p >= g.npix) a[u] = 0.f;
            b[u] = stem_gather(g, p, l31);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) mfma_32x32x2(a[u], b[u], acc);
    }
    float* o = part + wv * (SK * SK);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * hi) * SK + l31] = acc[r];
}
// dw[co][k] (+)= sum over the partial panels, one wave per output element, lanes stride over the panels + a fixed shuffle tree
__global__ void __launch_bounds__(256) stem_direct_wgrad_reduce_kernel(const float* __restrict__ part, long npanels, int K, float* __restrict__ dw, int accumulate) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // e = co * K + k over 32 x K
    const bool live = e < SK * K;
    const int co = live ? e / K : 0, k = live ? e - co * K : 0;
    float s = 0.f;
    if (live)
        for (long j = lane; j < npanels; j += 64) s += part[j * (SK * SK) + co * SK + k];
    s = wave_sum(s);
    if (live && lane == 0) dw[e] = accumulate ? dw[e] + s : s;
}

}  // namespace

namespace tf {
int gemm_precision();

bool stem_direct_ok(const tf_conv_geom* g, int C0, int C1) {
    static const bool on = [] { const char* e = getenv("TF_STEM_DIRECT"); return e ? e[0] != '0' : true; }();
    const int prec = gemm_precision();
    return on && (prec == 0 || prec == 2) && g->Cout == SK && g->groups == 1 && g->ksize * g->ksize * (C0 + C1) <= SK && C0 + C1 <= 4;
}
long stem_direct_wgrad_ws_floats(const tf_conv_geom* g) {
    const long npix = (long)g->B * g->Ho * g->Wo;
    return ((npix + kWgPix - 1) / kWgPix + 4) * (SK * SK);
}
static StemSrc make_src(const tf_conv_geom* g, const float* s0, int C0, const float* s1, int C1, int normalize) {
    StemSrc s;
    s.s0 = s0; s.s1 = s1; s.C0 = C0; s.C1 = C1; s.Hi = g->Hi; s.Wi = g->Wi; s.Ho = g->Ho; s.Wo = g->Wo; s.ks = g->ksize; s.stride = g->stride; s.pad = g->pad;
    s.Cin = C0 + C1; s.K = g->ksize * g->ksize * s.Cin; s.normalize = normalize; s.npix = (long)g->B * g->Ho * g->Wo;
    const float mean[4] = {0.485f, 0.456f, 0.406f, 0.f}, stdv[4] = {0.229f, 0.224f, 0.225f, 1.f};
    for (int i = 0; i < 4; ++i) { s.mean[i] = mean[i]; s.stdv[i] = stdv[i]; }
    return s;
}
int stem_direct_fwd(const tf_conv_geom* g, const float* s0, int C0, const float* s1, int C1, int normalize, const float* w, float* y, void* stream) {
    const StemSrc s = make_src(g, s0, C0, s1, C1, normalize);
    long blocks = ((s.npix + 31) / 32 + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    TF_LAUNCH(stem_direct_fwd_kernel, dim3((unsigned)blocks), dim3(256), stream, s, w, y);
    return launch_status("tf_stem_conv_fwd_f32[direct]");
}
int stem_direct_wgrad(const tf_conv_geom* g, const float* dy, const float* s0, int C0, const float* s1, int C1, int normalize, float* dw, int accumulate,
                      float* ws, void* stream) {
    const StemSrc s = make_src(g, s0, C0, s1, C1, normalize);
    const long nwaves = (s.npix + kWgPix - 1) / kWgPix, blocks = (nwaves + 3) / 4;
    TF_LAUNCH(stem_direct_wgrad_kernel, dim3((unsigned)blocks), dim3(256), stream, s, dy, ws);
    TF_LAUNCH(stem_direct_wgrad_reduce_kernel, dim3((unsigned)((SK * s.K + 3) / 4)), dim3(256), stream, (const float*)ws, blocks * 4, s.K, dw, accumulate);
    return launch_status("tf_stem_conv_wgrad_f32[direct]");
}
}  // namespace tf

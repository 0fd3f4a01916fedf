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

// Index arithmetic is the cost of these kernels, not the loads: everything that depends on the column k alone (tap, channel, source plane,
// normalisation constants) is decoded ONCE per block into LDS, pixels are decoded with 32-bit divisions once per tile (forward) or once per
// 16-pixel batch and then stepped (weight gradient).  (The first version decoded pixel and column with 64-bit divisions inside every gather and
// was no faster than the engine's im2col loader: 176 / 102 us.)
struct KCol { int kh, kw, ci; float mean, stdv; };          // kh < 0: column beyond K
__device__ __forceinline__ void stem_fill_cols(const StemSrc& g, KCol* kc) {
    if (threadIdx.x < SK) {
        const int k = threadIdx.x;
        KCol c;
        const int tap = k / g.Cin;
        c.ci = k - tap * g.Cin;
        c.kh = k < g.K ? tap / g.ks : -1;
        c.kw = tap - (tap / g.ks) * g.ks;
        c.mean = g.mean[c.ci & 3]; c.stdv = g.stdv[c.ci & 3];
        kc[k] = c;
    }
    __syncthreads();
}
// value of patch element (pixel (b, oy, ox), column c) - 0 outside the image / beyond K / for a dead pixel; the load itself is unconditional
__device__ __forceinline__ float stem_gather(const StemSrc& g, int b, int oy, int ox, bool plive, const KCol& c) {
    const int ih = oy * g.stride - g.pad + c.kh, iw = ox * g.stride - g.pad + c.kw;
    const bool ok = plive && c.kh >= 0 && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi;
    const int ihc = ih < 0 ? 0 : (ih > g.Hi - 1 ? g.Hi - 1 : ih), iwc = iw < 0 ? 0 : (iw > g.Wi - 1 ? g.Wi - 1 : iw);
    const bool first = c.ci < g.C0 || !g.s1;
    const float* src = first ? g.s0 : g.s1;
    const int cs = first ? c.ci : c.ci - g.C0, Cn = first ? g.C0 : g.C1;
    float v = src[(((long)b * Cn + cs) * g.Hi + ihc) * g.Wi + iwc];
    if (g.normalize) v = ((v / 255.0f) - c.mean) / c.stdv;
    return ok ? v : 0.f;
}

__global__ void __launch_bounds__(256) stem_direct_fwd_kernel(StemSrc g, const float* __restrict__ w, float* __restrict__ y) {
    __shared__ KCol kc[SK];
    stem_fill_cols(g, kc);
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int nwaves = (int)gridDim.x * 4, wave0 = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    float wf[SK / 2];                  // B fragment: W[co = l31][k = 2 kk + hi]
#pragma unroll
    for (int kk = 0; kk < SK / 2; ++kk) {
        const int k = 2 * kk + hi;
        wf[kk] = k < g.K ? w[(long)l31 * g.K + k] : 0.f;
    }
    const int npix = (int)g.npix, ntiles = (npix + 31) / 32;
    for (int tile = wave0; tile < ntiles; tile += nwaves) {
        const int p0 = tile * 32, p = p0 + l31;
        const bool plive = p < npix;
        const int pp = plive ? p : 0;
        const int ox = pp % g.Wo, t = pp / g.Wo, oy = t % g.Ho, b = t / g.Ho;
        float a[SK / 2];
#pragma unroll
        for (int kk = 0; kk < SK / 2; ++kk) a[kk] = stem_gather(g, b, oy, ox, plive, kc[2 * kk + hi]);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < SK / 2; ++kk) mfma_32x32x2(a[kk], wf[kk], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = p0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (q < npix) y[(long)q * SK + l31] = acc[r];
        }
    }
}

// partial panel of block-wave w: part[w][co][k] over its kWgPix pixels
__global__ void __launch_bounds__(256) stem_direct_wgrad_kernel(StemSrc g, const float* __restrict__ dy, float* __restrict__ part) {
    __shared__ KCol kc[SK];
    stem_fill_cols(g, kc);
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int npix = (int)g.npix;
    const long p0 = (long)wv * kWgPix;
    const KCol c = kc[l31];            // this lane's column: fixed
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int q = 0; q < kWgPix; q += 16) {             // 8 MFMA steps (16 pixels) per batch: 16 independent loads per lane in flight
        const long pb = p0 + q + hi;                   // this lane's first pixel of the batch, then + 2 per step
        const int pp = pb < npix ? (int)pb : 0;
        int ox = pp % g.Wo, t = pp / g.Wo, oy = t % g.Ho, b = t / g.Ho;
        float a[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long p = pb + 2 * u;
            const bool plive = p < npix;
            a[u] = dy[(plive ? p : 0) * SK + l31];
            if (!plive) a[u] = 0.f;
            bv[u] = stem_gather(g, b, oy, ox, plive, c);
            if (p + 2 < npix) {                       // never step past the last pixel: the (unconditional) gather of a dead pixel must stay inside the tensor
                ox += 2;
                while (ox >= g.Wo) { ox -= g.Wo; if (++oy >= g.Ho) { oy = 0; ++b; } }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) mfma_32x32x2(a[u], bv[u], acc);
    }
    float* o = part + (long)wv * (SK * SK);
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
    return on && (prec == 0 || prec == 2) && g->Cout == SK && g->groups == 1 && g->ksize * g->ksize * (C0 + C1) <= SK && C0 + C1 <= 4 &&
           (long)g->B * g->Ho * g->Wo < (1L << 30) && (long)g->B * 4 * g->Hi * g->Wi < (1L << 31);
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

// Spatial resampling kernels (HBM-bound):
//  - AdaptiveAvgPool2d -> token pack (+pos_emb [+vel_emb]) in one pass (transfuser.py:150-151,346-357)
//  - its backward
//  - bilinear interpolation fwd/bwd with arbitrary element strides, so the SAME kernel reads the
//    GPT's raw-viewed (B,C,h,w) token memory (quirk Q1, transfuser.py:363-364) and writes / adds
//    into NHWC feature maps (transfuser.py:154-157), nn.Upsample x2 (transfuser.py:103,114-116),
//    decoder x8/x4 (transfuser.py:241,243) and align_corners=True pred_bev (model.py:760).
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

// PyTorch adaptive pooling window: [floor(i*In/Out), ceil((i+1)*In/Out))
__device__ __forceinline__ int ap_start(int i, int in, int out) { return (int)(((long)i * in) / out); }
__device__ __forceinline__ int ap_end(int i, int in, int out) { return (int)(((long)(i + 1) * in + out - 1) / out); }

template <int V>
__global__ void __launch_bounds__(256) pool_tokens_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int oh, int ow,
                                                              const float* __restrict__ pos, const float* __restrict__ bvec,
                                                              float* __restrict__ tok, int T_total, int tok_off) {
    const int cv = C / V;
    const long total = (long)B * oh * ow * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * V;
        long t = idx / cv;
        const int j = (int)(t % ow); t /= ow;
        const int i = (int)(t % oh);
        const int b = (int)(t / oh);
        const int h0 = ap_start(i, H, oh), h1 = ap_end(i, H, oh), w0 = ap_start(j, W, ow), w1 = ap_end(j, W, ow);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int wb = w0; wb < w1; wb += 8) {        // 8 window columns per trip from clamped indices: the loads are in flight together
                const float* prow = x + (((long)b * H + h) * W) * C + c;
                if (V == 4) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(prow + (long)(wb + u < w1 ? wb + u : w1 - 1) * C);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (wb + u < w1) { acc[0] += v[u].x; acc[1 % V] += v[u].y; acc[2 % V] += v[u].z; acc[3 % V] += v[u].w; }
                } else {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = prow[(long)(wb + u < w1 ? wb + u : w1 - 1) * C];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (wb + u < w1) acc[0] += v[u];
                }
            }
        const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
        const int trow = tok_off + i * ow + j;
        float* o = tok + ((long)b * T_total + trow) * C + c;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = acc[k] * inv;
            if (pos) v += pos[(long)trow * C + c + k];
            if (bvec) v += bvec[(long)b * C + c + k];
            o[k] = v;
        }
    }
}

template <int V>
__global__ void __launch_bounds__(256) pool_tokens_bwd_kernel(const float* __restrict__ dtok, int B, int H, int W, int C, int oh, int ow,
                                                              int T_total, int tok_off, float* __restrict__ dx, const float* __restrict__ add) {
    const int cv = C / V;
    const long total = (long)B * H * W * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * V;
        long t = idx / cv;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
        // output windows containing (h, w): i in [floor(h*oh/H), ceil((h+1)*oh/H) - 1] (several when oh > H)
        const int i_lo = (int)(((long)h * oh) / H), i_hi = (int)((((long)(h + 1) * oh + H - 1) / H) - 1);
        const int j_lo = (int)(((long)w * ow) / W), j_hi = (int)((((long)(w + 1) * ow + W - 1) / W) - 1);
        for (int i = (i_lo > 0 ? i_lo - 1 : 0); i <= i_hi + 1 && i < oh; ++i) {
            const int h0 = ap_start(i, H, oh), h1 = ap_end(i, H, oh);
            if (h < h0 || h >= h1) continue;
            for (int j = (j_lo > 0 ? j_lo - 1 : 0); j <= j_hi + 1 && j < ow; ++j) {
                const int w0 = ap_start(j, W, ow), w1 = ap_end(j, W, ow);
                if (w < w0 || w >= w1) continue;
                const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
                const float* g = dtok + ((long)b * T_total + tok_off + i * ow + j) * C + c;
                if (V == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(g);
                    acc[0] += v.x * inv; acc[1 % V] += v.y * inv; acc[2 % V] += v.z * inv; acc[3 % V] += v.w * inv;
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[k] += g[k] * inv;
                }
            }
        }
        float* o = dx + idx * V;
        if (V == 4) {
            float4 r = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
            if (add) { const float4 a = *reinterpret_cast<const float4*>(add + idx * V); r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
            *reinterpret_cast<float4*>(o) = r;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = add ? add[idx * V + k] + acc[k] : acc[k];
        }
    }
}

// PyTorch upsample_bilinear2d source index (area_pixel_compute_source_index, cubic = false)
__device__ __forceinline__ void bl_src(int o, float scale, int align, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = align ? scale * (float)o : scale * ((float)o + 0.5f) - 0.5f;
    if (!align && s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// one thread per output element; output index order is (b, h, w, c) with c fastest so NHWC
// destinations are written coalesced
__global__ void __launch_bounds__(256) bilinear_fwd_kernel(tf_bilinear_desc d, const float* __restrict__ x, float* __restrict__ y,
                                                           const float* __restrict__ add, float sh, float sw) {
    const long total = (long)d.B * d.Ho * d.Wo * d.C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % d.C);
        long t = idx / d.C;
        const int wo = (int)(t % d.Wo); t /= d.Wo;
        const int ho = (int)(t % d.Ho);
        const int b = (int)(t / d.Ho);
        int h0, h1, w0, w1; float lh0, lh1, lw0, lw1;
        bl_src(ho, sh, d.align_corners, d.Hi, h0, h1, lh0, lh1);
        bl_src(wo, sw, d.align_corners, d.Wi, w0, w1, lw0, lw1);
        const float* p = x + b * d.sb_i + c * d.sc_i;
        const float v = lh0 * (lw0 * p[h0 * d.sh_i + w0 * d.sw_i] + lw1 * p[h0 * d.sh_i + w1 * d.sw_i]) +
                        lh1 * (lw0 * p[h1 * d.sh_i + w0 * d.sw_i] + lw1 * p[h1 * d.sh_i + w1 * d.sw_i]);
        const long oo = b * d.sb_o + c * d.sc_o + ho * d.sh_o + wo * d.sw_o;
        y[oo] = add ? add[oo] + v : v;
    }
}

// NHWC -> NHWC with C % 4 == 0: one thread per 4 channels of an output pixel (4x less index arithmetic, 16-byte accesses)
__global__ void __launch_bounds__(256) bilinear_fwd_v4_kernel(tf_bilinear_desc d, const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ add, float sh, float sw) {
    const int cv = d.C >> 2;
    const long total = (long)d.B * d.Ho * d.Wo * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * 4;
        long t = idx / cv;
        const int wo = (int)(t % d.Wo); t /= d.Wo;
        const int ho = (int)(t % d.Ho);
        const int b = (int)(t / d.Ho);
        int h0, h1, w0, w1; float lh0, lh1, lw0, lw1;
        bl_src(ho, sh, d.align_corners, d.Hi, h0, h1, lh0, lh1);
        bl_src(wo, sw, d.align_corners, d.Wi, w0, w1, lw0, lw1);
        const float* p = x + b * d.sb_i + c;
        const float4 a = *reinterpret_cast<const float4*>(p + h0 * d.sh_i + w0 * d.sw_i), bq = *reinterpret_cast<const float4*>(p + h0 * d.sh_i + w1 * d.sw_i);
        const float4 cq = *reinterpret_cast<const float4*>(p + h1 * d.sh_i + w0 * d.sw_i), dq = *reinterpret_cast<const float4*>(p + h1 * d.sh_i + w1 * d.sw_i);
        float4 v;
        v.x = lh0 * (lw0 * a.x + lw1 * bq.x) + lh1 * (lw0 * cq.x + lw1 * dq.x);
        v.y = lh0 * (lw0 * a.y + lw1 * bq.y) + lh1 * (lw0 * cq.y + lw1 * dq.y);
        v.z = lh0 * (lw0 * a.z + lw1 * bq.z) + lh1 * (lw0 * cq.z + lw1 * dq.z);
        v.w = lh0 * (lw0 * a.w + lw1 * bq.w) + lh1 * (lw0 * cq.w + lw1 * dq.w);
        const long oo = b * d.sb_o + c + ho * d.sh_o + wo * d.sw_o;
        if (add) { const float4 r = *reinterpret_cast<const float4*>(add + oo); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        *reinterpret_cast<float4*>(y + oo) = v;
    }
}

// NHWC dy -> NHWC dx with C % 4 == 0: one thread per 4 channels of an INPUT pixel; the column weights of the candidate output columns
// are computed once (not once per candidate row) and kept in registers.
constexpr int kBlMaxCand = 24;
__global__ void __launch_bounds__(256) bilinear_bwd_v4_kernel(tf_bilinear_desc d, const float* __restrict__ dy, float* __restrict__ dx, float sh,
                                                              float sw, int accumulate) {
    const int cv = d.C >> 2;
    const long total = (long)d.B * d.Hi * d.Wi * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * 4;
        long t = idx / cv;
        const int wi = (int)(t % d.Wi); t /= d.Wi;
        const int hi = (int)(t % d.Hi);
        const int b = (int)(t / d.Hi);
        int ho_lo, ho_hi, wo_lo, wo_hi;
        {
            const float inv = 1.0f / sh;
            const float lo = d.align_corners ? ((float)hi - 1.f) * inv : ((float)hi - 1.f + 0.5f) * inv - 0.5f;
            const float hi_ = d.align_corners ? ((float)hi + 1.f) * inv : ((float)hi + 1.f + 0.5f) * inv - 0.5f;
            ho_lo = (int)floorf(lo) - 1; ho_hi = (int)ceilf(hi_) + 1;
            if (ho_lo < 0) ho_lo = 0;
            if (ho_hi > d.Ho - 1) ho_hi = d.Ho - 1;
        }
        {
            const float inv = 1.0f / sw;
            const float lo = d.align_corners ? ((float)wi - 1.f) * inv : ((float)wi - 1.f + 0.5f) * inv - 0.5f;
            const float hi_ = d.align_corners ? ((float)wi + 1.f) * inv : ((float)wi + 1.f + 0.5f) * inv - 0.5f;
            wo_lo = (int)floorf(lo) - 1; wo_hi = (int)ceilf(hi_) + 1;
            if (wo_lo < 0) wo_lo = 0;
            if (wo_hi > d.Wo - 1) wo_hi = d.Wo - 1;
        }
        float ww[kBlMaxCand];
#pragma unroll
        for (int j = 0; j < kBlMaxCand; ++j) {
            const int wo = wo_lo + j;
            float w = 0.f;
            if (wo <= wo_hi) {
                int w0, w1; float m0, m1;
                bl_src(wo, sw, d.align_corners, d.Wi, w0, w1, m0, m1);
                if (w0 == wi) w += m0;
                if (w1 == wi) w += m1;
            }
            ww[j] = w;
        }
        const float* g = dy + b * d.sb_o + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            int h0, h1; float l0, l1;
            bl_src(ho, sh, d.align_corners, d.Hi, h0, h1, l0, l1);
            float wh = 0.f;
            if (h0 == hi) wh += l0;
            if (h1 == hi) wh += l1;
            if (wh == 0.f) continue;
            // UNCONDITIONAL loads from clamped addresses, 8 in flight, the weight test applied to the VALUE: a load under `if (w != 0)` makes
            // hipcc wait for every single one (the 24-candidate walk was a chain of dependent ~1 us loads)
            const float* grow = g + ho * d.sh_o;
#pragma unroll
            for (int j0 = 0; j0 < kBlMaxCand; j0 += 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int wo = (wo_lo + j0 + j < wo_hi) ? wo_lo + j0 + j : wo_hi;
                    v[j] = *reinterpret_cast<const float4*>(grow + wo * d.sw_o);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = wh * ww[j0 + j];
                    if (w != 0.f) { acc.x += w * v[j].x; acc.y += w * v[j].y; acc.z += w * v[j].z; acc.w += w * v[j].w; }
                }
            }
        }
        const long io = b * d.sb_i + c + hi * d.sh_i + wi * d.sw_i;
        float4* o = reinterpret_cast<float4*>(dx + io);
        if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
        *o = acc;
    }
}

// gather form of the backward: one thread per INPUT element, loops over the output pixels that
// reference it (no atomics, deterministic).  Index order follows the input's fastest stride.
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(tf_bilinear_desc d, const float* __restrict__ dy, float* __restrict__ dx, float sh,
                                                           float sw, int accumulate, int c_fastest) {
    const long total = (long)d.B * d.Hi * d.Wi * d.C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        int b, c, hi, wi;
        if (c_fastest) { c = (int)(idx % d.C); long t = idx / d.C; wi = (int)(t % d.Wi); t /= d.Wi; hi = (int)(t % d.Hi); b = (int)(t / d.Hi); }
        else { wi = (int)(idx % d.Wi); long t = idx / d.Wi; hi = (int)(t % d.Hi); t /= d.Hi; c = (int)(t % d.C); b = (int)(t / d.C); }
        // candidate output range: |src(o) - i| < 1  (+-1 guard band)
        int ho_lo, ho_hi, wo_lo, wo_hi;
        {
            const float inv = 1.0f / sh;
            float lo = d.align_corners ? ((float)hi - 1.f) * inv : ((float)hi - 1.f + 0.5f) * inv - 0.5f;
            float hi_ = d.align_corners ? ((float)hi + 1.f) * inv : ((float)hi + 1.f + 0.5f) * inv - 0.5f;
            ho_lo = (int)floorf(lo) - 1; ho_hi = (int)ceilf(hi_) + 1;
            if (ho_lo < 0) ho_lo = 0;
            if (ho_hi > d.Ho - 1) ho_hi = d.Ho - 1;
        }
        {
            const float inv = 1.0f / sw;
            float lo = d.align_corners ? ((float)wi - 1.f) * inv : ((float)wi - 1.f + 0.5f) * inv - 0.5f;
            float hi_ = d.align_corners ? ((float)wi + 1.f) * inv : ((float)wi + 1.f + 0.5f) * inv - 0.5f;
            wo_lo = (int)floorf(lo) - 1; wo_hi = (int)ceilf(hi_) + 1;
            if (wo_lo < 0) wo_lo = 0;
            if (wo_hi > d.Wo - 1) wo_hi = d.Wo - 1;
        }
        const float* g = dy + b * d.sb_o + c * d.sc_o;
        float acc = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            int h0, h1; float l0, l1;
            bl_src(ho, sh, d.align_corners, d.Hi, h0, h1, l0, l1);
            float wh = 0.f;
            if (h0 == hi) wh += l0;
            if (h1 == hi) wh += l1;
            if (wh == 0.f) continue;
            float rowacc = 0.f;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                int w0, w1; float m0, m1;
                bl_src(wo, sw, d.align_corners, d.Wi, w0, w1, m0, m1);
                float ww = 0.f;
                if (w0 == wi) ww += m0;
                if (w1 == wi) ww += m1;
                if (ww != 0.f) rowacc += ww * g[ho * d.sh_o + wo * d.sw_o];
            }
            acc += wh * rowacc;
        }
        const long io = b * d.sb_i + c * d.sc_i + hi * d.sh_i + wi * d.sw_i;
        dx[io] = accumulate ? dx[io] + acc : acc;
    }
}

// The same gather for SMALL inputs (the GPT stages' 8 x 22 / 8 x 8 token maps up-sampled 8x / 4x: 12 672 input elements that each walk ~20 x 20
// candidate output pixels - 50 workgroups of 400-step threads took 193 us): R = 8 adjacent lanes share one input element, lane r takes the
// candidate rows ho_lo + r, + R, ..., the column weights are computed once per thread, and the R partial sums are combined by a fixed shuffle
// tree (deterministic).  Channel-fastest element order as in bilinear_bwd_kernel: a wave reads 8 consecutive channels of 8 rows per load.
template <int R>
__global__ void __launch_bounds__(256) bilinear_bwd_split_kernel(tf_bilinear_desc d, const float* __restrict__ dy, float* __restrict__ dx, float sh,
                                                                 float sw, int accumulate) {
    const long total = (long)d.B * d.Hi * d.Wi * d.C;
    const long nthreads = (total * R + 255) / 256 * 256;               // whole blocks: every lane takes part in the shuffles
    for (long tidx = (long)blockIdx.x * 256 + threadIdx.x; tidx < nthreads; tidx += (long)gridDim.x * 256) {
        const long idx = tidx / R;
        const int r = (int)(tidx % R);
        const bool live = idx < total;
        const long e = live ? idx : 0;
        const int c = (int)(e % d.C);
        long t = e / d.C;
        const int wi = (int)(t % d.Wi); t /= d.Wi;
        const int hi = (int)(t % d.Hi);
        const int b = (int)(t / d.Hi);
        int ho_lo, ho_hi, wo_lo, wo_hi;
        {
            const float inv = 1.0f / sh;
            const float lo = d.align_corners ? ((float)hi - 1.f) * inv : ((float)hi - 1.f + 0.5f) * inv - 0.5f;
            const float hi_ = d.align_corners ? ((float)hi + 1.f) * inv : ((float)hi + 1.f + 0.5f) * inv - 0.5f;
            ho_lo = (int)floorf(lo) - 1; ho_hi = (int)ceilf(hi_) + 1;
            if (ho_lo < 0) ho_lo = 0;
            if (ho_hi > d.Ho - 1) ho_hi = d.Ho - 1;
        }
        {
            const float inv = 1.0f / sw;
            const float lo = d.align_corners ? ((float)wi - 1.f) * inv : ((float)wi - 1.f + 0.5f) * inv - 0.5f;
            const float hi_ = d.align_corners ? ((float)wi + 1.f) * inv : ((float)wi + 1.f + 0.5f) * inv - 0.5f;
            wo_lo = (int)floorf(lo) - 1; wo_hi = (int)ceilf(hi_) + 1;
            if (wo_lo < 0) wo_lo = 0;
            if (wo_hi > d.Wo - 1) wo_hi = d.Wo - 1;
        }
        // column weights of the candidate window, 8 at a time and only as many groups as the window has (the full 24-entry table cost ~500 VALU
        // operations per thread: the 2x / 4x maps, whose windows are 8 / 12 wide, were bound by it)
        const int ncand = wo_hi - wo_lo + 1;
        float ww[kBlMaxCand];
#pragma unroll
        for (int j0 = 0; j0 < kBlMaxCand; j0 += 8) {
            if (j0 < ncand) {
#pragma unroll
                for (int j = j0; j < j0 + 8; ++j) {
                    const int wo = wo_lo + j;
                    float w = 0.f;
                    if (wo <= wo_hi) {
                        int w0, w1; float m0, m1;
                        bl_src(wo, sw, d.align_corners, d.Wi, w0, w1, m0, m1);
                        if (w0 == wi) w += m0;
                        if (w1 == wi) w += m1;
                    }
                    ww[j] = w;
                }
            } else {
#pragma unroll
                for (int j = j0; j < j0 + 8; ++j) ww[j] = 0.f;
            }
        }
        const float* g = dy + b * d.sb_o + c * d.sc_o;
        float acc = 0.f;
        if (live)
            for (int ho = ho_lo + r; ho <= ho_hi; ho += R) {
                int h0, h1; float l0, l1;
                bl_src(ho, sh, d.align_corners, d.Hi, h0, h1, l0, l1);
                float wh = 0.f;
                if (h0 == hi) wh += l0;
                if (h1 == hi) wh += l1;
                if (wh == 0.f) continue;
                const float* grow = g + ho * d.sh_o;
                float rowacc = 0.f;
#pragma unroll
                for (int j0 = 0; j0 < kBlMaxCand; j0 += 8) {      // unconditional clamped loads, 8 in flight (see bilinear_bwd_v4_kernel)
                    if (j0 >= ncand) break;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int wo = (wo_lo + j0 + j < wo_hi) ? wo_lo + j0 + j : wo_hi;
                        v[j] = grow[wo * d.sw_o];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (ww[j0 + j] != 0.f) rowacc += ww[j0 + j] * v[j];
                }
                acc += wh * rowacc;
            }
#pragma unroll
        for (int m = R / 2; m >= 1; m >>= 1) acc += shfl_xor(acc, m);
        if (live && r == 0) {
            const long io = b * d.sb_i + c * d.sc_i + hi * d.sh_i + wi * d.sw_i;
            dx[io] = accumulate ? dx[io] + acc : acc;
        }
    }
}

inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
inline float bl_scale(int in, int out, int align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

// ---- G1: geometric-fusion correspondence gather (geometric_fusion.py:134-137,147-150).  The reference gathers B x B and keeps the
// diagonal; here each sample gathers only its own K correspondences.  idx holds (x, y) int64 pairs, src is (B, Hs*Ws, E).
constexpr int kGatherMaxCorr = 2048;   // n*K correspondences per sample (reference: 8*8*5 = 320 and 5*22*5 = 550)

__global__ void __launch_bounds__(256) gather_sum_fwd_kernel(const float* __restrict__ src, const long long* __restrict__ idx, int B, int S, int Ws,
                                                             int E, int n, int K, float* __restrict__ out) {
    const int ev = E >> 2;
    const long total = (long)B * n * ev;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int e = (int)(t % ev) * 4;
        const long cell = t / ev;                  // b * n + i
        const int b = (int)(cell / n);
        const long long* q = idx + cell * K * 2;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < K; ++k) {
            long long j = q[2 * k + 1] * Ws + q[2 * k];
            j = j < 0 ? 0 : (j >= S ? S - 1 : j);   // indices are validated on the host; clamp keeps a bad value from faulting
            const float4 v = *reinterpret_cast<const float4*>(src + ((long)b * S + j) * E + e);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + cell * E + e) = acc;
    }
}

// backward = the transposed gather, evaluated per SOURCE cell in a fixed order (deterministic, no atomics): every block owns one
// (sample, source cell), scans the n*K correspondences of that sample staged in LDS and adds the matching rows of dout.
__global__ void __launch_bounds__(128) gather_sum_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ idx, int B, int S, int Ws,
                                                             int E, int n, int K, float* __restrict__ dsrc, int accumulate) {
    __shared__ int hits[kGatherMaxCorr];            // destination cells i that read this source cell (with multiplicity)
    __shared__ int nhit;
    const int b = blockIdx.x / S, j = blockIdx.x % S;
    if (threadIdx.x == 0) nhit = 0;
    __syncthreads();
    const long long* q = idx + (long)b * n * K * 2;
    if (threadIdx.x == 0) {                         // n*K <= a few hundred: a serial ordered scan keeps the summation order fixed
        int c = 0;
        for (int t = 0; t < n * K; ++t) {
            long long jj = q[2 * t + 1] * Ws + q[2 * t];
            jj = jj < 0 ? 0 : (jj >= S ? S - 1 : jj);
            if ((int)jj == j) hits[c++] = t / K;
        }
        nhit = c;
    }
    __syncthreads();
    const int c = nhit;
    for (int e = threadIdx.x * 4; e < E; e += 128 * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int h = 0; h < c; ++h) {
            const float4 v = *reinterpret_cast<const float4*>(dout + ((long)b * n + hits[h]) * E + e);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float4* o = reinterpret_cast<float4*>(dsrc + ((long)b * S + j) * E + e);
        if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
        *o = acc;
    }
}

}  // namespace

extern "C" int tf_pool_tokens_fwd_f32(const float* x, int B, int H, int W, int C, int oh, int ow, const float* pos, const float* bvec, float* tok,
                                      int T_total, int tok_off, void* stream) {
    TF_REQUIRE(x && tok && B > 0 && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0 && tok_off + oh * ow <= T_total, "tf_pool_tokens_fwd_f32: bad arguments");
    const bool v4 = C % 4 == 0 && aligned16(x);
    const long n = (long)B * oh * ow * (C / (v4 ? 4 : 1));
    if (v4) TF_LAUNCH(pool_tokens_fwd_kernel<4>, dim3(ew_blocks(n)), dim3(256), stream, x, B, H, W, C, oh, ow, pos, bvec, tok, T_total, tok_off);
    else TF_LAUNCH(pool_tokens_fwd_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, x, B, H, W, C, oh, ow, pos, bvec, tok, T_total, tok_off);
    return launch_status("tf_pool_tokens_fwd_f32");
}

extern "C" int tf_pool_tokens_bwd_f32(const float* dtok, int B, int H, int W, int C, int oh, int ow, int T_total, int tok_off, float* dx,
                                      const float* add, void* stream) {
    TF_REQUIRE(dtok && dx && B > 0 && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0 && tok_off + oh * ow <= T_total, "tf_pool_tokens_bwd_f32: bad arguments");
    const long n = (long)B * H * W * C;
    // one thread per 4 channels where the layout allows (the scalar form spent ~100 integer operations of window arithmetic per ELEMENT: 104 us for
    // the 64 x 176 x 72 map against a 20 us copy)
    if (C % 4 == 0 && aligned16(dtok) && aligned16(dx) && (!add || aligned16(add)))
        TF_LAUNCH(pool_tokens_bwd_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, dtok, B, H, W, C, oh, ow, T_total, tok_off, dx, add);
    else
        TF_LAUNCH(pool_tokens_bwd_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, dtok, B, H, W, C, oh, ow, T_total, tok_off, dx, add);
    return launch_status("tf_pool_tokens_bwd_f32");
}

extern "C" int tf_bilinear_fwd_f32(const tf_bilinear_desc* d, const float* x, float* y, const float* add, void* stream) {
    TF_REQUIRE(d && x && y && d->B > 0 && d->C > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0, "tf_bilinear_fwd_f32: bad arguments");
    const long n = (long)d->B * d->Ho * d->Wo * d->C;
    const bool nhwc4 = d->sc_i == 1 && d->sc_o == 1 && d->C % 4 == 0 && aligned16(x) && aligned16(y) && (!add || aligned16(add)) && d->sw_i % 4 == 0 &&
                       d->sh_i % 4 == 0 && d->sb_i % 4 == 0 && d->sw_o % 4 == 0 && d->sh_o % 4 == 0 && d->sb_o % 4 == 0;
    if (nhwc4) {
        TF_LAUNCH(bilinear_fwd_v4_kernel, dim3(ew_blocks(n / 4)), dim3(256), stream, *d, x, y, add, bl_scale(d->Hi, d->Ho, d->align_corners),
                  bl_scale(d->Wi, d->Wo, d->align_corners));
        return launch_status("tf_bilinear_fwd_f32");
    }
    TF_LAUNCH(bilinear_fwd_kernel, dim3(ew_blocks(n)), dim3(256), stream, *d, x, y, add, bl_scale(d->Hi, d->Ho, d->align_corners),
              bl_scale(d->Wi, d->Wo, d->align_corners));
    return launch_status("tf_bilinear_fwd_f32");
}

extern "C" int tf_bilinear_bwd_f32(const tf_bilinear_desc* d, const float* dy, float* dx, int accumulate, void* stream) {
    TF_REQUIRE(d && dy && dx && d->B > 0 && d->C > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0, "tf_bilinear_bwd_f32: bad arguments");
    const long n = (long)d->B * d->Hi * d->Wi * d->C;
    const float bsh = bl_scale(d->Hi, d->Ho, d->align_corners), bsw = bl_scale(d->Wi, d->Wo, d->align_corners);
    // candidate output columns per input column: 2 / scale + 4 (guard bands) must fit the register table of the vector kernel
    const bool fits = bsw > 0.f && (2.0f / bsw + 5.0f) <= (float)kBlMaxCand;
    const bool nhwc4 = fits && d->sc_i == 1 && d->sc_o == 1 && d->C % 4 == 0 && aligned16(dy) && aligned16(dx) && d->sw_i % 4 == 0 && d->sh_i % 4 == 0 &&
                       d->sb_i % 4 == 0 && d->sw_o % 4 == 0 && d->sh_o % 4 == 0 && d->sb_o % 4 == 0;
    if (nhwc4) {
        TF_LAUNCH(bilinear_bwd_v4_kernel, dim3(ew_blocks(n / 4)), dim3(256), stream, *d, dy, dx, bsh, bsw, accumulate);
        return launch_status("tf_bilinear_bwd_f32");
    }
    const bool c_fastest = d->sc_i == 1 || d->sc_o == 1;
    if (fits && c_fastest && n <= (1L << 20) && bsh > 0.f && bsh <= 0.5f) {          // few input elements, each gathering >= 4 x 4 outputs: split the rows
        // lanes per input element = candidate rows / ~2.5: 8 for the 8x maps (20 rows), 4 for 4x (12 rows), 2 for 2x (8 rows)
        if (bsh <= 0.1875f) TF_LAUNCH(bilinear_bwd_split_kernel<8>, dim3(ew_blocks(n * 8)), dim3(256), stream, *d, dy, dx, bsh, bsw, accumulate);
        else if (bsh <= 0.375f) TF_LAUNCH(bilinear_bwd_split_kernel<4>, dim3(ew_blocks(n * 4)), dim3(256), stream, *d, dy, dx, bsh, bsw, accumulate);
        else TF_LAUNCH(bilinear_bwd_split_kernel<2>, dim3(ew_blocks(n * 2)), dim3(256), stream, *d, dy, dx, bsh, bsw, accumulate);
        return launch_status("tf_bilinear_bwd_f32");
    }
    TF_LAUNCH(bilinear_bwd_kernel, dim3(ew_blocks(n)), dim3(256), stream, *d, dy, dx, bl_scale(d->Hi, d->Ho, d->align_corners),
              bl_scale(d->Wi, d->Wo, d->align_corners), accumulate, c_fastest ? 1 : 0);   // channel-fastest threads whenever dY is NHWC:
              // an input element gathers ~(scale + 2)^2 dY values but is written once, so the READS must coalesce (the GPT stages' raw-view
              // layout, sc_i = ih * iw, ran pixel-fastest: 4-byte reads at stride C, 256 us for the 64 x 176 x 72 map)
    return launch_status("tf_bilinear_bwd_f32");
}

// ---- 3x3 / stride 2 / pad 1 max pooling on NHWC (timm ResNet stem, used by the reference's default resnet34 / resnet18 trunks under their
// own name ``maxpool``: transfuser.py:139,143).  Forward keeps the winning tap (0..8, first maximum in (kh, kw) scan order like ATen's
// max_pool2d) as one byte per output element; the backward is a gather over the <= 4 windows that contain an input pixel: deterministic.
namespace {
__global__ void __launch_bounds__(256) maxpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int B, int Hi, int Wi,
                                                           int C, int Ho, int Wo) {
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long t = i / C;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const long b = t / Ho;
        float best = -3.402823466e38f;
        int arg = 0;
        bool any = false;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int h = 2 * ho - 1 + kh, w = 2 * wo - 1 + kw;
                if ((unsigned)h < (unsigned)Hi && (unsigned)w < (unsigned)Wi) {
                    const float v = x[((b * Hi + h) * Wi + w) * C + c];
                    if (!any || v > best) { best = v; arg = kh * 3 + kw; any = true; }
                }
            }
        y[i] = best;
        idx[i] = (unsigned char)arg;
    }
}
__global__ void __launch_bounds__(256) maxpool3_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx, int B, int Hi,
                                                           int Wi, int C, int Ho, int Wo) {
    const long total = (long)B * Hi * Wi * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long t = i / C;
        const int w = (int)(t % Wi); t /= Wi;
        const int h = (int)(t % Hi);
        const long b = t / Hi;
        float acc = 0.f;
        for (int ho = (h + 1) / 2 - 1; ho <= (h + 1) / 2; ++ho) {
            const int kh = h - (2 * ho - 1);
            if (ho < 0 || ho >= Ho || kh < 0 || kh > 2) continue;
            for (int wo = (w + 1) / 2 - 1; wo <= (w + 1) / 2; ++wo) {
                const int kw = w - (2 * wo - 1);
                if (wo < 0 || wo >= Wo || kw < 0 || kw > 2) continue;
                const long o = ((b * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == kh * 3 + kw) acc += dy[o];
            }
        }
        dx[i] = acc;
    }
}
}  // namespace

extern "C" int tf_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int B, int Hi, int Wi, int C, void* stream) {
    TF_REQUIRE(x && y && idx && B > 0 && Hi > 0 && Wi > 0 && C > 0, "tf_maxpool3x3s2_fwd_f32: bad arguments");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long n = (long)B * Ho * Wo * C;
    TF_LAUNCH(maxpool3_fwd_kernel, dim3(ew_blocks(n)), dim3(256), stream, x, y, idx, B, Hi, Wi, C, Ho, Wo);
    return launch_status("tf_maxpool3x3s2_fwd_f32");
}
extern "C" int tf_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int B, int Hi, int Wi, int C, void* stream) {
    TF_REQUIRE(dy && dx && idx && B > 0 && Hi > 0 && Wi > 0 && C > 0, "tf_maxpool3x3s2_bwd_f32: bad arguments");
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long n = (long)B * Hi * Wi * C;
    TF_LAUNCH(maxpool3_bwd_kernel, dim3(ew_blocks(n)), dim3(256), stream, dy, idx, dx, B, Hi, Wi, C, Ho, Wo);
    return launch_status("tf_maxpool3x3s2_bwd_f32");
}

extern "C" int tf_gather_sum_fwd_f32(const float* src, const long long* idx, int B, int Hs, int Ws, int E, int n, int K, float* out, void* stream) {
    TF_REQUIRE(src && idx && out && B > 0 && Hs > 0 && Ws > 0 && E > 0 && E % 4 == 0 && n > 0 && K > 0 && aligned16(src) && aligned16(out),
               "tf_gather_sum_fwd_f32: bad arguments (E must be a multiple of 4, pointers 16-byte aligned)");
    const long total = (long)B * n * (E / 4);
    TF_LAUNCH(gather_sum_fwd_kernel, dim3(ew_blocks(total)), dim3(256), stream, src, idx, B, Hs * Ws, Ws, E, n, K, out);
    return launch_status("tf_gather_sum_fwd_f32");
}

extern "C" int tf_gather_sum_bwd_f32(const float* dout, const long long* idx, int B, int Hs, int Ws, int E, int n, int K, float* dsrc, int accumulate,
                                     void* stream) {
    TF_REQUIRE(dout && idx && dsrc && B > 0 && Hs > 0 && Ws > 0 && E > 0 && E % 4 == 0 && n > 0 && K > 0 && aligned16(dout) && aligned16(dsrc) &&
               (long)n * K <= kGatherMaxCorr, "tf_gather_sum_bwd_f32: bad arguments (E % 4 == 0, n*K <= 2048)");
    TF_LAUNCH(gather_sum_bwd_kernel, dim3(B * Hs * Ws), dim3(128), stream, dout, idx, B, Hs * Ws, Ws, E, n, K, dsrc, accumulate);
    return launch_status("tf_gather_sum_bwd_f32");
}

// Loss kernels.  All reductions are deterministic (per-block
// partials summed in a fixed order); gradients are produced un-scaled in the forward pass and
// multiplied by the upstream scalar(s) in tf_scale_dev_f32, or computed in a dedicated backward.
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int CE_MAXC = 16;

// F.cross_entropy(logits, target, weight) over NHWC logits (rows, C): model.py:763,783.
// dl[row][c] = w_y * (softmax_c - [c == y]);  partial[block] = (sum w_y * nll, sum w_y)
__global__ void __launch_bounds__(256) ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ cw, long rows, int C, float* __restrict__ dl,
                                                     float* __restrict__ partial) {
    __shared__ float red[4];
    float ls = 0.f, ws = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        float v[CE_MAXC];
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c)
            if (c < C) { v[c] = logits[r * C + c]; mx = fmaxf(mx, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
        const int y = (int)target[r];
        const float w = cw ? cw[y] : 1.f;
        const float inv = 1.f / s;
        float py = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c)
            if (c < C) {
                const float p = v[c] * inv;
                if (c == y) py = p;
                dl[r * C + c] = w * (p - (c == y ? 1.f : 0.f));
            }
        ls += -w * logf(py);
        ws += w;
    }
    ls = block_sum<4>(ls, red);
    ws = block_sum<4>(ws, red);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = ls; partial[blockIdx.x * 2 + 1] = ws; }
}
// one wave: the lanes stride over the block partials, then a shuffle tree (fixed order; a single thread walking ~1800 partials was a 61 us
// chain of dependent loads on the critical path between the forward and the backward pass)
__global__ void ce_finalize_kernel(const float* __restrict__ partial, int nb, float* __restrict__ loss, float* __restrict__ inv_wsum) {
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) { a += partial[i * 2]; b += partial[i * 2 + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        *loss = a / b;
        *inv_wsum = 1.f / b;
    }
}

// mean |f(pred) - target|, f = identity or sigmoid (model.py:765 loss_wp, :784 depth through
// transfuser.py:279 sigmoid).  dpred = sign * f' / n  (un-scaled by the upstream gradient).
__global__ void __launch_bounds__(256) l1_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, long n, int use_sigmoid,
                                                     float* __restrict__ dpred, float* __restrict__ partial) {
    __shared__ float red[4];
    float ls = 0.f;
    const float invn = 1.f / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float p = pred[i], d1 = 1.f;
        if (use_sigmoid) { p = 1.f / (1.f + expf(-p)); d1 = p * (1.f - p); }
        const float d = p - target[i];
        ls += fabsf(d);
        dpred[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * d1 * invn;
    }
    ls = block_sum<4>(ls, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = ls;
}
__global__ void l1_finalize_kernel(const float* __restrict__ partial, int nb, float invn, float* __restrict__ loss) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) a += partial[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) *loss = a * invn;
}

// x *= (*a) * (*b) * mult   (a, b optional device scalars)
__global__ void __launch_bounds__(256) scale_dev_kernel(float* __restrict__ x, long n, const float* a, const float* b, float mult) {
    const float s = (a ? *a : 1.f) * (b ? *b : 1.f) * mult;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= s;
}

// ---------------------------------------------------------------- CenterNet targets (model.py:285-374)
// mmdet gaussian_radius (fp32 op order of the tensor expression, see oracle/centernet.py)
__device__ __forceinline__ float gaussian_radius_f32(float h, float w) {
    // min_overlap = 0.1 (model.py:343); python-double constants are rounded to f32 when they meet a tensor
    const float k09 = (float)(1.0 - 0.1), k11 = (float)(1.0 + 0.1);
    const float b1 = h + w;
    const float c1 = ((w * h) * k09) / k11;
    const float r1 = (b1 - sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
    const float b2 = 2.f * (h + w);
    const float c2 = (k09 * w) * h;
    const float r2 = (b2 - sqrtf(b2 * b2 - 16.f * c2)) / 8.f;
    const float b3 = (float)(-2 * 0.1) * (h + w);
    const float c3 = ((float)(0.1 - 1) * w) * h;
    const float r3 = (b3 + sqrtf(b3 * b3 - (float)(4 * (4 * 0.1)) * c3)) / (float)(2 * (4 * 0.1));
    return fminf(r1, fminf(r2, r3));
}
__device__ __forceinline__ float remainder_f32(float a, float b) {  // torch.remainder (python modulo)
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}

// tgtf (B,fh,fw,8) = [hm, wh_w, wh_h, off_x, off_y, yaw_res, vel, weight]; tgti (B,fh,fw,2) = [yaw_cls, brake]
__global__ void __launch_bounds__(256) centernet_targets_kernel(const float* __restrict__ label, int nbox, int fh, int fw, float wr, float hr,
                                                                int nbins, float* __restrict__ tgtf, int32_t* __restrict__ tgti,
                                                                int32_t* __restrict__ cnt) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cells = fh * fw;
    float* tf_ = tgtf + (long)b * cells * 8;
    int32_t* ti = tgti + (long)b * cells * 2;
    for (int i = tid; i < cells * 8; i += 256) tf_[i] = 0.f;
    for (int i = tid; i < cells * 2; i += 256) ti[i] = 0;
    __syncthreads();
    const float two_pi = (float)(2.0 * 3.141592653589793);
    const float apc = (float)(2.0 * 3.141592653589793 / (double)nbins);
    const float apc_half = (float)(2.0 * 3.141592653589793 / (double)nbins / 2.0);
    for (int j = 0; j < nbox; ++j) {
        const float* L = label + ((long)b * nbox + j) * 7;
        const float s = ((((((L[0] + L[1]) + L[2]) + L[3]) + L[4]) + L[5]) + L[6]);
        if (s == 0.f) continue;  // gt_bboxes_ignore (model.py:774); uniform over the block
        const float cx = L[0] * wr, cy = L[1] * wr;  // quirk Q2: y uses the width ratio
        const int cxi = (int)cx, cyi = (int)cy;
        const float sh = L[3] * hr, sw = L[2] * wr;
        int radius = (int)gaussian_radius_f32(sh, sw);
        if (radius < 2) radius = 2;
        const double sigma = (double)(2 * radius + 1) / 6.0;
        const float denom = (float)(2.0 * sigma * sigma);
        int left = cxi < radius ? cxi : radius, right = (fw - cxi) < (radius + 1) ? (fw - cxi) : (radius + 1);
        int top = cyi < radius ? cyi : radius, bottom = (fh - cyi) < (radius + 1) ? (fh - cyi) : (radius + 1);
        const int pw = left + right, ph = top + bottom;
        if (pw > 0 && ph > 0 && cxi >= 0 && cyi >= 0) {
            for (int i = tid; i < pw * ph; i += 256) {
                const int dy = i / pw - top, dx = i % pw - left;
                float g = expf(-(float)(dx * dx + dy * dy) / denom);
                if (g < 1.1920929e-07f) g = 0.f;
                float* hp = tf_ + ((long)(cyi + dy) * fw + (cxi + dx)) * 8;
                hp[0] = fmaxf(hp[0], g);
            }
        }
        if (tid == 0 && cxi >= 0 && cxi < fw && cyi >= 0 && cyi < fh) {
            float* p = tf_ + ((long)cyi * fw + cxi) * 8;
            int32_t* q = ti + ((long)cyi * fw + cxi) * 2;
            p[1] = sw; p[2] = sh;
            p[3] = cx - (float)cxi; p[4] = cy - (float)cyi;
            const float ang = remainder_f32(L[4], two_pi);
            const float shifted = remainder_f32(ang + apc_half, two_pi);
            const float cls = truncf(shifted / apc);
            q[0] = (int32_t)cls;
            p[5] = shifted - (cls * apc + apc_half);
            p[6] = L[5];
            q[1] = (int32_t)L[6];
            p[7] = 1.f;
        }
        __syncthreads();
    }
    float c1 = 0.f;
    for (int i = tid; i < cells; i += 256) c1 += (tf_[(long)i * 8] == 1.f) ? 1.f : 0.f;
    c1 = block_sum<4>(c1, red);
    if (tid == 0) cnt[b] = (int32_t)c1;
}

// ---------------------------------------------------------------- CenterNet losses (model.py:150-248)
// pred (B,fh,fw,9+nbins) = [hm logit, wh(2), off(2), yaw_cls(nbins), yaw_res, vel, brake(2)]
// losses[7] = center_heatmap, wh, offset, yaw_class, yaw_res, velocity, brake  (mmdet weighted_loss,
// avg_factor = max(1, #hm==1); wh loss_weight 0.1; yaw_class / brake CE reproduce the (B,B,H,W)
// weight broadcast of mmdet's `loss * weight` - quirk Q3).
constexpr int CN_MAXBINS = 16;
__device__ __forceinline__ float avg_factor_of(const int32_t* cnt, int B) {
    int s = 0;
    for (int i = 0; i < B; ++i) s += cnt[i];
    return (float)(s > 1 ? s : 1);
}

template <bool BWD>
__global__ void __launch_bounds__(256) centernet_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgtf,
                                                             const int32_t* __restrict__ tgti, const int32_t* __restrict__ cnt, int B, int fh, int fw,
                                                             int nbins, float* __restrict__ partial, const float* __restrict__ gup,
                                                             float* __restrict__ dpred) {
    __shared__ float red[4];
    const int cells = fh * fw, P = 9 + nbins;
    const float af = avg_factor_of(cnt, B);
    const float eps32 = 1.1920929e-07f;
    const float inv1 = 1.f / (af + eps32), inv2 = 1.f / (af * 2.f + eps32);
    float acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = 0.f;
    float g[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) g[k] = (BWD && gup) ? gup[k] : 1.f;
    // forward: a thread owns a cell and walks the batch (the partial sums stay in registers).  backward: a thread owns ONE (sample, cell) pair -
    // B x more threads, each re-deriving the cell's batch weight sum: the 10-sample walk of 21-channel rows per thread was a 76 us latency chain
    const int nwork = BWD ? cells * B : cells;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < nwork; idx += gridDim.x * 256) {
        const int cell = BWD ? idx % cells : idx;
        const int b_lo = BWD ? idx / cells : 0, b_hi = BWD ? b_lo + 1 : B;
        // pass 1: per-cell sums over the batch for the broadcast quirk
        float wsum = 0.f, ce_y = 0.f, ce_b = 0.f;
        for (int b = 0; b < B; ++b) {
            const long o = (long)b * cells + cell;
            const float* p = pred + o * P;
            wsum += tgtf[o * 8 + 7];
            if (!BWD) {
                float mx = -3.0e38f;
                for (int c = 0; c < nbins; ++c) mx = fmaxf(mx, p[5 + c]);
                float s = 0.f;
                for (int c = 0; c < nbins; ++c) s += expf(p[5 + c] - mx);
                ce_y += -(p[5 + tgti[o * 2]] - mx - logf(s));
                const float b0 = p[7 + nbins], b1 = p[8 + nbins];
                const float m2 = fmaxf(b0, b1);
                const float s2 = expf(b0 - m2) + expf(b1 - m2);
                ce_b += -((tgti[o * 2 + 1] ? b1 : b0) - m2 - logf(s2));
            }
        }
        if (!BWD) { acc[3] += ce_y * wsum; acc[6] += ce_b * wsum; }
        for (int b = b_lo; b < b_hi; ++b) {
            const long o = (long)b * cells + cell;
            const float* p = pred + o * P;
            const float* t = tgtf + o * 8;
            const float w = t[7];
            // gaussian focal loss on sigmoid(hm logit)
            const float pr = 1.f / (1.f + expf(-p[0]));
            const float tg = t[0];
            const float negw = (1.f - tg) * (1.f - tg) * (1.f - tg) * (1.f - tg);
            const float lp = logf(pr + 1e-12f), ln = logf(1.f - pr + 1e-12f);
            if (!BWD) {
                acc[0] += (tg == 1.f ? -lp * (1.f - pr) * (1.f - pr) : 0.f) + (-ln * pr * pr * negw);
                acc[1] += (fabsf(p[1] - t[1]) + fabsf(p[2] - t[2])) * w;
                acc[2] += (fabsf(p[3] - t[3]) + fabsf(p[4] - t[4])) * w;
                const float d = fabsf(p[5 + nbins] - t[5]);
                acc[4] += (d < 1.f ? 0.5f * d * d : d - 0.5f) * w;
                acc[5] += fabsf(p[6 + nbins] - t[6]) * w;
            } else {
                float* dp = dpred + o * P;
                float dpos = 0.f;
                if (tg == 1.f) dpos = -(1.f - pr) * (1.f - pr) / (pr + 1e-12f) + 2.f * (1.f - pr) * lp;
                const float dneg = negw * (pr * pr / (1.f - pr + 1e-12f) - 2.f * pr * ln);
                dp[0] = g[0] * inv1 * (dpos + dneg) * pr * (1.f - pr);
                auto sign = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
                dp[1] = g[1] * 0.1f * inv2 * w * sign(p[1] - t[1]);
                dp[2] = g[1] * 0.1f * inv2 * w * sign(p[2] - t[2]);
                dp[3] = g[2] * inv2 * w * sign(p[3] - t[3]);
                dp[4] = g[2] * inv2 * w * sign(p[4] - t[4]);
                float mx = -3.0e38f;
                for (int c = 0; c < nbins; ++c) mx = fmaxf(mx, p[5 + c]);
                float s = 0.f;
                for (int c = 0; c < nbins; ++c) s += expf(p[5 + c] - mx);
                const int yc = tgti[o * 2];
                for (int c = 0; c < nbins; ++c) dp[5 + c] = g[3] * inv1 * wsum * (expf(p[5 + c] - mx) / s - (c == yc ? 1.f : 0.f));
                const float dd = p[5 + nbins] - t[5];
                dp[5 + nbins] = g[4] * inv1 * w * (fabsf(dd) < 1.f ? dd : sign(dd));
                dp[6 + nbins] = g[5] * inv1 * w * sign(p[6 + nbins] - t[6]);
                const float b0 = p[7 + nbins], b1 = p[8 + nbins];
                const float m2 = fmaxf(b0, b1);
                const float e0 = expf(b0 - m2), e1 = expf(b1 - m2);
                const int yb = tgti[o * 2 + 1];
                dp[7 + nbins] = g[6] * inv1 * wsum * (e0 / (e0 + e1) - (yb == 0 ? 1.f : 0.f));
                dp[8 + nbins] = g[6] * inv1 * wsum * (e1 / (e0 + e1) - (yb == 1 ? 1.f : 0.f));
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float v = block_sum<4>(acc[k], red);
            if (threadIdx.x == 0) partial[blockIdx.x * 7 + k] = v;
        }
    }
}
__global__ void centernet_loss_finalize_kernel(const float* __restrict__ partial, int nb, const int32_t* __restrict__ cnt, int B,
                                               float* __restrict__ losses) {
    if (blockIdx.x == 0 && threadIdx.x < 7) {
        const int k = threadIdx.x;
        float a = 0.f;
        for (int i = 0; i < nb; ++i) a += partial[i * 7 + k];
        const float af = avg_factor_of(cnt, B);
        const float eps32 = 1.1920929e-07f;
        const float lw = (k == 1) ? 0.1f : 1.f;
        const float den = (k == 1 || k == 2) ? (af * 2.f + eps32) : (af + eps32);
        losses[k] = lw * (a / den);
    }
}

inline int nblocks(long n, int cap) {
    long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int tf_ce_fwd_f32(const float* logits, const int64_t* target, const float* class_w, int64_t rows, int C, float* dlogits, float* loss,
                             float* inv_wsum, float* ws, void* stream) {
    TF_REQUIRE(logits && target && dlogits && loss && inv_wsum && ws && rows > 0 && C > 0 && C <= CE_MAXC, "tf_ce_fwd_f32: bad arguments (C=%d)", C);
    const int nb = nblocks(rows, 1024);
    TF_LAUNCH(ce_fwd_kernel, dim3(nb), dim3(256), stream, logits, target, class_w, (long)rows, C, dlogits, ws);
    TF_LAUNCH(ce_finalize_kernel, dim3(1), dim3(64), stream, (const float*)ws, nb, loss, inv_wsum);
    return launch_status("tf_ce_fwd_f32");
}

extern "C" int tf_l1_fwd_f32(const float* pred, const float* target, int64_t n, int use_sigmoid, float* dpred, float* loss, float* ws, void* stream) {
    TF_REQUIRE(pred && target && dpred && loss && ws && n > 0, "tf_l1_fwd_f32: bad arguments");
    const int nb = nblocks(n, 1024);
    TF_LAUNCH(l1_fwd_kernel, dim3(nb), dim3(256), stream, pred, target, (long)n, use_sigmoid, dpred, ws);
    TF_LAUNCH(l1_finalize_kernel, dim3(1), dim3(64), stream, (const float*)ws, nb, 1.f / (float)n, loss);
    return launch_status("tf_l1_fwd_f32");
}

extern "C" int tf_scale_dev_f32(float* x, int64_t n, const float* a_dev, const float* b_dev, float mult, void* stream) {
    TF_REQUIRE(x && n >= 0, "tf_scale_dev_f32: bad arguments");
    if (n == 0) return 0;
    TF_LAUNCH(scale_dev_kernel, dim3(nblocks(n, 4096)), dim3(256), stream, x, (long)n, a_dev, b_dev, mult);
    return launch_status("tf_scale_dev_f32");
}

extern "C" int tf_centernet_targets_f32(const float* label, int B, int nbox, int fh, int fw, float ratio_w, float ratio_h, int num_dir_bins,
                                        float* tgtf, int32_t* tgti, int32_t* cnt, void* stream) {
    TF_REQUIRE(label && tgtf && tgti && cnt && B > 0 && nbox >= 0 && fh > 0 && fw > 0 && num_dir_bins > 0 && num_dir_bins <= CN_MAXBINS,
               "tf_centernet_targets_f32: bad arguments");
    TF_LAUNCH(centernet_targets_kernel, dim3(B), dim3(256), stream, label, nbox, fh, fw, ratio_w, ratio_h, num_dir_bins, tgtf, tgti, cnt);
    return launch_status("tf_centernet_targets_f32");
}

extern "C" int tf_centernet_loss_fwd_f32(const float* pred, const float* tgtf, const int32_t* tgti, const int32_t* cnt, int B, int fh, int fw,
                                         int num_dir_bins, float* losses, float* ws, void* stream) {
    TF_REQUIRE(pred && tgtf && tgti && cnt && losses && ws && B > 0 && num_dir_bins <= CN_MAXBINS, "tf_centernet_loss_fwd_f32: bad arguments");
    const int nb = nblocks((long)fh * fw, 256);
    TF_LAUNCH(centernet_loss_kernel<false>, dim3(nb), dim3(256), stream, pred, tgtf, tgti, cnt, B, fh, fw, num_dir_bins, ws, (const float*)nullptr,
              (float*)nullptr);
    TF_LAUNCH(centernet_loss_finalize_kernel, dim3(1), dim3(64), stream, (const float*)ws, nb, cnt, B, losses);
    return launch_status("tf_centernet_loss_fwd_f32");
}

extern "C" int tf_centernet_loss_bwd_f32(const float* pred, const float* tgtf, const int32_t* tgti, const int32_t* cnt, const float* gup, int B, int fh,
                                         int fw, int num_dir_bins, float* dpred, void* stream) {
    TF_REQUIRE(pred && tgtf && tgti && cnt && dpred && B > 0 && num_dir_bins <= CN_MAXBINS, "tf_centernet_loss_bwd_f32: bad arguments");
    const int nb = nblocks((long)fh * fw * B, 1024);
    TF_LAUNCH(centernet_loss_kernel<true>, dim3(nb), dim3(256), stream, pred, tgtf, tgti, cnt, B, fh, fw, num_dir_bins, (float*)nullptr, gup, dpred);
    return launch_status("tf_centernet_loss_bwd_f32");
}

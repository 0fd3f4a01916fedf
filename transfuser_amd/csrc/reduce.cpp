// Column reductions over NHWC activations + the per-channel elementwise passes that go with them
// (HBM-bound; SURVEY.md section 2.2 rows K4 BatchNorm, K5 Squeeze-Excite, global pools, bias gradients).
//
// Reduction skeleton: a (rows x C) row-major matrix is cut into column tiles of CTV vectors and
// row chunks; a block's 256 threads form (256/CTV) row lanes x CTV vector columns, so every
// wave reads whole contiguous row segments (coalesced float4), accumulates in registers,
// combines its row lanes through LDS and writes ONE partial per (chunk, column) to a workspace;
// a tiny finalize kernel sums the chunk partials in a fixed order (deterministic, no atomics).
#include <cstdlib>
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

template <int V> struct vecf { float v[V]; };
template <int V> __device__ __forceinline__ vecf<V> ldv(const float* p) {
    vecf<V> r;
    if (V == 4) { float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1 % V] = t.y; r.v[2 % V] = t.z; r.v[3 % V] = t.w; }
    else r.v[0] = *p;
    return r;
}
template <int V> __device__ __forceinline__ void stv(float* p, const vecf<V>& a) {
    if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1 % V], a.v[2 % V], a.v[3 % V]);
    else *p = a.v[0];
}

constexpr int kBnSlots = 16;    // accumulator copies of the atomically accumulated BatchNorm statistics
constexpr int kMaxChunks = 512;   // finalize is wave-parallel over chunks, so many small partials are cheap
constexpr long kWsFloats = 4L << 20;  // 16 MiB workspace (floats), see tf_workspace_bytes()
constexpr int kTickets = 4096;        // spare ints behind the kWsFloats floats (tf_workspace_bytes() keeps its value)

struct RedPlan { int V, CTV, coltiles, rpp, nchunks, rows_per_chunk; };

inline RedPlan plan_reduce(int rows, int C, int nseg, int nacc, bool allow_vec = true) {
    RedPlan p;
    p.V = (allow_vec && C % 4 == 0) ? 4 : 1;
    const int cv = C / p.V;
    int ctv = 1;
    for (int d = 1; d <= 64 && d <= cv; ++d)
        if (cv % d == 0) ctv = d;
    p.CTV = ctv;
    p.coltiles = cv / ctv;
    p.rpp = 256 / ctv;
    int want = 2048 / (p.coltiles * nseg);   // ~8 blocks per CU: a streaming reduction needs many loads in flight
    if (want < 1) want = 1;
    if (want > kMaxChunks) want = kMaxChunks;
    int maxc = cdiv(rows, p.rpp * 4);
    if (maxc < 1) maxc = 1;
    if (want > maxc) want = maxc;
    while ((long)nseg * want * nacc * C > kWsFloats / 2 && want > 1) --want;
    p.rows_per_chunk = cdiv(rows, want);
    p.nchunks = cdiv(rows, p.rows_per_chunk);
    return p;
}

inline void cap_chunks(RedPlan& p, int rows, int maxc) {
    if (p.nchunks > maxc) { p.rows_per_chunk = cdiv(rows, maxc); p.nchunks = cdiv(rows, p.rows_per_chunk); }
}

// ---- functors: eval(global_row, c, out[NACC]) ------------------------------------------------
template <int V> struct SumF {
    const float* x; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const { o[0] = ldv<V>(x + row * C + c); }
};
template <int V> struct BnStatF {  // shifted moments: d = x - K[c]
    const float* x; const float* K; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> a = ldv<V>(x + row * C + c), k = ldv<V>(K + c);
#pragma unroll
        for (int i = 0; i < V; ++i) { float d = a.v[i] - k.v[i]; o[0].v[i] = d; o[1].v[i] = d * d; }
    }
};
template <int V> struct BnBwdF {  // g = dz * [z > 0]; (g, g * xhat)
    const float* dz; const float* z; const float* x; const float* mean; const float* invstd; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> g = ldv<V>(dz + row * C + c), a = ldv<V>(x + row * C + c), m = ldv<V>(mean + c), s = ldv<V>(invstd + c);
        if (z) { vecf<V> zz = ldv<V>(z + row * C + c);
#pragma unroll
            for (int i = 0; i < V; ++i) if (!(zz.v[i] > 0.f)) g.v[i] = 0.f; }
#pragma unroll
        for (int i = 0; i < V; ++i) { o[0].v[i] = g.v[i]; o[1].v[i] = g.v[i] * ((a.v[i] - m.v[i]) * s.v[i]); }
    }
};
template <int V> struct MulF {  // dy * x (SE gate gradient), optional relu mask y on dy
    const float* dy; const float* x; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> a = ldv<V>(dy + row * C + c), b = ldv<V>(x + row * C + c);
#pragma unroll
        for (int i = 0; i < V; ++i) o[0].v[i] = a.v[i] * b.v[i];
    }
};
template <int V> struct MaskSumF {  // dy * [y > 0] (bias gradient behind a fused bias+ReLU epilogue)
    const float* dy; const float* y; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> a = ldv<V>(dy + row * C + c);
        if (y) { vecf<V> b = ldv<V>(y + row * C + c);
#pragma unroll
            for (int i = 0; i < V; ++i) if (!(b.v[i] > 0.f)) a.v[i] = 0.f; }
        o[0] = a;
    }
};

// ---- BatchNorm apply folded into its CONSUMERS (the conv2 -> BN -> ReLU -> SE part of a RegNetY bottleneck): z = max(x sc + sh, 0) is
// recomputed from the convolution output x and the layer's (scale | shift) wherever it is needed and never written to memory.
// The expression is the one bn_apply_kernel evaluates (multiply, then add: -ffp-contract=off), so a recomputed ReLU mask is bit-identical.
template <int V> struct BnReluSumF {   // SE squeeze of the BN + ReLU output
    const float* x; const float* coef; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> a = ldv<V>(x + row * C + c), sc = ldv<V>(coef + c), sh = ldv<V>(coef + C + c);
#pragma unroll
        for (int i = 0; i < V; ++i) o[0].v[i] = fmaxf(a.v[i] * sc.v[i] + sh.v[i], 0.f);
    }
};
template <int V> struct BnReluMulF {   // SE gate gradient: dy * z
    const float* dy; const float* x; const float* coef; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> g = ldv<V>(dy + row * C + c), a = ldv<V>(x + row * C + c), sc = ldv<V>(coef + c), sh = ldv<V>(coef + C + c);
#pragma unroll
        for (int i = 0; i < V; ++i) o[0].v[i] = g.v[i] * fmaxf(a.v[i] * sc.v[i] + sh.v[i], 0.f);
    }
};
template <int V> struct BnBwdRemaskF {  // BnBwdF with the ReLU mask recomputed from x: g = dz * [x sc + sh > 0]; (g, g * xhat)
    const float* dz; const float* x; const float* coef; const float* mean; const float* invstd; int C;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        vecf<V> g = ldv<V>(dz + row * C + c), a = ldv<V>(x + row * C + c), m = ldv<V>(mean + c), s = ldv<V>(invstd + c), sc = ldv<V>(coef + c), sh = ldv<V>(coef + C + c);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if (!(a.v[i] * sc.v[i] + sh.v[i] > 0.f)) g.v[i] = 0.f;
            o[0].v[i] = g.v[i];
            o[1].v[i] = g.v[i] * ((a.v[i] - m.v[i]) * s.v[i]);
        }
    }
};

// BnBwdRemaskF with the SE scale's backward folded in: the incoming gradient is dz = dy * sigmoid(gate[b][c]) + dmean[b][c] * inv_hw (tf_se_scale_bwd_x_f32),
// recomputed here instead of being written by its own pass; rows are sample-major (b = row / HW)
template <int V> struct SeBnBwdRemaskF {
    const float* dy; const float* gate; const float* dmean; const float* x; const float* coef; const float* mean; const float* invstd; int C; int HW; float inv_hw;
    __device__ __forceinline__ void eval(long row, int c, vecf<V>* o) const {
        const long bc = (row / HW) * C + c;
        vecf<V> g = ldv<V>(dy + row * C + c), a = ldv<V>(x + row * C + c), m = ldv<V>(mean + c), s = ldv<V>(invstd + c), sc = ldv<V>(coef + c), sh = ldv<V>(coef + C + c);
        vecf<V> gt = ldv<V>(gate + bc), dm = ldv<V>(dmean + bc);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float gi = g.v[i] * (1.f / (1.f + expf(-gt.v[i]))) + dm.v[i] * inv_hw;
            if (!(a.v[i] * sc.v[i] + sh.v[i] > 0.f)) gi = 0.f;
            o[0].v[i] = gi;
            o[1].v[i] = gi * ((a.v[i] - m.v[i]) * s.v[i]);
        }
    }
};

// (A "last block finishes" form of this kernel - partials at agent scope, a ticket per column tile, the last block finalizes - removed ~270 finalize
// launches per step and was measured SLOWER: 55.6 vs 51.6 ms/step, 66.7 with __threadfence(); DESIGN section 3.  It is gone since round 5.)
template <int V, int NACC, class F>
__global__ void __launch_bounds__(256) colreduce_kernel(F f, int rows_per_seg, int C, int CTV, int rows_per_chunk, float* __restrict__ ws, int atomic,
                                                        float scale) {
    __shared__ __attribute__((aligned(16))) float red[NACC][256][V];
    const int tid = threadIdx.x;
    const int rpp = 256 / CTV;
    const int cq = tid % CTV, rl = tid / CTV;
    const int c = (blockIdx.x * CTV + cq) * V;
    const int seg = blockIdx.z, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > rows_per_seg) r1 = rows_per_seg;
    vecf<V> acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[a].v[i] = 0.f;
    if (rl < rpp) {
        const long base = (long)seg * rows_per_seg;
        int r = r0 + rl;
        // four rows per trip, evaluated from clamped row indices before any of them is accumulated (the selection is applied to the VALUES):
        // the loads of the four rows are in flight together instead of one ~1 us round trip per row of a run-time-bounded loop
        if (atomic >= 0)
            for (; r < r1; r += 4 * rpp) {
                vecf<V> o[4][NACC];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + u * rpp;
                    f.eval(base + (rr < r1 ? rr : r1 - 1), c, o[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = r + u * rpp < r1;
#pragma unroll
                    for (int a = 0; a < NACC; ++a)
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[a].v[i] += in ? o[u][a].v[i] : 0.f;
                }
            }
        for (; r < r1; r += rpp) {          // atomic < 0: the one-row-per-trip loop (A/B switch TF_COLREDUCE_UNROLL=0)
            vecf<V> o[NACC];
            f.eval(base + r, c, o);
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int i = 0; i < V; ++i) acc[a].v[i] += o[a].v[i];
        }
    }
    if (atomic < 0) atomic = 0;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < V; ++i) red[a][tid][i] = acc[a].v[i];
    __syncthreads();
    if (rl == 0) {
        const int nch = gridDim.y;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            vecf<V> t;
#pragma unroll
            for (int i = 0; i < V; ++i) t.v[i] = 0.f;
            for (int j = 0; j < rpp; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) t.v[i] += red[a][j * CTV + cq][i];
            if (atomic == 2) {   // BatchNorm statistics: kBnSlots accumulator copies [slot][NACC][C] (the consumer adds them up in fp64) -
                                 // same-address atomics serialise in the memory-side atomic unit, and <= nchunks/kBnSlots fp32 adds per
                                 // copy keep the rounding at the level of a summation tree
                float* wsl = ws + (long)(chunk % kBnSlots) * NACC * C;
#pragma unroll
                for (int i = 0; i < V; ++i) atomicAdd(wsl + (long)a * C + c + i, t.v[i]);
            } else if (atomic) {   // ws = accumulators [seg][NACC][C] (zeroed, or a gradient being accumulated): no partials, no finalize pass
#pragma unroll
                for (int i = 0; i < V; ++i) atomicAdd(ws + ((long)seg * NACC + a) * C + c + i, t.v[i] * scale);
            } else {
                stv<V>(ws + (((long)seg * nch + chunk) * NACC + a) * C + c, t);
            }
        }
    }
}

template <int NACC, class F4, class F1>
inline void launch_reduce(const RedPlan& p, const F4& f4, const F1& f1, int rows_per_seg, int C, int nseg, float* ws, void* stream, int atomic = 0,
                          float scale = 1.f) {
    dim3 grid(p.coltiles, p.nchunks, nseg);
    static const bool unroll = [] { const char* e = getenv("TF_COLREDUCE_UNROLL"); return e ? e[0] != '0' : true; }();
    if (!unroll && atomic == 0) atomic = -1;
    if (p.V == 4) TF_LAUNCH((colreduce_kernel<4, NACC, F4>), grid, dim3(256), stream, f4, rows_per_seg, C, p.CTV, p.rows_per_chunk, ws, atomic, scale);
    else TF_LAUNCH((colreduce_kernel<1, NACC, F1>), grid, dim3(256), stream, f1, rows_per_seg, C, p.CTV, p.rows_per_chunk, ws, atomic, scale);
}


// ---- finalize kernels: one wave per channel, lanes sum the chunk partials (was: one thread per channel
// walking up to 64 dependent loads = ~19 us per BatchNorm; now a shuffle tree) -------------------------------------------
// (four loads per trip from clamped indices, the out-of-range ones multiplied by 0: a run-time-bounded loop of single loads is a chain of
//  ~1 us round trips - up to 8 of them for the 352-512 chunk reductions of the BatchNorm backward)
__device__ __forceinline__ float chunk_sum(const float* __restrict__ p, long stride, int nch, int lane, bool live) {
    float s = 0.f;
    if (live)
        for (int j = lane; j < nch; j += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int jj = j + 64 * u; v[u] = p[(long)(jj < nch ? jj : nch - 1) * stride]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (j + 64 * u < nch) ? v[u] : 0.f;
        }
    return wave_sum(s);
}
// two sums over the same chunks (the BatchNorm finalizes): the loads of both are in flight together
__device__ __forceinline__ void chunk_sum2(const float* __restrict__ p, const float* __restrict__ q, long stride, int nch, int lane, bool live, float& sp, float& sq) {
    float a = 0.f, b = 0.f;
    if (live)
        for (int j = lane; j < nch; j += 256) {
            float v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + 64 * u;
                const long o = (long)(jj < nch ? jj : nch - 1) * stride;
                v[u] = p[o]; w[u] = q[o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const bool in = j + 64 * u < nch; a += in ? v[u] : 0.f; b += in ? w[u] : 0.f; }
        }
    sp = wave_sum(a);
    sq = wave_sum(b);
}

// out[seg][c] (+)= scale * sum_chunks ws[seg][chunk][0][c]
__global__ void __launch_bounds__(256) colsum_finalize_kernel(const float* __restrict__ ws, float* __restrict__ out, int C, int nch, int nseg, float scale,
                                                              int accumulate) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = i < nseg * C;
    const int seg = live ? i / C : 0, c = live ? i - seg * C : 0;
    float s = chunk_sum(ws + (long)seg * nch * C + c, C, nch, lane, live) * scale;
    if (live && lane == 0) { if (accumulate) out[i] += s; else out[i] = s; }
}

// BN training statistics (torch BatchNorm2d train mode: biased var for normalisation, unbiased for
// running_var, momentum 0.1, eps 1e-5 - timm BatchNormAct2d).  coef = [scale | shift] (2*C).
__global__ void __launch_bounds__(256) bn_fwd_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ K, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                              float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ coef,
                                                              int C, int nch, float n, float momentum, float eps) {
    const int c0 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = c0 < C;
    const int c = live ? c0 : 0;
    float s1, s2;
    chunk_sum2(ws + c, ws + C + c, 2L * C, nch, lane, live, s1, s2);
    if (!live || lane != 0) return;
    const float dm = s1 / n;
    const float mean = K[c] + dm;
    float var = s2 / n - dm * dm;
    if (var < 0.f) var = 0.f;
    const float invstd = 1.0f / sqrtf(var + eps);
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    coef[c] = sc;
    coef[C + c] = beta[c] - mean * sc;
    if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        const float unb = (n > 1.f) ? var * (n / (n - 1.f)) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
}
// The same from per-part Welford triples written by the PRODUCING convolution's epilogue (tf_gemm_desc.colstat, tf_conv*_colstat_f32):
// parts[(p * 3 + {0: count, 1: mean, 2: M2}) * C + c].  WPC waves per channel (1: up to 256 parts, four channels per block; 4: a whole block per
// channel, up to 2048 parts - the stem / stage-1 / stage-2 layers hand over 320 .. 1250): a thread keeps ALL its parts in registers (NPT triples:
// every load of the kernel is issued in one batch), first the count-weighted mean over the whole channel, then
// M2 = sum_p M2_p + n_p (mean_p - mean)^2 (Chan et al.) against THAT mean - the two-pass form: one rounding in the mean, no cancellation
// whatever the mean / spread ratio of the activations.  Fixed summation order: bitwise reproducible.  (Round 3 walked more than 256 parts in a
// loop of dependent single loads: 20 .. 60 us per launch on exactly those layers; a single-pass pairwise Welford merge was tried this round
// and rejected: its running mean carries 2-3 roundings, which this network amplifies into measurably noisier gradients.)
template <int WPC>
__device__ __forceinline__ float chan_sum(float v, float* red) {          // all T = 64 * WPC threads of the channel -> the total, in every thread
    v = wave_sum(v);
    if (WPC == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < WPC; ++i) t += red[i];
    return t;
}
template <int WPC, int NPT>
__global__ void __launch_bounds__(256) bn_fwd_finalize_parts_kernel(const float* __restrict__ parts, int nparts, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                    float* __restrict__ coef, int C, float n, float momentum, float eps) {
    constexpr int T = 64 * WPC;                                   // threads per channel
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = WPC == 1 ? blockIdx.x * 4 + wave : blockIdx.x, t = WPC == 1 ? lane : (int)threadIdx.x;
    const bool live = c0 < C;
    const int c = live ? c0 : 0;
    const long ps = 3L * C;
    float np[NPT], mp[NPT], qp[NPT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int p = t + T * u;
        const long o = (long)(p < nparts ? p : nparts - 1) * ps + c;
        np[u] = parts[o]; mp[u] = parts[o + C]; qp[u] = parts[o + 2L * C];
        if (!live || p >= nparts) { np[u] = 0.f; qp[u] = 0.f; }
    }
    float sn = 0.f, sm = 0.f, m2 = 0.f;
#pragma unroll
    for (int u = 0; u < NPT; ++u) { sn += np[u]; sm += np[u] * mp[u]; }
    sn = chan_sum<WPC>(sn, red);
    sm = chan_sum<WPC>(sm, red);
    const float mean = sn > 0.f ? sm / sn : 0.f;
#pragma unroll
    for (int u = 0; u < NPT; ++u) { const float d = mp[u] - mean; m2 += qp[u] + np[u] * d * d; }
    m2 = chan_sum<WPC>(m2, red);
    if (!live || t != 0) return;
    float var = m2 / n;
    if (var < 0.f) var = 0.f;
    const float invstd = 1.0f / sqrtf(var + eps);
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    coef[c] = sc;
    coef[C + c] = beta[c] - mean * sc;
    if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        const float unb = (n > 1.f) ? var * (n / (n - 1.f)) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
}
__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rmean,
                                    const float* __restrict__ rvar, float* __restrict__ coef, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    coef[c] = sc;
    coef[C + c] = beta[c] - rmean[c] * sc;
}
// dgamma += sum g*xhat, dbeta += sum g; dx = A*g + Bc*x + Cc  (coef = [A | Bc | Cc])
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, int C, int nch, float n) {
    const int c0 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = c0 < C;
    const int c = live ? c0 : 0;
    float sg, sgx;
    chunk_sum2(ws + c, ws + C + c, 2L * C, nch, lane, live, sg, sgx);
    if (!live || lane != 0) return;
    if (dgamma) dgamma[c] += sgx;
    if (dbeta) dbeta[c] += sg;
    const float A = gamma[c] * invstd[c];
    const float Bc = -A * invstd[c] * (sgx / n);
    coef[c] = A;
    coef[C + c] = Bc;
    coef[2 * C + c] = -A * (sg / n) - Bc * mean[c];
}
// dgate_pre[b][c] = (sum_hw dy*x) * s * (1 - s), s = sigmoid(gate)
__global__ void __launch_bounds__(256) se_bwd_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ gate, float* __restrict__ dgate, int C,
                                                              int nch, int nseg) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = i < nseg * C;
    const int seg = live ? i / C : 0, c = live ? i - seg * C : 0;
    const float s = chunk_sum(ws + (long)seg * nch * C + c, C, nch, lane, live);
    if (live && lane == 0) {
        const float sg = 1.f / (1.f + expf(-gate[i]));
        dgate[i] = s * sg * (1.f - sg);
    }
}

// ---- per-channel elementwise passes (grid-stride over vectors) ------------------------------------
template <int V>
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ coef, const float* __restrict__ res,
                                                       float* __restrict__ y, long nvec, int C, int relu) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        vecf<V> a = ldv<V>(x + i * V), s = ldv<V>(coef + c), t = ldv<V>(coef + C + c);
        vecf<V> r;
        if (res) r = ldv<V>(res + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = a.v[k] * s.v[k] + t.v[k];
            if (res) v += r.v[k];
            if (relu) v = fmaxf(v, 0.f);
            a.v[k] = v;
        }
        stv<V>(y + i * V, a);
    }
}
template <int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ x,
                                                           const float* __restrict__ coef, float* __restrict__ dx, float* __restrict__ dres,
                                                           long nvec, int C) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        vecf<V> g = ldv<V>(dz + i * V), a = ldv<V>(x + i * V), A = ldv<V>(coef + c), Bc = ldv<V>(coef + C + c), Cc = ldv<V>(coef + 2 * C + c);
        if (z) { vecf<V> zz = ldv<V>(z + i * V);
#pragma unroll
            for (int k = 0; k < V; ++k) if (!(zz.v[k] > 0.f)) g.v[k] = 0.f; }
        if (dres) stv<V>(dres + i * V, g);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] = A.v[k] * g.v[k] + Bc.v[k] * a.v[k] + Cc.v[k];
        stv<V>(dx + i * V, a);
    }
}
// ---- single-pass-after-reduce BatchNorm: the statistics were ACCUMULATED with atomics into acc = [S1 | S2] (2*C floats), so there is
// no finalize kernel: threads own a fixed channel group (the colreduce thread layout), derive its scale / shift from the sums once,
// then stream their rows.  The chunk-0 block of every column tile also writes save_mean / save_invstd and the running statistics.
template <int V>
__global__ void __launch_bounds__(256) bn_apply_cols_kernel(const float* __restrict__ x, const float* __restrict__ acc, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                            float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                            const float* __restrict__ res, float* __restrict__ y, int rows, int C, int CTV,
                                                            int rows_per_chunk, float n, float momentum, float eps, int relu) {
    const int tid = threadIdx.x, rpp = 256 / CTV, cq = tid % CTV, rl = tid / CTV;
    if (rl >= rpp) return;
    const int c = (blockIdx.x * CTV + cq) * V;
    vecf<V> sc, sh;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int sl = 0; sl < kBnSlots; ++sl) { s1 += (double)acc[(long)sl * 2 * C + c + i]; s2 += (double)acc[(long)sl * 2 * C + C + c + i]; }
        const double dmd = s1 / (double)n;
        const float dm = (float)dmd;
        const float mean = x[c + i] + dm;               // shift K = first row of x (same as BnStatF)
        float var = (float)(s2 / (double)n - dmd * dmd);
        if (var < 0.f) var = 0.f;
        const float invstd = 1.0f / sqrtf(var + eps);
        sc.v[i] = gamma[c + i] * invstd;
        sh.v[i] = beta[c + i] - mean * sc.v[i];
        if (blockIdx.y == 0 && rl == 0) {
            save_mean[c + i] = mean;
            save_invstd[c + i] = invstd;
            if (rmean) {
                rmean[c + i] = (1.f - momentum) * rmean[c + i] + momentum * mean;
                const float unb = (n > 1.f) ? var * (n / (n - 1.f)) : var;
                rvar[c + i] = (1.f - momentum) * rvar[c + i] + momentum * unb;
            }
        }
    }
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    for (int r = r0 + rl; r < r1; r += rpp) {
        const long o = (long)r * C + c;
        vecf<V> a = ldv<V>(x + o), rr;
        if (res) rr = ldv<V>(res + o);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = a.v[k] * sc.v[k] + sh.v[k];
            if (res) v += rr.v[k];
            if (relu) v = fmaxf(v, 0.f);
            a.v[k] = v;
        }
        stv<V>(y + o, a);
    }
}
// acc = [sum g | sum g*xhat]; dx = A*g + Bc*x + Cc; dgamma += sum g*xhat, dbeta += sum g (chunk-0 blocks)
template <int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_cols_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ x,
                                                                const float* __restrict__ acc, const float* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dx,
                                                                float* __restrict__ dres, int rows, int C, int CTV, int rows_per_chunk, float n) {
    const int tid = threadIdx.x, rpp = 256 / CTV, cq = tid % CTV, rl = tid / CTV;
    if (rl >= rpp) return;
    const int c = (blockIdx.x * CTV + cq) * V;
    vecf<V> A, Bc, Cc;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int sl = 0; sl < kBnSlots; ++sl) { d1 += (double)acc[(long)sl * 2 * C + c + i]; d2 += (double)acc[(long)sl * 2 * C + C + c + i]; }
        const float sg = (float)d1, sgx = (float)d2;
        A.v[i] = gamma[c + i] * invstd[c + i];
        Bc.v[i] = -A.v[i] * invstd[c + i] * (sgx / n);
        Cc.v[i] = -A.v[i] * (sg / n) - Bc.v[i] * mean[c + i];
        if (blockIdx.y == 0 && rl == 0) {
            if (dgamma) dgamma[c + i] += sgx;
            if (dbeta) dbeta[c + i] += sg;
        }
    }
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    for (int r = r0 + rl; r < r1; r += rpp) {
        const long o = (long)r * C + c;
        vecf<V> g = ldv<V>(dz + o), a = ldv<V>(x + o);
        if (z) { vecf<V> zz = ldv<V>(z + o);
#pragma unroll
            for (int k = 0; k < V; ++k) if (!(zz.v[k] > 0.f)) g.v[k] = 0.f; }
        if (dres) stv<V>(dres + o, g);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] = A.v[k] * g.v[k] + Bc.v[k] * a.v[k] + Cc.v[k];
        stv<V>(dx + o, a);
    }
}
// y = x * sigmoid(gate[b][c])
template <int V>
__global__ void __launch_bounds__(256) se_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate, float* __restrict__ y, long nvec,
                                                       int C, long vec_per_b) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        const long b = i / vec_per_b;
        vecf<V> a = ldv<V>(x + i * V), g = ldv<V>(gate + b * C + c);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] = a.v[k] * (1.f / (1.f + expf(-g.v[k])));
        stv<V>(y + i * V, a);
    }
}
// y = max(x sc + sh, 0) * sigmoid(gate[b][c]): BatchNorm apply + ReLU + SE scale in one pass (the BN output itself is never stored)
template <int V>
__global__ void __launch_bounds__(256) se_scale_bn_kernel(const float* __restrict__ x, const float* __restrict__ coef, const float* __restrict__ gate,
                                                          float* __restrict__ y, long nvec, int C, long vec_per_b) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        const long b = i / vec_per_b;
        vecf<V> a = ldv<V>(x + i * V), g = ldv<V>(gate + b * C + c), sc = ldv<V>(coef + c), sh = ldv<V>(coef + C + c);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] = fmaxf(a.v[k] * sc.v[k] + sh.v[k], 0.f) * (1.f / (1.f + expf(-g.v[k])));
        stv<V>(y + i * V, a);
    }
}
// bn_bwd_apply_kernel with the ReLU mask recomputed from x and the forward (scale | shift)
template <int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_remask_kernel(const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ fcoef,
                                                                  const float* __restrict__ coef, float* __restrict__ dx, long nvec, int C) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        vecf<V> g = ldv<V>(dz + i * V), a = ldv<V>(x + i * V), A = ldv<V>(coef + c), Bc = ldv<V>(coef + C + c), Cc = ldv<V>(coef + 2 * C + c);
        vecf<V> sc = ldv<V>(fcoef + c), sh = ldv<V>(fcoef + C + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (!(a.v[k] * sc.v[k] + sh.v[k] > 0.f)) g.v[k] = 0.f;
            a.v[k] = A.v[k] * g.v[k] + Bc.v[k] * a.v[k] + Cc.v[k];
        }
        stv<V>(dx + i * V, a);
    }
}
// bn_bwd_apply_remask_kernel on dz = dy * sigmoid(gate) + dmean * inv_hw (SeBnBwdRemaskF): dz itself is never written
template <int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_remask_se_kernel(const float* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dmean,
                                                                     const float* __restrict__ x, const float* __restrict__ fcoef, const float* __restrict__ coef,
                                                                     float* __restrict__ dx, long nvec, int C, long vec_per_b, float inv_hw) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        const long bc = (i / vec_per_b) * C + c;
        vecf<V> g = ldv<V>(dy + i * V), a = ldv<V>(x + i * V), A = ldv<V>(coef + c), Bc = ldv<V>(coef + C + c), Cc = ldv<V>(coef + 2 * C + c);
        vecf<V> sc = ldv<V>(fcoef + c), sh = ldv<V>(fcoef + C + c), gt = ldv<V>(gate + bc), dm = ldv<V>(dmean + bc);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float gi = g.v[k] * (1.f / (1.f + expf(-gt.v[k]))) + dm.v[k] * inv_hw;
            if (!(a.v[k] * sc.v[k] + sh.v[k] > 0.f)) gi = 0.f;
            a.v[k] = A.v[k] * gi + Bc.v[k] * a.v[k] + Cc.v[k];
        }
        stv<V>(dx + i * V, a);
    }
}
// dx (+)= dy * sigmoid(gate[b][c]) + dmean[b][c] * inv_hw      (either term optional)
template <int V>
__global__ void __launch_bounds__(256) se_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dmean,
                                                           float* __restrict__ dx, long nvec, int C, long vec_per_b, float inv_hw, int accumulate) {
    const int cv = C / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        const long b = i / vec_per_b;
        vecf<V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = 0.f;
        if (dy) {
            vecf<V> a = ldv<V>(dy + i * V), g = ldv<V>(gate + b * C + c);
#pragma unroll
            for (int k = 0; k < V; ++k) o.v[k] = a.v[k] * (1.f / (1.f + expf(-g.v[k])));
        }
        if (dmean) {
            vecf<V> m = ldv<V>(dmean + b * C + c);
#pragma unroll
            for (int k = 0; k < V; ++k) o.v[k] += m.v[k] * inv_hw;
        }
        if (accumulate) { vecf<V> p = ldv<V>(dx + i * V);
#pragma unroll
            for (int k = 0; k < V; ++k) o.v[k] += p.v[k]; }
        stv<V>(dx + i * V, o);
    }
}

inline int ew_blocks(long nvec) {
    long b = (nvec + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- several column sums over the SAME rows in ONE single-pass launch (the four bias gradients of a transformer Block, transfuser.py:500-541:
// fc2 / fc1 / proj / key-query-value): out_e[c] += sum_r x_e[r][c].  A block owns one 32-column strip (one 128-byte line per row) of one
// entry and ALL the rows: 8 column threads x 32 row lanes, 8 loads in flight per thread, the row lanes are combined through LDS in a fixed
// order - no second stage, no atomics, bitwise reproducible.  Replaces 2 launches (reduce + finalize) per bias on the single-stream
// critical path of the GPT stages.
inline void launch_finalize_parts(const float* parts, int nparts, const float* gamma, const float* beta, float* rmean, float* rvar, float* save_mean,
                                  float* save_invstd, float* coef, int C, float n, float momentum, float eps, void* stream) {
#define TF_FIN(WPC_, NPT_, GRID_) TF_LAUNCH((bn_fwd_finalize_parts_kernel<WPC_, NPT_>), dim3(GRID_), dim3(256), stream, parts, nparts, gamma, beta, rmean, rvar, save_mean, save_invstd, coef, C, n, momentum, eps)
    if (nparts <= 256) TF_FIN(1, 4, cdiv(C, 4));
    else if (nparts <= 1024) TF_FIN(4, 4, C);
    else if (nparts <= 2048) TF_FIN(4, 8, C);
    else TF_FIN(4, 64, C);               // up to 16384 parts (524288 rows at 32 rows per part): registers / scratch, never reached by the trunks (<= 40000 rows)
#undef TF_FIN
}
constexpr int kMultiMax = 8;
struct MultiSum { const float* x[kMultiMax]; float* out[kMultiMax]; int C[kMultiMax]; long ld[kMultiMax]; int strip0[kMultiMax + 1]; int n; };
__global__ void __launch_bounds__(256) colsum_multi_kernel(MultiSum d, int rows) {
    __shared__ float4 red[32][8];
    int e = 0;
#pragma unroll
    for (int i = 1; i < kMultiMax; ++i)
        if (i < d.n && (int)blockIdx.x >= d.strip0[i]) e = i;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int C = d.C[e];
    const int c = ((int)blockIdx.x - d.strip0[e]) * 32 + tx * 4;
    const bool live = c < C;                        // C % 4 == 0: a float4 is all in or all out
    const float* p = d.x[e] + (live ? c : 0);
    const long ld = d.ld[e];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = ty;
    for (; r + 7 * 32 < rows; r += 8 * 32) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(r + 32 * u) * ld);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; r < rows; r += 32) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long)r * ld);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && live) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < 32; ++j) { const float4 v = red[j][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float* o = d.out[e] + c;
        o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
    }
}

// ---- 16-bit operand copies written by the element-wise PRODUCERS of a bottleneck's 1x1 convolutions (round 5; bf16 / fp16 storage modes).
// In those modes conv1 / conv3 of a RegNetY Bottleneck (timm, transfuser.py:380,442) run as packed-16 NT GEMMs (tf_gemm16_nt_f32) like the GPT
// linear layers: y = x W^T needs x as a row-major 16-bit matrix, dW = dy^T x needs dy AND x transposed, dx = dy W needs dy row-major.  A cast
// launch per operand costs more than the packed GEMMs gain (measured in round 4), so the four element-wise passes that produce those operands
// write the copies themselves: BatchNorm apply (+ shortcut + ReLU: the block output = the next block's conv1 input), BatchNorm apply + ReLU + SE scale
// (conv3's input), and the two BatchNorm backward applies (the gradients entering conv3 / conv1).  One 64 x 64 tile per block: the value of an element
// is computed ONCE by the functor (the expression of the fp32 kernel it replaces, so the copy is bitwise cast16 of that kernel's output), stored as
// fp32 where something still reads it, packed row-major, and transposed through LDS (rows zero-padded to a multiple of 8) - cast16.cpp's tile.
constexpr int T16S = 64, T16P = T16S + 8;
template <class F>
__global__ void __launch_bounds__(256) tile16_kernel(F f, int rows, int C, uint16_t* __restrict__ y, uint16_t* __restrict__ yt, long ldyt, int f16) {
    __shared__ __attribute__((aligned(16))) uint16_t T[T16S * T16P];
    const int tid = threadIdx.x, r0 = blockIdx.y * T16S, c0 = blockIdx.x * T16S, c4 = (tid & 15) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 16 + (tid >> 4), r = r0 + row, c = c0 + c4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool in = r < rows && c < C;          // C % 4 == 0: a float4 never straddles the edge
        if (in) f(r, c, v);
        uint16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = cvt16_bits(v[e], f16 != 0);
        if (y && in) *reinterpret_cast<float2*>(y + (long)r * C + c) = make_float2(__uint_as_float((uint32_t)h[0] | ((uint32_t)h[1] << 16)), __uint_as_float((uint32_t)h[2] | ((uint32_t)h[3] << 16)));
        if (yt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) T[(c4 + e) * T16P + row] = h[e];
        }
    }
    if (!yt) return;
    __syncthreads();
    const int rows8 = (rows + 7) & ~7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int id = p * 256 + tid, col = id >> 3, rc = (id & 7) * 8;
        if (c0 + col < C && r0 + rc < rows8) *reinterpret_cast<float4*>(yt + (long)(c0 + col) * ldyt + r0 + rc) = *reinterpret_cast<const float4*>(T + col * T16P + rc);
    }
}
struct BnApply16F {          // bn_apply_kernel: y = x sc + sh (+ res) (relu); y32 (may be NULL) keeps the fp32 result
    const float* x; const float* coef; const float* res; float* y32; int C; int relu;
    __device__ __forceinline__ void operator()(int r, int c, float (&v)[4]) const {
        const long o = (long)r * C + c;
        const vecf<4> a = ldv<4>(x + o), s = ldv<4>(coef + c), t = ldv<4>(coef + C + c);
        vecf<4> rr;
        if (res) rr = ldv<4>(res + o);
        vecf<4> out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w = a.v[k] * s.v[k] + t.v[k];
            if (res) w += rr.v[k];
            if (relu) w = fmaxf(w, 0.f);
            out.v[k] = v[k] = w;
        }
        if (y32) stv<4>(y32 + o, out);
    }
};
struct SeScaleBn16F {        // se_scale_bn_kernel: y = max(x sc + sh, 0) * sigmoid(gate[b][c]), b = row / HW
    const float* x; const float* coef; const float* gate; int C; int HW;
    __device__ __forceinline__ void operator()(int r, int c, float (&v)[4]) const {
        const long o = (long)r * C + c;
        const vecf<4> a = ldv<4>(x + o), g = ldv<4>(gate + (long)(r / HW) * C + c), sc = ldv<4>(coef + c), sh = ldv<4>(coef + C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(a.v[k] * sc.v[k] + sh.v[k], 0.f) * (1.f / (1.f + expf(-g.v[k])));
    }
};
struct BnBwdApply16F {       // bn_bwd_apply_kernel: g = dz [z > 0] (-> dres), dx = A g + Bc x + Cc; dx32 (may be NULL) keeps the fp32 result
    const float* dz; const float* z; const float* x; const float* coef; float* dx32; float* dres; int C;
    __device__ __forceinline__ void operator()(int r, int c, float (&v)[4]) const {
        const long o = (long)r * C + c;
        vecf<4> g = ldv<4>(dz + o);
        const vecf<4> a = ldv<4>(x + o), A = ldv<4>(coef + c), Bc = ldv<4>(coef + C + c), Cc = ldv<4>(coef + 2 * C + c);
        if (z) { const vecf<4> zz = ldv<4>(z + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (!(zz.v[k] > 0.f)) g.v[k] = 0.f; }
        if (dres) stv<4>(dres + o, g);
        vecf<4> out;
#pragma unroll
        for (int k = 0; k < 4; ++k) out.v[k] = v[k] = A.v[k] * g.v[k] + Bc.v[k] * a.v[k] + Cc.v[k];
        if (dx32) stv<4>(dx32 + o, out);
    }
};
struct BnBwdRemask16F {      // bn_bwd_apply_remask_kernel: the ReLU mask recomputed from x and the forward (scale | shift)
    const float* dz; const float* x; const float* fcoef; const float* coef; float* dx32; int C;
    __device__ __forceinline__ void operator()(int r, int c, float (&v)[4]) const {
        const long o = (long)r * C + c;
        vecf<4> g = ldv<4>(dz + o);
        const vecf<4> a = ldv<4>(x + o), A = ldv<4>(coef + c), Bc = ldv<4>(coef + C + c), Cc = ldv<4>(coef + 2 * C + c), sc = ldv<4>(fcoef + c), sh = ldv<4>(fcoef + C + c);
        vecf<4> out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(a.v[k] * sc.v[k] + sh.v[k] > 0.f)) g.v[k] = 0.f;
            out.v[k] = v[k] = A.v[k] * g.v[k] + Bc.v[k] * a.v[k] + Cc.v[k];
        }
        if (dx32) stv<4>(dx32 + o, out);
    }
};
template <class F>
inline void launch_tile16(const F& f, int rows, int C, void* y16, void* y16t, int ldyt, int dtype, void* stream) {
    TF_LAUNCH(tile16_kernel<F>, dim3(cdiv(C, T16S), cdiv(rows, T16S)), dim3(256), stream, f, rows, C, (uint16_t*)y16, (uint16_t*)y16t, (long)ldyt, dtype == 2 ? 1 : 0);
}
inline bool tile16_ok(int rows, int C, const void* y16, const void* y16t, int ldyt, int dtype) {
    return rows > 0 && C > 0 && C % 4 == 0 && (dtype == 1 || dtype == 2) && (y16 || y16t) && (!y16 || (((uintptr_t)y16) & 7) == 0) &&
           (!y16t || (aligned16(y16t) && ldyt % 8 == 0 && ldyt >= ((rows + 7) & ~7)));
}
}  // namespace

extern "C" long tf_workspace_bytes(void) { return (kWsFloats + kTickets) * 4; }
extern "C" int tf_bn_zacc_floats(int C) { return kBnSlots * 2 * C; }

// BatchNorm2d forward on NHWC (rows = B*H*W).  training: batch statistics (+ running update);
// eval: running statistics.  y = bn(x) (+res) (relu).  ws: tf_workspace_bytes() scratch.
extern "C" int tf_bn_fwd_f32(const float* x, int rows, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, const float* res, int relu, float* y, float* save_mean, float* save_invstd, float* ws,
                             int training, float* zacc, void* stream) {
    TF_REQUIRE(x && gamma && beta && y && ws && rows > 0 && C > 0, "tf_bn_fwd_f32: bad arguments");
    float* coef = ws + kWsFloats / 2;
    if (training && zacc) {   // 2 launches: atomically accumulated statistics, then a column-owned normalise pass (no finalize kernel)
        TF_REQUIRE(save_mean && save_invstd, "tf_bn_fwd_f32: training needs save_mean/save_invstd");
        const bool v4ok = aligned16(x) && aligned16(y) && (!res || aligned16(res));
        RedPlan p = plan_reduce(rows, C, 1, 2, v4ok);
        BnStatF<4> f4{x, x, C};
        BnStatF<1> f1{x, x, C};
        launch_reduce<2>(p, f4, f1, rows, C, 1, zacc, stream, 2, 1.f);
        dim3 grid(p.coltiles, p.nchunks);
        if (p.V == 4) TF_LAUNCH(bn_apply_cols_kernel<4>, grid, dim3(256), stream, x, (const float*)zacc, gamma, beta, running_mean, running_var, save_mean,
                                save_invstd, res, y, rows, C, p.CTV, p.rows_per_chunk, (float)rows, momentum, eps, relu);
        else TF_LAUNCH(bn_apply_cols_kernel<1>, grid, dim3(256), stream, x, (const float*)zacc, gamma, beta, running_mean, running_var, save_mean,
                       save_invstd, res, y, rows, C, p.CTV, p.rows_per_chunk, (float)rows, momentum, eps, relu);
        return launch_status("tf_bn_fwd_f32");
    }
    if (training) {
        TF_REQUIRE(save_mean && save_invstd, "tf_bn_fwd_f32: training needs save_mean/save_invstd");
        RedPlan p = plan_reduce(rows, C, 1, 2);
        BnStatF<4> f4{x, x, C};  // shift K = first row of x
        BnStatF<1> f1{x, x, C};
        launch_reduce<2>(p, f4, f1, rows, C, 1, ws, stream);
        TF_LAUNCH(bn_fwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), stream, (const float*)ws, x, gamma, beta, running_mean, running_var,
                  save_mean, save_invstd, coef, C, p.nchunks, (float)rows, momentum, eps);
    } else {
        TF_REQUIRE(running_mean && running_var, "tf_bn_fwd_f32: eval needs running statistics");
        TF_LAUNCH(bn_eval_coef_kernel, dim3(cdiv(C, 256)), dim3(256), stream, gamma, beta, (const float*)running_mean, (const float*)running_var,
                  coef, C, eps);
    }
    const bool v4 = (C % 4 == 0) && aligned16(x) && aligned16(y) && (!res || aligned16(res));
    const long n = (long)rows * C;
    if (v4) TF_LAUNCH(bn_apply_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, x, (const float*)coef, res, y, n / 4, C, relu);
    else TF_LAUNCH(bn_apply_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, x, (const float*)coef, res, y, n, C, relu);
    return launch_status("tf_bn_fwd_f32");
}

// Train-mode BatchNorm2d forward when the statistics of x were already gathered by the kernel that PRODUCED x (per-part Welford triples,
// see bn_fwd_finalize_parts_kernel): finalize + apply, no pass over x for the moments.  y = bn(x) (+res) (relu).
extern "C" int tf_bn_fwd_parts_f32(const float* x, int rows, int C, const float* parts, int nparts, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, const float* res, int relu, float* y,
                                   float* save_mean, float* save_invstd, float* ws, void* stream) {
    TF_REQUIRE(x && parts && nparts > 0 && nparts <= 16384 && gamma && beta && y && save_mean && save_invstd && ws && rows > 0 && C > 0, "tf_bn_fwd_parts_f32: bad arguments");
    float* coef = ws + kWsFloats / 2;
    launch_finalize_parts(parts, nparts, gamma, beta, running_mean, running_var, save_mean, save_invstd, coef, C, (float)rows, momentum, eps, stream);
    const bool v4 = (C % 4 == 0) && aligned16(x) && aligned16(y) && (!res || aligned16(res));
    const long n = (long)rows * C;
    if (v4) TF_LAUNCH(bn_apply_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, x, (const float*)coef, res, y, n / 4, C, relu);
    else TF_LAUNCH(bn_apply_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, x, (const float*)coef, res, y, n, C, relu);
    return launch_status("tf_bn_fwd_parts_f32");
}

// BatchNorm2d backward (training statistics).  dz: grad of the (post-residual, post-ReLU) output;
// z: that output when a ReLU followed (mask) else NULL.  dgamma/dbeta are ACCUMULATED.
extern "C" int tf_bn_bwd_f32(const float* dz, const float* z, const float* x, int rows, int C, const float* gamma, const float* save_mean,
                             const float* save_invstd, float* dx, float* dres, float* dgamma, float* dbeta, float* ws, float* zacc, void* stream) {
    TF_REQUIRE(dz && x && gamma && save_mean && save_invstd && dx && ws && rows > 0 && C > 0, "tf_bn_bwd_f32: bad arguments");
    float* coef = ws + kWsFloats / 2;
    const bool v4ok = aligned16(dz) && aligned16(x) && aligned16(dx) && (!z || aligned16(z)) && (!dres || aligned16(dres));
    RedPlan p = plan_reduce(rows, C, 1, 2, zacc ? v4ok : true);
    BnBwdF<4> f4{dz, z, x, save_mean, save_invstd, C};
    BnBwdF<1> f1{dz, z, x, save_mean, save_invstd, C};
    if (zacc) {
        launch_reduce<2>(p, f4, f1, rows, C, 1, zacc, stream, 2, 1.f);
        dim3 grid(p.coltiles, p.nchunks);
        if (p.V == 4) TF_LAUNCH(bn_bwd_apply_cols_kernel<4>, grid, dim3(256), stream, dz, z, x, (const float*)zacc, gamma, save_mean, save_invstd, dgamma,
                                dbeta, dx, dres, rows, C, p.CTV, p.rows_per_chunk, (float)rows);
        else TF_LAUNCH(bn_bwd_apply_cols_kernel<1>, grid, dim3(256), stream, dz, z, x, (const float*)zacc, gamma, save_mean, save_invstd, dgamma, dbeta, dx,
                       dres, rows, C, p.CTV, p.rows_per_chunk, (float)rows);
        return launch_status("tf_bn_bwd_f32");
    }
    launch_reduce<2>(p, f4, f1, rows, C, 1, ws, stream);
    TF_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), stream, (const float*)ws, gamma, save_mean, save_invstd, dgamma, dbeta, coef,
              C, p.nchunks, (float)rows);
    const bool v4 = (C % 4 == 0) && aligned16(dz) && aligned16(x) && aligned16(dx) && (!z || aligned16(z)) && (!dres || aligned16(dres));
    const long n = (long)rows * C;
    if (v4) TF_LAUNCH(bn_bwd_apply_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, dz, z, x, (const float*)coef, dx, dres, n / 4, C);
    else TF_LAUNCH(bn_bwd_apply_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, dz, z, x, (const float*)coef, dx, dres, n, C);
    return launch_status("tf_bn_bwd_f32");
}

// out[seg][c] (+)= scale * sum over the segment's rows of x (* [mask > 0]).  Uses: SE squeeze /
// global average pool (scale = 1/HW, nseg = B), bias gradients (nseg = 1, mask = ReLU output).
extern "C" int tf_colsum_f32(const float* x, const float* mask, int nseg, int rows_per_seg, int C, float scale, float* out, int accumulate,
                             float* ws, void* stream) {
    TF_REQUIRE(x && out && ws && nseg > 0 && rows_per_seg > 0 && C > 0, "tf_colsum_f32: bad arguments");
    RedPlan p = plan_reduce(rows_per_seg, C, nseg, 1, aligned16(x) && (!mask || aligned16(mask)));
    MaskSumF<4> f4{x, mask, C};
    MaskSumF<1> f1{x, mask, C};
    static const bool atomic_colsum = [] { const char* e = getenv("TF_ATOMIC_COLSUM"); return e ? e[0] != '0' : false; }();   // measured slower on the MI355X (165.4 vs 167.1 samples/s): opt-in
    if (accumulate && atomic_colsum) {   // += : the blocks add straight into the destination with fp32 atomics - one launch, no partials, no finalize
        launch_reduce<1>(p, f4, f1, rows_per_seg, C, nseg, out, stream, 1, scale);
        return launch_status("tf_colsum_f32");
    }
    launch_reduce<1>(p, f4, f1, rows_per_seg, C, nseg, ws, stream);
    TF_LAUNCH(colsum_finalize_kernel, dim3(cdiv((long)nseg * C, 4)), dim3(256), stream, (const float*)ws, out, C, p.nchunks, nseg, scale, accumulate);
    return launch_status("tf_colsum_f32");
}

// Squeeze-Excite scale (timm SEModule): y = x * sigmoid(gate[b][c]); x: (B, HW, C), gate: (B, C) pre-sigmoid.
extern "C" int tf_se_scale_fwd_f32(const float* x, const float* gate, float* y, int B, int HW, int C, void* stream) {
    TF_REQUIRE(x && gate && y && B > 0 && HW > 0 && C > 0, "tf_se_scale_fwd_f32: bad arguments");
    const long n = (long)B * HW * C;
    if (C % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gate))
        TF_LAUNCH(se_scale_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, x, gate, y, n / 4, C, (long)HW * C / 4);
    else
        TF_LAUNCH(se_scale_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, x, gate, y, n, C, (long)HW * C);
    return launch_status("tf_se_scale_fwd_f32");
}
// dgate_pre[b][c] = (sum_hw dy * x) * s(1-s)
extern "C" int tf_se_scale_bwd_gate_f32(const float* dy, const float* x, const float* gate, float* dgate, int B, int HW, int C, float* ws, void* stream) {
    TF_REQUIRE(dy && x && gate && dgate && ws && B > 0 && HW > 0 && C > 0, "tf_se_scale_bwd_gate_f32: bad arguments");
    RedPlan p = plan_reduce(HW, C, B, 1);
    MulF<4> f4{dy, x, C};
    MulF<1> f1{dy, x, C};
    launch_reduce<1>(p, f4, f1, HW, C, B, ws, stream);
    TF_LAUNCH(se_bwd_finalize_kernel, dim3(cdiv((long)B * C, 4)), dim3(256), stream, (const float*)ws, gate, dgate, C, p.nchunks, B);
    return launch_status("tf_se_scale_bwd_gate_f32");
}
// dx (+)= dy * sigmoid(gate) + dmean / HW   (dy/gate pair optional, dmean optional): the input
// gradient of the SE block, and of a global average pool (dy = NULL).
extern "C" int tf_se_scale_bwd_x_f32(const float* dy, const float* gate, const float* dmean, float* dx, int B, int HW, int C, int accumulate,
                                     void* stream) {
    TF_REQUIRE(dx && (dy || dmean) && (!dy || gate) && B > 0 && HW > 0 && C > 0, "tf_se_scale_bwd_x_f32: bad arguments");
    const long n = (long)B * HW * C;
    const bool v4 = C % 4 == 0 && aligned16(dx) && (!dy || (aligned16(dy) && aligned16(gate))) && (!dmean || aligned16(dmean));
    if (v4) TF_LAUNCH(se_bwd_apply_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, dy, gate, dmean, dx, n / 4, C, (long)HW * C / 4, 1.f / HW, accumulate);
    else TF_LAUNCH(se_bwd_apply_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, dy, gate, dmean, dx, n, C, (long)HW * C, 1.f / HW, accumulate);
    return launch_status("tf_se_scale_bwd_x_f32");
}

// ---- BatchNorm apply folded into the consumers (YBlockFn's conv2 -> BN -> ReLU -> SE -> conv3 segment)
// finalize only: save_mean / save_invstd / running statistics + coef_out = [scale | shift] (2 C floats the CALLER owns: they are read by the
// consumers of this layer in the forward AND the backward pass)
extern "C" int tf_bn_finalize_parts_f32(const float* parts, int nparts, int rows, int C, const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, float momentum, float eps, float* save_mean, float* save_invstd, float* coef_out, void* stream) {
    TF_REQUIRE(parts && nparts > 0 && nparts <= 16384 && gamma && beta && save_mean && save_invstd && coef_out && rows > 0 && C > 0, "tf_bn_finalize_parts_f32: bad arguments");
    launch_finalize_parts(parts, nparts, gamma, beta, running_mean, running_var, save_mean, save_invstd, coef_out, C, (float)rows, momentum, eps, stream);
    return launch_status("tf_bn_finalize_parts_f32");
}
// partial column sums (one per row chunk) of max(x sc + sh, 0) per segment: ws[(seg * nchunks + chunk) * C + c]; *nchunks is set on the host.
// The consumer (tf_se_excite_fwd_parts_f32) adds the chunks up itself: no finalize launch.
extern "C" int tf_colsum_bnrelu_parts_f32(const float* x, const float* coef, int nseg, int rows_per_seg, int C, float* ws, int* nchunks, void* stream) {
    TF_REQUIRE(x && coef && ws && nchunks && nseg > 0 && rows_per_seg > 0 && C > 0, "tf_colsum_bnrelu_parts_f32: bad arguments");
    RedPlan p = plan_reduce(rows_per_seg, C, nseg, 1, aligned16(x) && aligned16(coef));
    cap_chunks(p, rows_per_seg, 8);      // the consumer sums the chunks serially: few, fat chunks (nseg x coltiles x 8 blocks still fill the GPU)
    BnReluSumF<4> f4{x, coef, C};
    BnReluSumF<1> f1{x, coef, C};
    launch_reduce<1>(p, f4, f1, rows_per_seg, C, nseg, ws, stream);
    *nchunks = p.nchunks;
    return launch_status("tf_colsum_bnrelu_parts_f32");
}
// the same for the SE gate gradient: partial sums over hw of dy * max(x sc + sh, 0) per sample
extern "C" int tf_se_gate_grad_parts_f32(const float* dy, const float* x, const float* coef, int B, int HW, int C, float* ws, int* nchunks, void* stream) {
    TF_REQUIRE(dy && x && coef && ws && nchunks && B > 0 && HW > 0 && C > 0, "tf_se_gate_grad_parts_f32: bad arguments");
    RedPlan p = plan_reduce(HW, C, B, 1, aligned16(x) && aligned16(coef) && aligned16(dy));
    cap_chunks(p, HW, 8);
    BnReluMulF<4> f4{dy, x, coef, C};
    BnReluMulF<1> f1{dy, x, coef, C};
    launch_reduce<1>(p, f4, f1, HW, C, B, ws, stream);
    *nchunks = p.nchunks;
    return launch_status("tf_se_gate_grad_parts_f32");
}
extern "C" int tf_se_scale_bn_fwd_f32(const float* x, const float* coef, const float* gate, float* y, int B, int HW, int C, void* stream) {
    TF_REQUIRE(x && coef && gate && y && B > 0 && HW > 0 && C > 0, "tf_se_scale_bn_fwd_f32: bad arguments");
    const long n = (long)B * HW * C;
    if (C % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gate) && aligned16(coef))
        TF_LAUNCH(se_scale_bn_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, x, coef, gate, y, n / 4, C, (long)HW * C / 4);
    else
        TF_LAUNCH(se_scale_bn_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, x, coef, gate, y, n, C, (long)HW * C);
    return launch_status("tf_se_scale_bn_fwd_f32");
}
// BatchNorm backward behind a ReLU whose output was never stored: the mask is [x sc + sh > 0] with the forward's (scale | shift) = fcoef
extern "C" int tf_bn_bwd_remask_f32(const float* dz, const float* x, const float* fcoef, int rows, int C, const float* gamma, const float* save_mean,
                                    const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    TF_REQUIRE(dz && x && fcoef && gamma && save_mean && save_invstd && dx && ws && rows > 0 && C > 0, "tf_bn_bwd_remask_f32: bad arguments");
    float* coef = ws + kWsFloats / 2;
    const bool v4 = (C % 4 == 0) && aligned16(dz) && aligned16(x) && aligned16(dx) && aligned16(fcoef);
    RedPlan p = plan_reduce(rows, C, 1, 2, v4);
    BnBwdRemaskF<4> f4{dz, x, fcoef, save_mean, save_invstd, C};
    BnBwdRemaskF<1> f1{dz, x, fcoef, save_mean, save_invstd, C};
    launch_reduce<2>(p, f4, f1, rows, C, 1, ws, stream);
    TF_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), stream, (const float*)ws, gamma, save_mean, save_invstd, dgamma, dbeta, coef, C, p.nchunks,
              (float)rows);
    const long n = (long)rows * C;
    if (v4) TF_LAUNCH(bn_bwd_apply_remask_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, dz, x, fcoef, (const float*)coef, dx, n / 4, C);
    else TF_LAUNCH(bn_bwd_apply_remask_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, dz, x, fcoef, (const float*)coef, dx, n, C);
    return launch_status("tf_bn_bwd_remask_f32");
}

// the same with the SE scale's backward folded in (tf_se_scale_bwd_x_f32 + tf_bn_bwd_remask_f32 in one reduction + one apply pass):
// dz = dy * sigmoid(gate[b][c]) + dmean[b][c] / HW is recomputed by both passes, never written
extern "C" int tf_bn_bwd_remask_se_f32(const float* dy, const float* gate, const float* dmean, int B, int HW, int C, const float* x, const float* fcoef,
                                       const float* gamma, const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* ws,
                                       void* stream) {
    TF_REQUIRE(dy && gate && dmean && x && fcoef && gamma && save_mean && save_invstd && dx && ws && B > 0 && HW > 0 && C > 0, "tf_bn_bwd_remask_se_f32: bad arguments");
    const int rows = B * HW;
    float* coef = ws + kWsFloats / 2;
    const bool v4 = (C % 4 == 0) && aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(fcoef) && aligned16(gate) && aligned16(dmean);
    RedPlan p = plan_reduce(rows, C, 1, 2, v4);
    const float inv_hw = 1.f / (float)HW;
    SeBnBwdRemaskF<4> f4{dy, gate, dmean, x, fcoef, save_mean, save_invstd, C, HW, inv_hw};
    SeBnBwdRemaskF<1> f1{dy, gate, dmean, x, fcoef, save_mean, save_invstd, C, HW, inv_hw};
    launch_reduce<2>(p, f4, f1, rows, C, 1, ws, stream);
    TF_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), stream, (const float*)ws, gamma, save_mean, save_invstd, dgamma, dbeta, coef, C, p.nchunks,
              (float)rows);
    const long n = (long)rows * C;
    if (v4) TF_LAUNCH(bn_bwd_apply_remask_se_kernel<4>, dim3(ew_blocks(n / 4)), dim3(256), stream, dy, gate, dmean, x, fcoef, (const float*)coef, dx, n / 4, C, (long)HW * C / 4, inv_hw);
    else TF_LAUNCH(bn_bwd_apply_remask_se_kernel<1>, dim3(ew_blocks(n)), dim3(256), stream, dy, gate, dmean, x, fcoef, (const float*)coef, dx, n, C, (long)HW * C, inv_hw);
    return launch_status("tf_bn_bwd_remask_se_f32");
}

// out[c] (+)= sum over rows of a[r][c] * b[r][c]  (ConvNeXt layer-scale gradient d gamma = sum dy * branch output, transfuser.py:395 via timm)
extern "C" int tf_colsum_mul_f32(const float* a, const float* b, int rows, int C, float* out, int accumulate, float* ws, void* stream) {
    TF_REQUIRE(a && b && out && ws && rows > 0 && C > 0, "tf_colsum_mul_f32: bad arguments");
    RedPlan p = plan_reduce(rows, C, 1, 1, aligned16(a) && aligned16(b));
    MulF<4> f4{a, b, C};
    MulF<1> f1{a, b, C};
    launch_reduce<1>(p, f4, f1, rows, C, 1, ws, stream);
    TF_LAUNCH(colsum_finalize_kernel, dim3(cdiv((long)C, 4)), dim3(256), stream, (const float*)ws, out, C, p.nchunks, 1, 1.f, accumulate);
    return launch_status("tf_colsum_mul_f32");
}

// out_e[c] += sum over the rows of x_e[r][c] for n <= 8 matrices with the SAME row count (ld_e = row stride in floats, C_e % 4 == 0, 16-byte
// aligned bases and strides): one single-pass launch - the bias gradients of a transformer Block (nn.Linear biases, transfuser.py:500-541).
extern "C" int tf_colsum_multi_f32(int n, const float* const* xs, const int* Cs, const long* lds, float* const* outs, int rows, void* stream) {
    TF_REQUIRE(n >= 1 && n <= kMultiMax && xs && Cs && lds && outs && rows > 0, "tf_colsum_multi_f32: 1 <= n <= %d matrices with rows > 0", kMultiMax);
    MultiSum d;
    d.n = n;
    int strips = 0;
    for (int i = 0; i < kMultiMax; ++i) {
        const int j = i < n ? i : n - 1;
        d.x[i] = xs[j]; d.out[i] = outs[j]; d.C[i] = Cs[j]; d.ld[i] = lds[j];
        d.strip0[i] = strips;
        if (i < n) {
            TF_REQUIRE(xs[i] && outs[i] && Cs[i] > 0 && Cs[i] % 4 == 0 && lds[i] % 4 == 0 && lds[i] >= Cs[i] && aligned16(xs[i]),
                       "tf_colsum_multi_f32: entry %d needs C %% 4 == 0, ld %% 4 == 0, ld >= C and a 16-byte aligned base", i);
            strips += cdiv(Cs[i], 32);
        }
    }
    d.strip0[kMultiMax] = strips;
    TF_LAUNCH(colsum_multi_kernel, dim3(strips), dim3(256), stream, d, rows);
    return launch_status("tf_colsum_multi_f32");
}

// ---- the four producers of a bottleneck's 1x1-convolution operands with their 16-bit copies (tile16_kernel above).  y16 (rows x C, contiguous) and / or
// y16t (C x rows8, row stride ldyt % 8 == 0, rows zero-padded to a multiple of 8) as tf_cast16_f32 writes them; dtype 1 = bf16, 2 = IEEE half.
// Every tensor 16-byte aligned, C % 4 == 0.  The fp32 output is optional (NULL) wherever only the 16-bit GEMMs read the result.
extern "C" int tf_bn_apply16_f32(const float* x, const float* coef, const float* res, int relu, float* y32, int rows, int C, void* y16, void* y16t, int ldyt,
                                 int dtype, void* stream) {
    TF_REQUIRE(x && coef && tile16_ok(rows, C, y16, y16t, ldyt, dtype) && aligned16(x) && aligned16(coef) && (!res || aligned16(res)) && (!y32 || aligned16(y32)),
               "tf_bn_apply16_f32: needs C %% 4 == 0, 16-byte aligned tensors, dtype 1 / 2, ldyt %% 8 == 0 and >= rows rounded up to 8");
    launch_tile16(BnApply16F{x, coef, res, y32, C, relu}, rows, C, y16, y16t, ldyt, dtype, stream);
    return launch_status("tf_bn_apply16_f32");
}
extern "C" int tf_se_scale_bn16_f32(const float* x, const float* coef, const float* gate, int B, int HW, int C, void* y16, void* y16t, int ldyt, int dtype,
                                    void* stream) {
    TF_REQUIRE(x && coef && gate && B > 0 && HW > 0 && tile16_ok(B * HW, C, y16, y16t, ldyt, dtype) && aligned16(x) && aligned16(coef) && aligned16(gate),
               "tf_se_scale_bn16_f32: needs C %% 4 == 0, 16-byte aligned tensors, dtype 1 / 2, ldyt %% 8 == 0 and >= rows rounded up to 8");
    launch_tile16(SeScaleBn16F{x, coef, gate, C, HW}, B * HW, C, y16, y16t, ldyt, dtype, stream);
    return launch_status("tf_se_scale_bn16_f32");
}
// tf_bn_bwd_f32 (chunk partials + finalize, no atomics) whose apply pass writes dx as 16-bit copies (+ fp32 dx32 when not NULL, + dres)
extern "C" int tf_bn_bwd16_f32(const float* dz, const float* z, const float* x, int rows, int C, const float* gamma, const float* save_mean,
                               const float* save_invstd, float* dx32, float* dres, float* dgamma, float* dbeta, float* ws, void* dx16, void* dx16t, int ldyt,
                               int dtype, void* stream) {
    TF_REQUIRE(dz && x && gamma && save_mean && save_invstd && ws && tile16_ok(rows, C, dx16, dx16t, ldyt, dtype) && aligned16(dz) && aligned16(x) &&
               (!z || aligned16(z)) && (!dres || aligned16(dres)) && (!dx32 || aligned16(dx32)),
               "tf_bn_bwd16_f32: needs C %% 4 == 0, 16-byte aligned tensors, dtype 1 / 2, ldyt %% 8 == 0 and >= rows rounded up to 8");
    float* coef = ws + kWsFloats / 2;
    RedPlan p = plan_reduce(rows, C, 1, 2, true);
    BnBwdF<4> f4{dz, z, x, save_mean, save_invstd, C};
    BnBwdF<1> f1{dz, z, x, save_mean, save_invstd, C};
    launch_reduce<2>(p, f4, f1, rows, C, 1, ws, stream);
    TF_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), stream, (const float*)ws, gamma, save_mean, save_invstd, dgamma, dbeta, coef, C, p.nchunks,
              (float)rows);
    launch_tile16(BnBwdApply16F{dz, z, x, (const float*)coef, dx32, dres, C}, rows, C, dx16, dx16t, ldyt, dtype, stream);
    return launch_status("tf_bn_bwd16_f32");
}
extern "C" int tf_bn_bwd_remask16_f32(const float* dz, const float* x, const float* fcoef, int rows, int C, const float* gamma, const float* save_mean,
                                      const float* save_invstd, float* dx32, float* dgamma, float* dbeta, float* ws, void* dx16, void* dx16t, int ldyt, int dtype,
                                      void* stream) {
    TF_REQUIRE(dz && x && fcoef && gamma && save_mean && save_invstd && ws && tile16_ok(rows, C, dx16, dx16t, ldyt, dtype) && aligned16(dz) && aligned16(x) &&
               aligned16(fcoef) && (!dx32 || aligned16(dx32)),
               "tf_bn_bwd_remask16_f32: needs C %% 4 == 0, 16-byte aligned tensors, dtype 1 / 2, ldyt %% 8 == 0 and >= rows rounded up to 8");
    float* coef = ws + kWsFloats / 2;
    RedPlan p = plan_reduce(rows, C, 1, 2, true);
    BnBwdRemaskF<4> f4{dz, x, fcoef, save_mean, save_invstd, C};
    BnBwdRemaskF<1> f1{dz, x, fcoef, save_mean, save_invstd, C};
    launch_reduce<2>(p, f4, f1, rows, C, 1, ws, stream);
    TF_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), stream, (const float*)ws, gamma, save_mean, save_invstd, dgamma, dbeta, coef, C, p.nchunks,
              (float)rows);
    launch_tile16(BnBwdRemask16F{dz, x, fcoef, (const float*)coef, dx32, C}, rows, C, dx16, dx16t, ldyt, dtype, stream);
    return launch_status("tf_bn_bwd_remask16_f32");
}

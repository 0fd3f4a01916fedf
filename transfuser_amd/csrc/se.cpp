// Squeeze-excite excitation MLP (timm SEModule fc1 -> ReLU -> fc2; 42 blocks x 2 trunks per step), fused.
//
// M = batch rows (<= 16): as separate "small-M" linears the excitation costs ~17 launches per block in forward+backward
// (2 FCs, ReLU mask, 2 bias sums, 2 dgrads with init, 2 wgrads ...), each a few microseconds of mostly latency.
// Here: ONE forward kernel and a 3-launch backward (zero scratch, fc2 pass, fc1 pass).  The weights (<= 2.3 MB) are streamed with
// 16-byte loads; all batch rows live in LDS / registers.
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int SE_MAXB = 16;
constexpr int SE_SPLIT = 8;        // forward: blocks per sample (each recomputes fc1 - cheap - and owns 1/8 of the fc2 outputs)
constexpr int SE_U = 4;            // scalar path: independent dot products in flight per wave
constexpr int SE_MAXC = 3072, SE_MAXR = 1024;

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// U rows of a row-major (rows x 4 n4) matrix against a vector in LDS, for one wave: every weight load of the call is issued before the first
// multiply (U * NK 16-byte loads in flight per lane; the run-time-bounded k loop this replaces waited for each trip's loads - a chain of
// ~1 us round trips, 9 per block at C = 576).  NK = float4 per lane and row; lanes past n4 load a clamped address and multiply by 0.
template <int U, int NK>
__device__ __forceinline__ void se_dots(const float* const* w, const float* __restrict__ vec, int n4, int lane, float* acc) {
    float4 wv[U][NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int k4 = lane + 64 * i, kc = k4 < n4 ? k4 : n4 - 1;
#pragma unroll
        for (int u = 0; u < U; ++u) wv[u][i] = reinterpret_cast<const float4*>(w[u])[kc];
    }
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int k4 = lane + 64 * i;
        float4 sv = reinterpret_cast<const float4*>(vec)[k4 < n4 ? k4 : n4 - 1];
        if (k4 >= n4) sv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += dot4(wv[u][i], sv);
    }
}

// out[r] = f(bias[r] + W[r][:] . vec) for rows r0 .. r1 of W (row length 4 n4), the block's waves striding over groups of U rows
template <int U, int NK, class Store>
__device__ __forceinline__ void se_rows(const float* __restrict__ W, const float* __restrict__ vec, int n4, int r0, int r1, int wave, int nw, int lane, Store store) {
    for (int rb = r0 + wave * U; rb < r1; rb += nw * U) {
        float acc[U];
        const float* w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[u] = 0.f;
            w[u] = W + (long)((rb + u < r1) ? rb + u : r1 - 1) * (4 * n4);
        }
        se_dots<U, NK>(w, vec, n4, lane, acc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float a = wave_sum(acc[u]);
            if (lane == 0 && rb + u < r1) store(rb + u, a);
        }
    }
}
// float4 per lane and row -> (rows in flight, NK) with <= 16 loads per lane outstanding
template <class Store>
__device__ __forceinline__ void se_rows_v4(const float* __restrict__ W, const float* __restrict__ vec, int n4, int r0, int r1, int wave, int nw, int lane, Store store) {
    const int nk = (n4 + 63) >> 6;
    if (nk <= 1) se_rows<8, 1>(W, vec, n4, r0, r1, wave, nw, lane, store);
    else if (nk <= 2) se_rows<8, 2>(W, vec, n4, r0, r1, wave, nw, lane, store);
    else if (nk <= 3) se_rows<4, 3>(W, vec, n4, r0, r1, wave, nw, lane, store);
    else if (nk <= 4) se_rows<4, 4>(W, vec, n4, r0, r1, wave, nw, lane, store);
    else if (nk <= 6) se_rows<2, 6>(W, vec, n4, r0, r1, wave, nw, lane, store);
    else if (nk <= 8) se_rows<2, 8>(W, vec, n4, r0, r1, wave, nw, lane, store);
    else se_rows<1, 12>(W, vec, n4, r0, r1, wave, nw, lane, store);          // up to 768 float4 (the host requires C <= 3072)
}

// g1[b][j] = relu(b1[j] + sum_k W1[j][k] s[b][k]);   gate[b][n] = b2[n] + sum_j W2[n][j] g1[b][j]
// nch > 0: ``s`` holds the squeeze as nch <= 8 partial column sums per sample, [b][chunk][C] (tf_colsum_bnrelu_parts_f32): they are added up here
// (x scale = 1 / HW) instead of by a finalize launch, and block (b, 0) writes the squeezed vector to s_out for the backward.
template <bool V4>
__global__ void __launch_bounds__(1024) se_excite_fwd_kernel(const float* __restrict__ s, const float* __restrict__ W1, const float* __restrict__ b1,
                                                             const float* __restrict__ W2, const float* __restrict__ b2, int C, int Cr,
                                                             float* __restrict__ g1, float* __restrict__ gate, float* __restrict__ zs, int nch = 0,
                                                             float scale = 1.f, float* __restrict__ s_out = nullptr) {
    __shared__ __attribute__((aligned(16))) float ss[SE_MAXC];
    __shared__ __attribute__((aligned(16))) float hh[SE_MAXR];
    const int b = blockIdx.x, part = blockIdx.y;
    const int tid = threadIdx.x, wave = wave_uniform(tid >> 6), lane = tid & 63, nw = blockDim.x >> 6;      // wave in an SGPR: the row pointers stay scalar
    if (nch > 0) {
        for (int k = tid; k < C; k += blockDim.x) {
            const float* p = s + (long)b * nch * C + k;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(long)(j < nch ? j : nch - 1) * C];          // all chunk loads in flight together
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += j < nch ? v[j] : 0.f;
            for (int j = 8; j < nch; ++j) t += p[(long)j * C];                                 // (the producer caps nch at 8)
            t *= scale;
            ss[k] = t;
            if (part == 0) s_out[(long)b * C + k] = t;
        }
    } else {
        for (int k = tid; k < C; k += blockDim.x) ss[k] = s[(long)b * C + k];
    }
    __syncthreads();
    auto store1 = [&](int j, float a) {
        const float v = fmaxf(a + (b1 ? b1[j] : 0.f), 0.f);
        hh[j] = v;
        if (part == 0) {
            g1[(long)b * Cr + j] = v;
            if (zs) zs[(long)b * Cr + j] = 0.f;      // the backward's (B, Cr) atomic accumulator, cleared here instead of by its own launch
        }
    };
    const int per = (C + SE_SPLIT - 1) / SE_SPLIT;
    const int n0 = part * per, n1 = (n0 + per < C) ? n0 + per : C;
    auto store2 = [&](int n, float a) { gate[(long)b * C + n] = a + (b2 ? b2[n] : 0.f); };
    if (V4) {
        se_rows_v4(W1, ss, C >> 2, 0, Cr, wave, nw, lane, store1);
        __syncthreads();
        if (n0 < n1) se_rows_v4(W2, hh, Cr >> 2, n0, n1, wave, nw, lane, store2);
    } else {
        for (int j0 = wave * SE_U; j0 < Cr; j0 += nw * SE_U) {
            float acc[SE_U];
            const float* w[SE_U];
#pragma unroll
            for (int u = 0; u < SE_U; ++u) {
                acc[u] = 0.f;
                w[u] = W1 + (long)((j0 + u < Cr) ? j0 + u : Cr - 1) * C;
            }
            for (int k = lane; k < C; k += 64) {
#pragma unroll
                for (int u = 0; u < SE_U; ++u) acc[u] += w[u][k] * ss[k];
            }
#pragma unroll
            for (int u = 0; u < SE_U; ++u) {
                const float a = wave_sum(acc[u]);
                if (lane == 0 && j0 + u < Cr) store1(j0 + u, a);
            }
        }
        __syncthreads();
        for (int nb = n0 + wave * SE_U; nb < n1; nb += nw * SE_U) {
            float acc[SE_U];
            const float* w[SE_U];
#pragma unroll
            for (int u = 0; u < SE_U; ++u) {
                acc[u] = 0.f;
                w[u] = W2 + (long)((nb + u < n1) ? nb + u : n1 - 1) * Cr;
            }
            for (int j = lane; j < Cr; j += 64) {
#pragma unroll
                for (int u = 0; u < SE_U; ++u) acc[u] += w[u][j] * hh[j];
            }
#pragma unroll
            for (int u = 0; u < SE_U; ++u) {
                const float a = wave_sum(acc[u]);
                if (lane == 0 && nb + u < n1) store2(nb + u, a);
            }
        }
    }
}

__global__ void __launch_bounds__(256) se_zero_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// fc2 pass, one block per SE_ROWS rows n of W2:  dW2[n][:] += dgate[:,n]^T g1,  db2[n] += sum_b dgate[b][n],
// and this slice's contribution to dg1[b][j] = sum_n dgate[b][n] W2[n][j] (atomics into the zeroed scratch).
constexpr int SE_ROWS = 8;
__global__ void __launch_bounds__(256) se_excite_bwd2_kernel(const float* __restrict__ dgate, const float* __restrict__ g1, const float* __restrict__ W2,
                                                             int B, int C, int Cr, float* __restrict__ dW2, float* __restrict__ db2,
                                                             float* __restrict__ dg1, float* __restrict__ ds_zero, int nch = 0,
                                                             const float* __restrict__ gate = nullptr) {
    __shared__ float dgs[SE_MAXB][SE_ROWS];
    // ds (B x C) is ACCUMULATED by the j-sliced fc1 pass that follows: cleared here, one slice per block
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * C; i += (long)gridDim.x * 256) ds_zero[i] = 0.f;
    __shared__ float g1s[SE_MAXB * SE_MAXR / 2];     // B * Cr <= 8192 floats (32 KB): checked on the host
    const int n0 = blockIdx.x * SE_ROWS, tid = threadIdx.x;
    const int rows = (C - n0 < SE_ROWS) ? C - n0 : SE_ROWS;
    for (int i = tid; i < B * SE_ROWS; i += 256) {
        const int b = i / SE_ROWS, r = i % SE_ROWS;
        float v = 0.f;
        if (r < rows) {
            if (nch > 0) {     // dgate arrives as nch partial sums per sample of sum_hw dy z (tf_se_gate_grad_parts_f32): finish it here, x s (1 - s)
                const float* p = dgate + (long)b * nch * C + n0 + r;
                float v1 = 0.f, v2 = 0.f, v3 = 0.f;
                int j = 0;
                for (; j + 3 < nch; j += 4) { v += p[(long)j * C]; v1 += p[(long)(j + 1) * C]; v2 += p[(long)(j + 2) * C]; v3 += p[(long)(j + 3) * C]; }
                for (; j < nch; ++j) v += p[(long)j * C];
                v = (v + v1) + (v2 + v3);
                const float sg = 1.f / (1.f + expf(-gate[(long)b * C + n0 + r]));
                v *= sg * (1.f - sg);
            } else {
                v = dgate[(long)b * C + n0 + r];
            }
        }
        dgs[b][r] = v;
    }
    for (int i = tid; i < B * Cr; i += 256) g1s[i] = g1[i];
    __syncthreads();
    for (int i = tid; i < rows * Cr; i += 256) {
        const int r = i / Cr, j = i - r * Cr;
        float v = 0.f;
        for (int b = 0; b < B; ++b) v += dgs[b][r] * g1s[b * Cr + j];
        dW2[(long)(n0 + r) * Cr + j] += v;
    }
    if (tid < rows && db2) {
        float v = 0.f;
        for (int b = 0; b < B; ++b) v += dgs[b][tid];
        db2[n0 + tid] += v;
    }
    for (int j = tid; j < Cr; j += 256) {
        float acc[SE_MAXB];
#pragma unroll
        for (int b = 0; b < SE_MAXB; ++b) acc[b] = 0.f;
        // the SE_ROWS weight loads of a column are issued together (clamped row index, rows beyond the slice multiply by dgs = 0): the
        // run-time-bounded loop made them a chain of ~1 us round trips
        float wv[SE_ROWS];
#pragma unroll
        for (int r = 0; r < SE_ROWS; ++r) wv[r] = W2[(long)(n0 + (r < rows ? r : rows - 1)) * Cr + j];
#pragma unroll
        for (int r = 0; r < SE_ROWS; ++r) {
            const float w = r < rows ? wv[r] : 0.f;
#pragma unroll
            for (int b = 0; b < SE_MAXB; ++b)
                if (b < B) acc[b] += dgs[b][r] * w;
        }
#pragma unroll
        for (int b = 0; b < SE_MAXB; ++b)
            if (b < B) atomicAdd(dg1 + b * Cr + j, acc[b]);
    }
}

// fc1 pass, one block per (32 columns k of W1, slice of the Cr rows j):  dg = dg1 * (g1 > 0);  dW1[j][k] += sum_b dg[b][j] s[b][k];
// db1[j] += sum_b dg[b][j] (column block 0);  ds[b][k] += sum_{j in slice} dg[b][j] W1[j][k] (fp32 atomics into the buffer the fc2 pass
// cleared: SE_JS j-slices give C / 32 x SE_JS blocks instead of 18-47 - the kernel was a 23 us latency chain on a 10 % filled GPU).
constexpr int SE_JS = 6;
__global__ void __launch_bounds__(256) se_excite_bwd1_kernel(const float* __restrict__ dg1, const float* __restrict__ g1, const float* __restrict__ s,
                                                             const float* __restrict__ W1, int B, int C, int Cr, float* __restrict__ dW1,
                                                             float* __restrict__ db1, float* __restrict__ ds) {
    __shared__ float dgm[SE_MAXB * SE_MAXR / 2];
    __shared__ float ssl[SE_MAXB][32];
    __shared__ float red[8][SE_MAXB][32];
    const int tid = threadIdx.x, kx = tid & 31, jg = tid >> 5;
    const int k = blockIdx.x * 32 + kx;
    const int jper = (Cr + SE_JS - 1) / SE_JS, j0 = blockIdx.y * jper, j1 = (j0 + jper < Cr) ? j0 + jper : Cr;
    for (int i = tid; i < B * Cr; i += 256) dgm[i] = g1[i] > 0.f ? dg1[i] : 0.f;
    for (int i = tid; i < B * 32; i += 256) {
        const int b = i >> 5, c = blockIdx.x * 32 + (i & 31);
        ssl[b][i & 31] = c < C ? s[(long)b * C + c] : 0.f;
    }
    __syncthreads();
    float acc[SE_MAXB];
#pragma unroll
    for (int b = 0; b < SE_MAXB; ++b) acc[b] = 0.f;
    if (k < C)
        for (int jb = j0 + jg; jb < j1; jb += 32) {       // four rows per trip: their W1 / dW1 loads are in flight together
            float w[4], old[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = jb + 8 * u < j1 ? jb + 8 * u : j1 - 1;
                w[u] = W1[(long)j * C + k];
                old[u] = dW1[(long)j * C + k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = jb + 8 * u;
                if (j < j1) {
                    float dw = 0.f;
#pragma unroll
                    for (int b = 0; b < SE_MAXB; ++b)
                        if (b < B) {
                            const float d = dgm[b * Cr + j];
                            acc[b] += d * w[u];
                            dw += d * ssl[b][kx];
                        }
                    dW1[(long)j * C + k] = old[u] + dw;
                }
            }
        }
#pragma unroll
    for (int b = 0; b < SE_MAXB; ++b) red[jg][b][kx] = acc[b];
    __syncthreads();
    for (int b = jg; b < B; b += 8)
        if (k < C) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += red[g][b][kx];
            atomicAdd(ds + (long)b * C + k, v);
        }
    if (blockIdx.x == 0 && db1)
        for (int j = j0 + tid; j < j1; j += 256) {
            float v = 0.f;
            for (int b = 0; b < B; ++b) v += dgm[b * Cr + j];
            db1[j] += v;
        }
}

}  // namespace

extern "C" int tf_se_excite_fwd_f32(const float* s, const float* W1, const float* b1, const float* W2, const float* b2, int B, int C, int Cr, float* g1,
                                    float* gate, float* bwd_scratch, void* stream) {
    TF_REQUIRE(s && W1 && W2 && g1 && gate && B > 0 && C > 0 && Cr > 0 && C <= SE_MAXC && Cr <= SE_MAXR,
               "tf_se_excite_fwd_f32: bad arguments (C <= 3072, Cr <= 1024)");
    const bool v4 = C % 4 == 0 && Cr % 4 == 0 && aligned16(W1) && aligned16(W2);
    if (v4) TF_LAUNCH(se_excite_fwd_kernel<true>, dim3(B, SE_SPLIT), dim3(1024), stream, s, W1, b1, W2, b2, C, Cr, g1, gate, bwd_scratch);
    else TF_LAUNCH(se_excite_fwd_kernel<false>, dim3(B, SE_SPLIT), dim3(1024), stream, s, W1, b1, W2, b2, C, Cr, g1, gate, bwd_scratch);
    return launch_status("tf_se_excite_fwd_f32");
}

extern "C" int tf_se_excite_bwd_f32(const float* dgate, const float* s, const float* g1, const float* W1, const float* W2, int B, int C, int Cr,
                                    float* dW1, float* db1, float* dW2, float* db2, float* ds, float* scratch, int scratch_is_zero, void* stream) {
    TF_REQUIRE(dgate && s && g1 && W1 && W2 && dW1 && dW2 && ds && scratch && B > 0 && B <= SE_MAXB && C > 0 && Cr > 0 &&
                   (long)B * Cr <= SE_MAXB * SE_MAXR / 2, "tf_se_excite_bwd_f32: bad arguments (B <= 16, B*Cr <= 8192)");
    if (!scratch_is_zero) TF_LAUNCH(se_zero_kernel, dim3(cdiv((long)B * Cr, 256)), dim3(256), stream, scratch, B * Cr);
    TF_LAUNCH(se_excite_bwd2_kernel, dim3(cdiv(C, SE_ROWS)), dim3(256), stream, dgate, g1, W2, B, C, Cr, dW2, db2, scratch, ds);
    TF_LAUNCH(se_excite_bwd1_kernel, dim3(cdiv(C, 32), SE_JS), dim3(256), stream, (const float*)scratch, g1, s, W1, B, C, Cr, dW1, db1, ds);
    return launch_status("tf_se_excite_bwd_f32");
}

// The same two passes fed by PARTIAL sums (the BatchNorm-apply-in-the-consumer path of YBlockFn): the squeeze / the gate gradient arrive as
// [b][chunk][C] chunk sums in the reduction workspace and are finished inside the kernels (no finalize launches).
extern "C" int tf_se_excite_fwd_parts_f32(const float* parts, int nchunks, float scale, const float* W1, const float* b1, const float* W2, const float* b2, int B,
                                          int C, int Cr, float* s_out, float* g1, float* gate, float* bwd_scratch, void* stream) {
    TF_REQUIRE(parts && nchunks > 0 && W1 && W2 && s_out && g1 && gate && B > 0 && C > 0 && Cr > 0 && C <= SE_MAXC && Cr <= SE_MAXR,
               "tf_se_excite_fwd_parts_f32: bad arguments (C <= 3072, Cr <= 1024)");
    const bool v4 = C % 4 == 0 && Cr % 4 == 0 && aligned16(W1) && aligned16(W2);
    if (v4) TF_LAUNCH(se_excite_fwd_kernel<true>, dim3(B, SE_SPLIT), dim3(1024), stream, parts, W1, b1, W2, b2, C, Cr, g1, gate, bwd_scratch, nchunks, scale, s_out);
    else TF_LAUNCH(se_excite_fwd_kernel<false>, dim3(B, SE_SPLIT), dim3(1024), stream, parts, W1, b1, W2, b2, C, Cr, g1, gate, bwd_scratch, nchunks, scale, s_out);
    return launch_status("tf_se_excite_fwd_parts_f32");
}

extern "C" int tf_se_excite_bwd_parts_f32(const float* parts, int nchunks, const float* gate, const float* s, const float* g1, const float* W1, const float* W2,
                                          int B, int C, int Cr, float* dW1, float* db1, float* dW2, float* db2, float* ds, float* scratch, int scratch_is_zero,
                                          void* stream) {
    TF_REQUIRE(parts && nchunks > 0 && gate && s && g1 && W1 && W2 && dW1 && dW2 && ds && scratch && B > 0 && B <= SE_MAXB && C > 0 && Cr > 0 &&
                   (long)B * Cr <= SE_MAXB * SE_MAXR / 2, "tf_se_excite_bwd_parts_f32: bad arguments (B <= 16, B*Cr <= 8192)");
    if (!scratch_is_zero) TF_LAUNCH(se_zero_kernel, dim3(cdiv((long)B * Cr, 256)), dim3(256), stream, scratch, B * Cr);
    TF_LAUNCH(se_excite_bwd2_kernel, dim3(cdiv(C, SE_ROWS)), dim3(256), stream, parts, g1, W2, B, C, Cr, dW2, db2, scratch, ds, nchunks, gate);
    TF_LAUNCH(se_excite_bwd1_kernel, dim3(cdiv(C, 32), SE_JS), dim3(256), stream, (const float*)scratch, g1, s, W1, B, C, Cr, dW1, db1, ds);
    return launch_status("tf_se_excite_bwd_parts_f32");
}

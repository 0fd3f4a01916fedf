// Row-wise normalisation kernels (HBM-bound): LayerNorm fwd/bwd, attention softmax fwd/bwd.
// One wave64 per row, float4 traffic, reductions by cross-lane shuffles only.
#include "tf_common.h"
#include <stdlib.h>
#include "../../include/transfuser_hip.h"

using namespace tf;

// ------------------------------------------------------------------ LayerNorm
// transfuser.py:319,535-536 (nn.LayerNorm, eps 1e-5): y = (x - mean) * rstd * gamma + beta.
// mean / biased variance two-pass (second pass hits L1/L2), matches torch within fp32 roundoff.
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
                                                            int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = row < rows;  // whole wave shares the predicate
    const float* xr = x + (long)(live ? row : 0) * C;
    float s = 0.f;
    if (live)
        for (int c = lane; c < C; c += 64) s += xr[c];
    s = wave_sum(s);
    const float mean = s / (float)C;
    float q = 0.f;
    if (live)
        for (int c = lane; c < C; c += 64) { float d = xr[c] - mean; q += d * d; }
    q = wave_sum(q);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (!live) return;
    float* yr = y + (long)row * C;
    for (int c = lane; c < C; c += 64) yr[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Optional second output of the backward (round 5): out = nn.Dropout(dx) with the mask tf_dropout_f32 generates for (seed, site) over the contiguous
// (rows, C) tensor - the gradient that enters the NEXT residual branch of a transformer Block (x_mid = x + resid_drop(proj(.)), transfuser.py:543),
// written by the launch that finishes dx instead of a separate dropout launch.
struct LnDrop { float* out; const uint32_t* seed; uint32_t site, thresh; float keep_scale; };

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma
__global__ void __launch_bounds__(256) layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx, int rows, int C,
                                                               int accumulate, LnDrop dr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = row < rows;
    const long o = (long)(live ? row : 0) * C;
    const float m = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (live)
        for (int c = lane; c < C; c += 64) {
            float g = dy[o + c] * gamma[c];
            s1 += g;
            s2 += g * ((x[o + c] - m) * rs);
        }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    if (!live) return;
    for (int c = lane; c < C; c += 64) {
        float g = dy[o + c] * gamma[c];
        float xh = (x[o + c] - m) * rs;
        float v = rs * (g - s1 - xh * s2);
        if (accumulate) v += dx[o + c];
        dx[o + c] = v;
        if (dr.out) dr.out[o + c] = dropout_keep(*dr.seed, dr.site, (uint32_t)(o + c), dr.thresh) ? v * dr.keep_scale : 0.f;
    }
}

// ---- register-resident rows: C % 4 == 0, C <= 2048.  A lane holds its NV float4 of the row (unconditional loads from clamped indices, all in
// flight together), both passes of the statistics run on registers and the row is read from memory ONCE (the scalar kernels above walk the row
// three times with one 4-byte load per trip: 16.6 us for [1740 x 1512] = 1.3 TB/s).  Same two-pass arithmetic, lane-interleaved summation order.
template <int NV>
__global__ void __launch_bounds__(256) layernorm_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int C,
                                                               float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = row < rows;
    const int cv = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)(live ? row : 0) * C);
    float4 v[NV];
    bool in[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = lane + 64 * u;
        in[u] = live && i < cv;
        v[u] = xr[i < cv ? i : cv - 1];
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) s += in[u] ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.f;
    s = wave_sum(s);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const float a = v[u].x - mean, b = v[u].y - mean, c = v[u].z - mean, d = v[u].w - mean;
        q += in[u] ? (a * a + b * b) + (c * c + d * d) : 0.f;
    }
    q = wave_sum(q);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (!live) return;
    float4* yr = reinterpret_cast<float4*>(y + (long)row * C);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = lane + 64 * u;
        if (in[u]) {
            const float4 g = g4[i], b = b4[i];
            yr[i] = make_float4((v[u].x - mean) * rstd * g.x + b.x, (v[u].y - mean) * rstd * g.y + b.y, (v[u].z - mean) * rstd * g.z + b.z,
                                (v[u].w - mean) * rstd * g.w + b.w);
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}
template <int NV>
__global__ void __launch_bounds__(256) layernorm_bwd_dx_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, int rows,
                                                                  int C, int accumulate, LnDrop dr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = row < rows;
    const int cv = C >> 2;
    const long o = (long)(live ? row : 0) * C;
    const float4* d4 = reinterpret_cast<const float4*>(dy + o);
    const float4* x4 = reinterpret_cast<const float4*>(x + o);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    float4* o4 = reinterpret_cast<float4*>(dx + o);
    const float m = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
    float4 g[NV], xh[NV], old[NV];
    bool in[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = lane + 64 * u, ic = i < cv ? i : cv - 1;
        in[u] = live && i < cv;
        const float4 d = d4[ic], w = g4[ic], xv = x4[ic];
        if (accumulate) old[u] = o4[ic];
        g[u] = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
        xh[u] = make_float4((xv.x - m) * rs, (xv.y - m) * rs, (xv.z - m) * rs, (xv.w - m) * rs);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
        if (in[u]) {
            s1 += (g[u].x + g[u].y) + (g[u].z + g[u].w);
            s2 += (g[u].x * xh[u].x + g[u].y * xh[u].y) + (g[u].z * xh[u].z + g[u].w * xh[u].w);
        }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    if (!live) return;
#pragma unroll
    for (int u = 0; u < NV; ++u)
        if (in[u]) {
            float4 v = make_float4(rs * (g[u].x - s1 - xh[u].x * s2), rs * (g[u].y - s1 - xh[u].y * s2), rs * (g[u].z - s1 - xh[u].z * s2),
                                   rs * (g[u].w - s1 - xh[u].w * s2));
            if (accumulate) { v.x += old[u].x; v.y += old[u].y; v.z += old[u].z; v.w += old[u].w; }
            o4[lane + 64 * u] = v;
            if (dr.out) {
                const uint32_t sd = *dr.seed, e0 = (uint32_t)(o + 4 * (lane + 64 * u));
                reinterpret_cast<float4*>(dr.out + o)[lane + 64 * u] =
                    make_float4(dropout_keep(sd, dr.site, e0, dr.thresh) ? v.x * dr.keep_scale : 0.f, dropout_keep(sd, dr.site, e0 + 1, dr.thresh) ? v.y * dr.keep_scale : 0.f,
                                dropout_keep(sd, dr.site, e0 + 2, dr.thresh) ? v.z * dr.keep_scale : 0.f, dropout_keep(sd, dr.site, e0 + 3, dr.thresh) ? v.w * dr.keep_scale : 0.f);
            }
        }
}
// ---- LayerNorm forward whose ONLY outputs are the 16-bit operand copies of the packed-16 GEMM path (round 5; bf16 / fp16 storage modes): y16 [rows][C]
// and its transpose y16t [C][rows8] (rows zero-padded to a multiple of 8), exactly what tf_cast16_f32 writes from the fp32 result of
// tf_layernorm_fwd_f32 - which nothing reads in those modes (the Block's backward needs x, mean, rstd and the TRANSPOSED 16-bit copy).  8 rows per
// block (one wave each, the row in registers); the transposed copy goes through LDS as an [8][C] tile so that every column's 8 rows leave as one
// 16-byte store.  12 B / element (4 read + 4 written + 4 re-read by the cast + 2 x 2 written) become 8.
template <int NV>
__global__ void __launch_bounds__(512) layernorm_fwd16_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              uint16_t* __restrict__ y, long ldy, uint16_t* __restrict__ yt, long ldyt,
                                                              float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int C, float eps, int f16) {
    __shared__ __attribute__((aligned(16))) uint16_t ln_tile[8 * 256 * NV];      // [8][C], C <= 256 NV
    const int w = threadIdx.x >> 6, row = blockIdx.x * 8 + w, lane = threadIdx.x & 63;
    const bool live = row < rows;
    const int cv = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)(live ? row : 0) * C);
    float4 v[NV];
    bool in[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = lane + 64 * u;
        in[u] = live && i < cv;
        v[u] = xr[i < cv ? i : cv - 1];
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) s += in[u] ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.f;
    s = wave_sum(s);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const float a = v[u].x - mean, b = v[u].y - mean, c = v[u].z - mean, d = v[u].w - mean;
        q += in[u] ? (a * a + b * b) + (c * c + d * d) : 0.f;
    }
    q = wave_sum(q);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = lane + 64 * u;
        if (i < cv) {
            uint16_t h[4] = {0, 0, 0, 0};
            if (live) {
                const float4 g = g4[i], b = b4[i];
                h[0] = cvt16_bits((v[u].x - mean) * rstd * g.x + b.x, f16 != 0); h[1] = cvt16_bits((v[u].y - mean) * rstd * g.y + b.y, f16 != 0);
                h[2] = cvt16_bits((v[u].z - mean) * rstd * g.z + b.z, f16 != 0); h[3] = cvt16_bits((v[u].w - mean) * rstd * g.w + b.w, f16 != 0);
            }
            const float2 pk = make_float2(__uint_as_float((uint32_t)h[0] | ((uint32_t)h[1] << 16)), __uint_as_float((uint32_t)h[2] | ((uint32_t)h[3] << 16)));      // bit containers
            if (live && y) *reinterpret_cast<float2*>(y + (long)row * ldy + 4 * i) = pk;
            *reinterpret_cast<float2*>(ln_tile + (long)w * C + 4 * i) = pk;          // dead rows (the zero padding of the transposed copy) store zeros
        }
    }
    if (live && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    if (!yt) return;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 512) {
        uint32_t d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = (uint32_t)ln_tile[(2 * r) * C + c] | ((uint32_t)ln_tile[(2 * r + 1) * C + c] << 16);
        *reinterpret_cast<float4*>(yt + (long)c * ldyt + blockIdx.x * 8) = make_float4(__uint_as_float(d[0]), __uint_as_float(d[1]), __uint_as_float(d[2]), __uint_as_float(d[3]));
    }
}
static const bool g_ln_v4 = [] { const char* e = getenv("TF_LN_V4"); return e ? e[0] != '0' : true; }();
static inline int ln_nv(int C, const void* a, const void* b, const void* c) {
    if (!g_ln_v4 || C % 4 != 0 || C > 2048 || !aligned16(a) || !aligned16(b) || (c && !aligned16(c))) return 0;
    const int need = (C / 4 + 63) / 64;
    return need <= 1 ? 1 : need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : need <= 6 ? 6 : 8;
}

// dgamma[c] += sum_rows dy * xhat ; dbeta[c] += sum_rows dy.  Block = 64 columns x 4 row-lanes,
// each block reduces a chunk of rows, then one atomic per column per block.
__global__ void __launch_bounds__(256) layernorm_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C,
                                                               int rows_per_block) {
    __shared__ float sg[4][64], sb[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float ag = 0.f, ab = 0.f;
    if (c < C)
        for (int r = r0 + ry; r < r1; r += 16) {          // four rows per trip from clamped indices: 16 loads in flight instead of 4
            float d[4], xv[4], m[4], rs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + 4 * u < r1 ? r + 4 * u : r1 - 1;
                d[u] = dy[(long)rr * C + c]; xv[u] = x[(long)rr * C + c]; m[u] = mean[rr]; rs[u] = rstd[rr];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 4 * u < r1) {
                    ag += d[u] * ((xv[u] - m[u]) * rs[u]);
                    ab += d[u];
                }
        }
    sg[ry][threadIdx.x & 63] = ag;
    sb[ry][threadIdx.x & 63] = ab;
    __syncthreads();
    if (ry == 0 && c < C) {
        const int l = threadIdx.x;
        atomicAdd(&dgamma[c], (sg[0][l] + sg[1][l]) + (sg[2][l] + sg[3][l]));
        atomicAdd(&dbeta[c], (sb[0][l] + sb[1][l]) + (sb[2][l] + sb[3][l]));
    }
}

extern "C" int tf_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int rows,
                                    int C, float eps, void* stream) {
    TF_REQUIRE(x && gamma && beta && y && mean && rstd && rows >= 0 && C > 0, "tf_layernorm_fwd_f32: bad arguments");
    if (rows == 0) return 0;
    switch (ln_nv(C, x, y, gamma) && aligned16(beta) ? ln_nv(C, x, y, gamma) : 0) {
#define TF_LNF(NV_) case NV_: TF_LAUNCH(layernorm_fwd_v4_kernel<NV_>, dim3(cdiv(rows, 4)), dim3(256), stream, x, gamma, beta, y, mean, rstd, rows, C, eps); break
        TF_LNF(1); TF_LNF(2); TF_LNF(3); TF_LNF(4); TF_LNF(6); TF_LNF(8);
#undef TF_LNF
        default: TF_LAUNCH(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), stream, x, gamma, beta, y, mean, rstd, rows, C, eps);
    }
    return launch_status("tf_layernorm_fwd_f32");
}

extern "C" int tf_layernorm_fwd16_f32(const float* x, const float* gamma, const float* beta, void* y16, int ldy, void* y16t, int ldyt, float* mean, float* rstd,
                                      int rows, int C, float eps, int dtype, void* stream) {
    TF_REQUIRE(x && gamma && beta && mean && rstd && (y16 || y16t) && rows >= 0 && C > 0 && (dtype == 1 || dtype == 2), "tf_layernorm_fwd16_f32: bad arguments (dtype 1 = bf16, 2 = fp16)");
    TF_REQUIRE(C % 4 == 0 && C <= 2048 && aligned16(x) && aligned16(gamma) && aligned16(beta), "tf_layernorm_fwd16_f32: needs C %% 4 == 0, C <= 2048, 16-byte aligned x / gamma / beta");
    TF_REQUIRE(!y16 || (ldy >= C && ldy % 4 == 0 && ((uintptr_t)y16 & 7) == 0), "tf_layernorm_fwd16_f32: y16 needs ldy >= C, ldy %% 4 == 0, 8-byte alignment");
    TF_REQUIRE(!y16t || (ldyt >= ((rows + 7) & ~7) && ldyt % 8 == 0 && aligned16(y16t)), "tf_layernorm_fwd16_f32: the transposed copy needs ldyt %% 8 == 0, ldyt >= rows rounded up to 8, 16-byte aligned");
    if (rows == 0) return 0;
    const int need = (C / 4 + 63) / 64, nv = need <= 1 ? 1 : need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : need <= 6 ? 6 : 8;
    switch (nv) {
#define TF_LNF16(NV_) case NV_: TF_LAUNCH(layernorm_fwd16_kernel<NV_>, dim3(cdiv(rows, 8)), dim3(512), stream, x, gamma, beta, (uint16_t*)y16, (long)ldy, (uint16_t*)y16t, (long)ldyt, mean, rstd, rows, C, eps, dtype == 2 ? 1 : 0); break
        TF_LNF16(1); TF_LNF16(2); TF_LNF16(3); TF_LNF16(4); TF_LNF16(6); TF_LNF16(8);
#undef TF_LNF16
    }
    return launch_status("tf_layernorm_fwd16_f32");
}

static int layernorm_bwd_impl(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                              int dx_accumulate, float* dgamma, float* dbeta, int rows, int C, LnDrop dr, void* stream, const char* what) {
    if (rows == 0) return 0;
    switch (ln_nv(C, dy, x, gamma) && aligned16(dx) && (!dr.out || aligned16(dr.out)) ? ln_nv(C, dy, x, gamma) : 0) {
#define TF_LNB(NV_) case NV_: TF_LAUNCH(layernorm_bwd_dx_v4_kernel<NV_>, dim3(cdiv(rows, 4)), dim3(256), stream, dy, x, gamma, mean, rstd, dx, rows, C, dx_accumulate, dr); break
        TF_LNB(1); TF_LNB(2); TF_LNB(3); TF_LNB(4); TF_LNB(6); TF_LNB(8);
#undef TF_LNB
        default: TF_LAUNCH(layernorm_bwd_dx_kernel, dim3(cdiv(rows, 4)), dim3(256), stream, dy, x, gamma, mean, rstd, dx, rows, C, dx_accumulate, dr);
    }
    if (dgamma && dbeta) {
        const int rpb = 64;
        TF_LAUNCH(layernorm_bwd_dw_kernel, dim3(cdiv(C, 64), cdiv(rows, rpb)), dim3(256), stream, dy, x, mean, rstd, dgamma, dbeta, rows,
                  C, rpb);
    }
    return launch_status(what);
}

extern "C" int tf_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                                    int dx_accumulate, float* dgamma, float* dbeta, int rows, int C, void* stream) {
    TF_REQUIRE(dy && x && gamma && mean && rstd && dx && rows >= 0 && C > 0, "tf_layernorm_bwd_f32: bad arguments");
    return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, dx_accumulate, dgamma, dbeta, rows, C, LnDrop{nullptr, nullptr, 0u, 0u, 1.f}, stream, "tf_layernorm_bwd_f32");
}

extern "C" int tf_layernorm_bwd_drop_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                                         int dx_accumulate, float* dgamma, float* dbeta, int rows, int C, float* dropped, const uint32_t* seed_dev,
                                         uint32_t site, float p, void* stream) {
    TF_REQUIRE(dy && x && gamma && mean && rstd && dx && rows >= 0 && C > 0 && dropped && dropped != dx && seed_dev && p >= 0.f && p < 1.f && (long)rows * C < 4294967296L,
               "tf_layernorm_bwd_drop_f32: bad arguments (dropped must be a second buffer, 0 <= p < 1)");
    const LnDrop dr{dropped, seed_dev, site, (uint32_t)((double)p * 4294967296.0), 1.f / (1.f - p)};
    return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, dx_accumulate, dgamma, dbeta, rows, C, dr, stream, "tf_layernorm_bwd_drop_f32");
}

// ------------------------------------------------------------------ softmax (attention rows)
// transfuser.py:520-521: att = softmax(q k^T / sqrt(hs)); the 1/sqrt(hs) lives in the GEMM alpha.
// Row length n <= 64 * SM_MAXV (T = 174 here); row stride ld >= n.  In place.
constexpr int SM_MAXV = 8;

// DROP: additionally writes attn_drop(probabilities) to sd (transfuser.py:521; mask = tf_dropout_f32's for the flat index row * ld + column)
template <bool DROP>
__global__ void __launch_bounds__(256) softmax_fwd_kernel(float* __restrict__ s, int rows, int n, int ld, float* __restrict__ sd = nullptr,
                                                          const uint32_t* __restrict__ seed = nullptr, uint32_t site = 0, uint32_t thresh = 0,
                                                          float keep_scale = 1.f) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = row < rows;
    float* p = s + (long)(live ? row : 0) * ld;
    float v[SM_MAXV];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = (live && c < n) ? p[c] : -3.0e38f;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = (live && c < n) ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    if (!live) return;
    const float inv = 1.0f / sum;
    uint32_t sdv = 0;
    if constexpr (DROP) sdv = *seed;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < n) {
            const float pr = v[i] * inv;
            p[c] = pr;
            if constexpr (DROP) {
                const long o = (long)row * ld + c;
                sd[o] = dropout_keep(sdv, site, (uint32_t)o, thresh) ? pr * keep_scale : 0.f;
            }
        }
    }
}

// In: p = probabilities, dp = dL/dp.  Out (in place in dp): dL/ds = p * (dp - sum(dp * p)).
// DROP: dp is the gradient w.r.t. the DROPPED probabilities; the attn_drop backward (same mask) is applied on load
template <bool DROP>
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, int rows, int n, int ld,
                                                          const uint32_t* __restrict__ seed = nullptr, uint32_t site = 0, uint32_t thresh = 0,
                                                          float keep_scale = 1.f) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool live = row < rows;
    const long o = (long)(live ? row : 0) * ld;
    float y[SM_MAXV], g[SM_MAXV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = lane + i * 64;
        const bool ok = live && c < n;
        y[i] = ok ? p[o + c] : 0.f;
        g[i] = ok ? dp[o + c] : 0.f;
        if constexpr (DROP) g[i] = (ok && dropout_keep(*seed, site, (uint32_t)(o + c), thresh)) ? g[i] * keep_scale : 0.f;
        dot += y[i] * g[i];
    }
    dot = wave_sum(dot);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < n) dp[o + c] = y[i] * (g[i] - dot);
    }
}

extern "C" int tf_softmax_fwd_f32(float* s, int rows, int n, int ld, void* stream) {
    TF_REQUIRE(s && rows >= 0 && n > 0 && n <= 64 * SM_MAXV && ld >= n, "tf_softmax_fwd_f32: bad arguments (n=%d)", n);
    if (rows == 0) return 0;
    TF_LAUNCH(softmax_fwd_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), stream, s, rows, n, ld, (float*)nullptr, (const uint32_t*)nullptr, 0u, 0u, 1.f);
    return launch_status("tf_softmax_fwd_f32");
}

extern "C" int tf_softmax_bwd_f32(const float* p, float* dp, int rows, int n, int ld, void* stream) {
    TF_REQUIRE(p && dp && rows >= 0 && n > 0 && n <= 64 * SM_MAXV && ld >= n, "tf_softmax_bwd_f32: bad arguments (n=%d)", n);
    if (rows == 0) return 0;
    TF_LAUNCH(softmax_bwd_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), stream, p, dp, rows, n, ld, (const uint32_t*)nullptr, 0u, 0u, 1.f);
    return launch_status("tf_softmax_bwd_f32");
}

// softmax + attn_drop in one pass (transfuser.py:520-521): s <- probabilities (kept for the backward), sd <- dropped probabilities
extern "C" int tf_softmax_dropout_fwd_f32(float* s, float* sd, int rows, int n, int ld, const uint32_t* seed_dev, uint32_t site, float p, void* stream) {
    TF_REQUIRE(s && sd && seed_dev && rows >= 0 && n > 0 && n <= 64 * SM_MAXV && ld >= n && p >= 0.f && p < 1.f && (long)rows * ld < (1L << 32),
               "tf_softmax_dropout_fwd_f32: bad arguments (n=%d)", n);
    if (rows == 0) return 0;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    TF_LAUNCH(softmax_fwd_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), stream, s, rows, n, ld, sd, seed_dev, site, thresh, 1.f / (1.f - p));
    return launch_status("tf_softmax_dropout_fwd_f32");
}
// backward of the pair: dp (gradient w.r.t. the dropped probabilities) -> dL/d(scores), in place
extern "C" int tf_softmax_dropout_bwd_f32(const float* p, float* dp, int rows, int n, int ld, const uint32_t* seed_dev, uint32_t site, float pdrop,
                                          void* stream) {
    TF_REQUIRE(p && dp && seed_dev && rows >= 0 && n > 0 && n <= 64 * SM_MAXV && ld >= n && pdrop >= 0.f && pdrop < 1.f && (long)rows * ld < (1L << 32),
               "tf_softmax_dropout_bwd_f32: bad arguments (n=%d)", n);
    if (rows == 0) return 0;
    const uint32_t thresh = (uint32_t)((double)pdrop * 4294967296.0);
    TF_LAUNCH(softmax_bwd_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), stream, p, dp, rows, n, ld, seed_dev, site, thresh, 1.f / (1.f - pdrop));
    return launch_status("tf_softmax_dropout_bwd_f32");
}

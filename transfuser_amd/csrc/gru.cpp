// Fused auto-regressive waypoint decoder (model.py:611-646): all pred_len
// GRUCell steps + the 64->3 output layer + the running waypoint sum in ONE launch each way (the
// reference issues ~9 tiny kernels per step).  One workgroup handles the whole (B <= 32) batch:
// W_hh lives in LDS (padded rows, conflict-free), the recurrence state never leaves the CU.
//
//   x_{-1} = 0;  xin_t = [x_{t-1}, tp_x, -tp_y]  (or x_{t-1} alone);  h_t = GRUCell(xin_t, h_{t-1})
//   x_t = x_{t-1} + (W_out h_t + b_out)[:2];  wp[:, t] = x_t - (shift_x, 0)
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int GH = 64;       // config.gru_hidden_size
constexpr int GMAXB = 32;
constexpr int GMAXIN = 4;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// cache layout (floats): hs[(P+1)][B][64] | gates[P][B][4][64] (r, z, n, gh_n) | xin[P][B][4]
__global__ void __launch_bounds__(256) gru_wp_fwd_kernel(const float* __restrict__ z0, const float* __restrict__ tp, const float* __restrict__ w_ih,
                                                         const float* __restrict__ w_hh, const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         const float* __restrict__ w_out, const float* __restrict__ b_out, int B, int P, int nin,
                                                         float shift_x, float* __restrict__ wp, float* __restrict__ cache) {
    __shared__ float Whh[3 * GH][GH + 1];
    __shared__ float Wih[3 * GH][GMAXIN];
    __shared__ float h[2][GMAXB][GH];
    __shared__ float x[GMAXB][2];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * GH * GH; i += 256) Whh[i / GH][i % GH] = w_hh[i];
    for (int i = tid; i < 3 * GH * nin; i += 256) Wih[i / nin][i % nin] = w_ih[i];
    for (int i = tid; i < B * GH; i += 256) h[0][i / GH][i % GH] = z0[i];
    for (int i = tid; i < B * 2; i += 256) x[i / 2][i % 2] = 0.f;
    __syncthreads();
    float* hs = cache;
    float* gates = cache + (long)(P + 1) * B * GH;
    float* xins = gates + (long)P * B * 4 * GH;
    for (int i = tid; i < B * GH; i += 256) hs[i] = z0[i];
    int cur = 0;
    for (int t = 0; t < P; ++t) {
        for (int idx = tid; idx < B * GH; idx += 256) {
            const int b = idx / GH, j = idx % GH;
            float xin[GMAXIN];
            xin[0] = x[b][0]; xin[1] = x[b][1];
            if (nin == 4) { xin[2] = tp[b * 2]; xin[3] = -tp[b * 2 + 1]; }
            float gi[3], gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int J = g * GH + j;
                float a = b_ih[J];
                for (int c = 0; c < nin; ++c) a += Wih[J][c] * xin[c];
                gi[g] = a;
                float s = b_hh[J];
                for (int k = 0; k < GH; ++k) s += Whh[J][k] * h[cur][b][k];
                gh[g] = s;
            }
            const float r = sigmoidf_(gi[0] + gh[0]), zz = sigmoidf_(gi[1] + gh[1]);
            const float n = tanhf(gi[2] + r * gh[2]);
            h[cur ^ 1][b][j] = (1.f - zz) * n + zz * h[cur][b][j];
            float* gp = gates + (((long)t * B + b) * 4) * GH + j;
            gp[0] = r; gp[GH] = zz; gp[2 * GH] = n; gp[3 * GH] = gh[2];
            if (j < nin) xins[((long)t * B + b) * GMAXIN + j] = xin[j];
        }
        __syncthreads();
        cur ^= 1;
        for (int i = tid; i < B * GH; i += 256) hs[(long)(t + 1) * B * GH + i] = h[cur][i / GH][i % GH];
        for (int idx = tid; idx < B * 2; idx += 256) {
            const int b = idx / 2, c = idx % 2;
            float a = b_out[c];
            for (int k = 0; k < GH; ++k) a += w_out[c * GH + k] * h[cur][b][k];
            const float xn = a + x[b][c];
            x[b][c] = xn;
            wp[((long)b * P + t) * 2 + c] = xn - (c == 0 ? shift_x : 0.f);
        }
        __syncthreads();
    }
}

constexpr int GBT = 1024;     // backward workgroup: 16 waves hide the LDS latency of the per-step matrix products (4 waves: 175 us, LDS-latency bound)
// Gradients of every GRU / output parameter are accumulated (+=) into the given buffers; dz0 is written.
__global__ void __launch_bounds__(GBT) gru_wp_bwd_kernel(const float* __restrict__ dwp, const float* __restrict__ cache, const float* __restrict__ w_ih,
                                                         const float* __restrict__ w_hh, const float* __restrict__ w_out, int B, int P, int nin,
                                                         float* __restrict__ dz0, float* __restrict__ dw_ih, float* __restrict__ dw_hh,
                                                         float* __restrict__ db_ih, float* __restrict__ db_hh, float* __restrict__ dw_out,
                                                         float* __restrict__ db_out) {
    __shared__ float Whh[3 * GH][GH + 1];
    __shared__ float Wih[3 * GH][GMAXIN];
    __shared__ float dgi[GMAXB][3 * GH], dgh[GMAXB][3 * GH];
    __shared__ float dh[GMAXB][GH], dhn[GMAXB][GH];
    __shared__ float dxc[GMAXB][2], dxt[GMAXB][2];
    __shared__ float hst[2][GMAXB][GH];            // h_t and h_{t-1} of the current step (staged once per step; were re-read from global memory per product term)
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * GH * GH; i += GBT) Whh[i / GH][i % GH] = w_hh[i];
    for (int i = tid; i < 3 * GH * nin; i += GBT) Wih[i / nin][i % nin] = w_ih[i];
    for (int i = tid; i < B * GH; i += GBT) dh[i / GH][i % GH] = 0.f;
    for (int i = tid; i < B * 2; i += GBT) dxc[i / 2][i % 2] = 0.f;
    const float* hs = cache;
    const float* gates = cache + (long)(P + 1) * B * GH;
    const float* xins = gates + (long)P * B * 4 * GH;
    for (int i = tid; i < B * GH; i += GBT) hst[P & 1][i / GH][i % GH] = hs[(long)P * B * GH + i];
    __syncthreads();
    // The parameter gradients are summed over the P steps in REGISTERS (fixed element ownership: thread tid owns elements tid + GBT u) and
    // added to the global accumulators once at the end - the per-step global read-modify-write of 12 288 + 768 + 512 values made this
    // single-workgroup kernel 268 us long on the critical path of the step (the summation order over (t, b) is unchanged).
    constexpr int NHH = 3 * GH * GH / GBT, NIH = (3 * GH * GMAXIN + GBT - 1) / GBT;
    static_assert(3 * GH * GH % GBT == 0, "dW_hh elements per thread");
    float a_hh[NHH], a_ih[NIH], a_bi = 0.f, a_bh = 0.f, a_wo = 0.f, a_bo = 0.f;
#pragma unroll
    for (int u = 0; u < NHH; ++u) a_hh[u] = 0.f;
#pragma unroll
    for (int u = 0; u < NIH; ++u) a_ih[u] = 0.f;
    for (int t = P - 1; t >= 0; --t) {
        float (*ht)[GH] = hst[(t + 1) & 1];          // h_t
        float (*hp)[GH] = hst[t & 1];                // h_{t-1}
        for (int i = tid; i < B * GH; i += GBT) hp[i / GH][i % GH] = hs[(long)t * B * GH + i];
        for (int i = tid; i < B * 2; i += GBT) dxt[i / 2][i % 2] = dxc[i / 2][i % 2] + dwp[((long)(i / 2) * P + t) * 2 + (i % 2)];
        __syncthreads();
        // output layer: x_t = x_{t-1} + (W_out h_t + b_out)[:2]
        if (tid < 2 * GH) {
            const int c = tid / GH, k = tid % GH;
            float a = 0.f;
            for (int b = 0; b < B; ++b) a += dxt[b][c] * ht[b][k];
            a_wo += a;
        }
        if (tid >= GBT - 2) { const int c = tid - (GBT - 2); float a = 0.f; for (int b = 0; b < B; ++b) a += dxt[b][c]; a_bo += a; }
        for (int i = tid; i < B * GH; i += GBT) {
            const int b = i / GH, k = i % GH;
            dh[b][k] += w_out[k] * dxt[b][0] + w_out[GH + k] * dxt[b][1];
        }
        __syncthreads();
        // gates
        for (int i = tid; i < B * GH; i += GBT) {
            const int b = i / GH, j = i % GH;
            const float* gp = gates + (((long)t * B + b) * 4) * GH + j;
            const float r = gp[0], zz = gp[GH], n = gp[2 * GH], ghn = gp[3 * GH];
            const float g = dh[b][j];
            const float dn = g * (1.f - zz) * (1.f - n * n);
            const float dzp = g * (hp[b][j] - n) * zz * (1.f - zz);
            const float drp = dn * ghn * r * (1.f - r);
            dgi[b][j] = drp; dgi[b][GH + j] = dzp; dgi[b][2 * GH + j] = dn;
            dgh[b][j] = drp; dgh[b][GH + j] = dzp; dgh[b][2 * GH + j] = dn * r;
            dhn[b][j] = g * zz;
        }
        __syncthreads();
        // parameter gradients
#pragma unroll
        for (int u = 0; u < NHH; ++u) {
            const int i = tid + GBT * u, J = i / GH, k = i % GH;
            float a = 0.f;
            for (int b = 0; b < B; ++b) a += dgh[b][J] * hp[b][k];
            a_hh[u] += a;
        }
#pragma unroll
        for (int u = 0; u < NIH; ++u) {
            const int i = tid + GBT * u, J = i / GMAXIN, c = i % GMAXIN;     // (J, c) over 3 GH x GMAXIN; only c < nin is used
            float a = 0.f;
            if (c < nin && J < 3 * GH)
                for (int b = 0; b < B; ++b) a += dgi[b][J] * xins[((long)t * B + b) * GMAXIN + c];
            a_ih[u] += a;
        }
        if (tid < 3 * GH) {
            float a = 0.f, c = 0.f;
            for (int b = 0; b < B; ++b) { a += dgi[b][tid]; c += dgh[b][tid]; }
            a_bi += a; a_bh += c;
        }
        // state gradients
        for (int i = tid; i < B * GH; i += GBT) {
            const int b = i / GH, k = i % GH;
            float a = dhn[b][k];
            for (int J = 0; J < 3 * GH; ++J) a += Whh[J][k] * dgh[b][J];
            dhn[b][k] = a;
        }
        for (int i = tid; i < B * 2; i += GBT) {
            const int b = i / 2, c = i % 2;
            float a = dxt[b][c];
            for (int J = 0; J < 3 * GH; ++J) a += Wih[J][c] * dgi[b][J];
            dxc[b][c] = a;
        }
        __syncthreads();
        for (int i = tid; i < B * GH; i += GBT) dh[i / GH][i % GH] = dhn[i / GH][i % GH];
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NHH; ++u) dw_hh[tid + GBT * u] += a_hh[u];
#pragma unroll
    for (int u = 0; u < NIH; ++u) {
        const int i = tid + GBT * u, J = i / GMAXIN, c = i % GMAXIN;
        if (c < nin && J < 3 * GH) dw_ih[J * nin + c] += a_ih[u];
    }
    if (tid < 3 * GH) { db_ih[tid] += a_bi; db_hh[tid] += a_bh; }
    if (tid < 2 * GH) dw_out[tid] += a_wo;
    if (tid >= GBT - 2) db_out[tid - (GBT - 2)] += a_bo;
    for (int i = tid; i < B * GH; i += GBT) dz0[i] = dh[i / GH][i % GH];
}

}  // namespace

extern "C" long tf_gru_waypoints_cache_floats(int B, int P) { return (long)(P + 1) * B * GH + (long)P * B * 4 * GH + (long)P * B * GMAXIN; }

extern "C" int tf_gru_waypoints_fwd_f32(const float* z0, const float* target_point, const float* w_ih, const float* w_hh, const float* b_ih,
                                        const float* b_hh, const float* w_out, const float* b_out, int B, int hidden, int pred_len, int nin,
                                        float shift_x, float* wp, float* cache, void* stream) {
    TF_REQUIRE(z0 && w_ih && w_hh && b_ih && b_hh && w_out && b_out && wp && cache, "tf_gru_waypoints_fwd_f32: null argument");
    TF_REQUIRE(hidden == GH && B >= 1 && B <= GMAXB && (nin == 2 || (nin == 4 && target_point)) && pred_len >= 1,
               "tf_gru_waypoints_fwd_f32: unsupported shape (hidden=%d, B=%d, nin=%d)", hidden, B, nin);
    TF_LAUNCH(gru_wp_fwd_kernel, dim3(1), dim3(256), stream, z0, target_point, w_ih, w_hh, b_ih, b_hh, w_out, b_out, B, pred_len, nin, shift_x, wp, cache);
    return launch_status("tf_gru_waypoints_fwd_f32");
}

extern "C" int tf_gru_waypoints_bwd_f32(const float* dwp, const float* cache, const float* w_ih, const float* w_hh, const float* w_out, int B,
                                        int hidden, int pred_len, int nin, float* dz0, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                                        float* dw_out, float* db_out, void* stream) {
    TF_REQUIRE(dwp && cache && w_ih && w_hh && w_out && dz0 && dw_ih && dw_hh && db_ih && db_hh && dw_out && db_out, "tf_gru_waypoints_bwd_f32: null argument");
    TF_REQUIRE(hidden == GH && B >= 1 && B <= GMAXB && (nin == 2 || nin == 4), "tf_gru_waypoints_bwd_f32: unsupported shape");
    TF_LAUNCH(gru_wp_bwd_kernel, dim3(1), dim3(GBT), stream, dwp, cache, w_ih, w_hh, w_out, B, pred_len, nin, dz0, dw_ih, dw_hh, db_ih, db_hh, dw_out, db_out);
    return launch_status("tf_gru_waypoints_bwd_f32");
}

// GEMM main-loop lab (diagnostics, not part of the library): C = A (MxK) * B(NxK)^T, fp32 MFMA 32x32x2, interior tiles only.
// Variants of the global->LDS->MFMA pipeline of tf_gemm_engine.h, to find what bounds it on gfx950.
//   LAYOUT 0: K-major LDS [k][i+4], transposing scalar stash, ds_read_b32 fragments            (= the engine today)
//   LAYOUT 1: row-major LDS [i][BK+4], ds_write_b128 stash, ds_read_b128 fragments with a K permutation
//   PF: global prefetch distance in k-steps (1 = engine today, 2 = two register sets in flight)
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_lab tools/probe/gemm_lab.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int BM, int BN, int WAVES_M, int BK, int LAYOUT, int PF, int OCC, int PADK = 4, int NT = 256>
__global__ void __launch_bounds__(NT, OCC) lab(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int WAVES_N = (NT / 64) / WAVES_M, WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int KQ = BK / 4, NLA = BM * KQ / NT, NLB = BN * KQ / NT;
    static_assert(NLA >= 1 && NLB >= 1, "tile too small for the thread count");
    constexpr int PA = LAYOUT ? BK + PADK : BM + 4, PB = LAYOUT ? BK + PADK : BN + 4;
    constexpr int RA = LAYOUT ? BM : BK, RB = LAYOUT ? BN : BK;
    __shared__ __attribute__((aligned(16))) float As[2][RA][PA];
    __shared__ __attribute__((aligned(16))) float Bs[2][RB][PB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = N / BN, nt = (M / BM) * tiles_n, bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int i0 = (tile / tiles_n) * BM, j0 = (tile % tiles_n) * BN;
    const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;
    const float* pa[NLA]; const float* pb[NLB];
#pragma unroll
    for (int p = 0; p < NLA; ++p) { const int f = tid + p * NT; pa[p] = A + (long)(i0 + f / KQ) * K + (f % KQ) * 4; }
#pragma unroll
    for (int p = 0; p < NLB; ++p) { const int f = tid + p * NT; pb[p] = B + (long)(j0 + f / KQ) * K + (f % KQ) * 4; }
    float4 ra[PF][NLA], rb[PF][NLB];
    auto fetch = [&](int s, int k0) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) ra[s][p] = *reinterpret_cast<const float4*>(pa[p] + k0);
#pragma unroll
        for (int p = 0; p < NLB; ++p) rb[s][p] = *reinterpret_cast<const float4*>(pb[p] + k0);
    };
    auto stash = [&](int s, int buf) {
#pragma unroll
        for (int p = 0; p < NLA; ++p) {
            const int f = tid + p * NT, rr = f / KQ, kq = (f % KQ) * 4;
            if (LAYOUT) *reinterpret_cast<float4*>(&As[buf][rr][kq]) = ra[s][p];
            else { As[buf][kq][rr] = ra[s][p].x; As[buf][kq + 1][rr] = ra[s][p].y; As[buf][kq + 2][rr] = ra[s][p].z; As[buf][kq + 3][rr] = ra[s][p].w; }
        }
#pragma unroll
        for (int p = 0; p < NLB; ++p) {
            const int f = tid + p * NT, rr = f / KQ, kq = (f % KQ) * 4;
            if (LAYOUT) *reinterpret_cast<float4*>(&Bs[buf][rr][kq]) = rb[s][p];
            else { Bs[buf][kq][rr] = rb[s][p].x; Bs[buf][kq + 1][rr] = rb[s][p].y; Bs[buf][kq + 2][rr] = rb[s][p].z; Bs[buf][kq + 3][rr] = rb[s][p].w; }
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const int nkt = K / BK;
    fetch(0, 0);
    stash(0, 0);
    if (PF == 2 && nkt > 1) fetch(1 % PF, BK);
    __syncthreads();
    auto step = [&](int kt, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;          // register set holding tile kt (already in LDS); refilled with tile kt+PF
        const int cur = kt & 1;
        if (kt + PF < nkt) fetch(SLOT, (kt + PF) * BK);
        if (LAYOUT == 0) {
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[0][t] = As[cur][hi][wm0 + t * 32 + l31];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[0][t] = Bs[cur][hi][wn0 + t * 32 + l31];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int s = kk & 1;
                if (kk + 1 < BK / 2) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[s ^ 1][t] = As[cur][kk * 2 + 2 + hi][wm0 + t * 32 + l31];
#pragma unroll
                    for (int t = 0; t < TN; ++t) b[s ^ 1][t] = Bs[cur][kk * 2 + 2 + hi][wn0 + t * 32 + l31];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) MFMA(a[s][t], b[s][u], acc[t][u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // lane (row, hi) reads k = 8g + 4hi .. +3 with one ds_read_b128 and feeds MFMA steps t = 0..3 (A and B use the same k)
            float4 a[2][TM], b[2][TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[0][t] = *reinterpret_cast<const float4*>(&As[cur][wm0 + t * 32 + l31][4 * hi]);
#pragma unroll
            for (int t = 0; t < TN; ++t) b[0][t] = *reinterpret_cast<const float4*>(&Bs[cur][wn0 + t * 32 + l31][4 * hi]);
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                const int s = g & 1;
                if (g + 1 < BK / 8) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[s ^ 1][t] = *reinterpret_cast<const float4*>(&As[cur][wm0 + t * 32 + l31][8 * (g + 1) + 4 * hi]);
#pragma unroll
                    for (int t = 0; t < TN; ++t) b[s ^ 1][t] = *reinterpret_cast<const float4*>(&Bs[cur][wn0 + t * 32 + l31][8 * (g + 1) + 4 * hi]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) { MFMA(a[s][t].x, b[s][u].x, acc[t][u]); }
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) { MFMA(a[s][t].y, b[s][u].y, acc[t][u]); }
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) { MFMA(a[s][t].z, b[s][u].z, acc[t][u]); }
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) { MFMA(a[s][t].w, b[s][u].w, acc[t][u]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kt + 1 < nkt) stash((SLOT + 1) % PF, cur ^ 1);
        __syncthreads();
    };
    for (int kt = 0; kt < nkt; kt += PF) {
        step(kt, std::integral_constant<int, 0>());
        if (PF == 2 && kt + 1 < nkt) step(kt + 1, std::integral_constant<int, 1 % PF>());
    }
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int j = j0 + wn0 + u * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wm0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                C[(long)i * N + j] = acc[t][u][e];
            }
        }
}

static float *dA, *dB, *dC; static std::vector<float> hC, hRef; static int gM, gN, gK;
template <int BM, int BN, int WAVES_M, int BK, int LAYOUT, int PF, int OCC, int PADK = 4, int NT = 256> void run(const char* name) {
    const int M = gM / BM * BM, N = gN / BN * BN, K = gK;
    dim3 grid((M / BM) * (N / BN));
    hipMemset(dC, 0, (size_t)gM * gN * 4);
    lab<BM, BN, WAVES_M, BK, LAYOUT, PF, OCC, PADK, NT><<<grid, NT>>>(dA, dB, dC, M, N, K);
    hipMemcpy(hC.data(), dC, (size_t)64 * gN * 4, hipMemcpyDeviceToHost);
    double err = 0;   // check the first 64 rows against the reference (row stride N of the cropped problem)
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double d = hC[(size_t)i * N + j] - hRef[(size_t)i * 64 + j]; err = d * d > err ? d * d : err; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) lab<BM, BN, WAVES_M, BK, LAYOUT, PF, OCC, PADK, NT><<<grid, NT>>>(dA, dB, dC, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %dx%dx%d: %7.1f us %6.1f TFLOP/s  maxerr %.1e%s\n", name, M, N, K, ms * 50, 2.0 * M * N * K / (ms / 20 * 1e9), err > 0 ? sqrt(err) : 0.0,
           hipGetLastError() != hipSuccess ? " LAUNCH-ERROR" : "");
}
#define RUN(...) run<__VA_ARGS__>(#__VA_ARGS__)

int main(int argc, char** argv) {
    gM = argc > 1 ? atoi(argv[1]) : 1792; gN = argc > 2 ? atoi(argv[2]) : 6016; gK = argc > 3 ? atoi(argv[3]) : 1504;
    std::vector<float> hA((size_t)gM * gK), hB((size_t)gN * gK);
    srand(1);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
    hRef.assign(64 * 64, 0.f); hC.resize((size_t)64 * gN);
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double s = 0; for (int k = 0; k < gK; ++k) s += (double)hA[(size_t)i * gK + k] * hB[(size_t)j * gK + k]; hRef[i * 64 + j] = (float)s; }
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)gM * gN * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    RUN(64, 64, 2, 16, 0, 1, 4); RUN(128, 128, 2, 16, 0, 2, 3);
    RUN(128, 128, 2, 16, 0, 1, 2, 4, 512); RUN(128, 128, 4, 16, 0, 1, 2, 4, 512); RUN(128, 128, 2, 16, 0, 2, 2, 4, 512); RUN(128, 128, 2, 32, 0, 1, 2, 4, 512);
    RUN(128, 256, 2, 16, 0, 1, 1, 4, 512); RUN(256, 128, 4, 16, 0, 1, 1, 4, 512); RUN(128, 128, 2, 16, 0, 1, 1, 4, 512);

    return 0;
}

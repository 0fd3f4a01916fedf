// GEMM lab 2 (diagnostics, not part of the library): the LDS-DMA kernels of transfuser_amd/csrc/tf_gemm_dma.h against the r01 engine
// kernel (tf_gemm_engine.h) on the training step's plain GEMM shapes.  Every configuration is checked against a naive GPU reference.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I transfuser_amd/csrc -o tools/probe/gemm_lab2 tools/probe/gemm_lab2.cpp
#include "tf_gemm_dma.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>

namespace tf {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
bool plan_lookup(const char*, int, int, int, int, int, GemmPlan*) { return false; }
void plan_store(const char*, int, int, int, int, int, const GemmPlan&) {}
bool autotune_enabled() { return false; }
bool forced_plan(GemmPlan*) { return false; }
}
using namespace tf;

__global__ void ref_gemm(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, int at, int bt) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const float a = at ? A[(long)k * lda + i] : A[(long)i * lda + k];
        const float b = bt ? B[(long)k * ldb + j] : B[(long)j * ldb + k];
        s = fmaf(a, b, s);
    }
    C[(long)i * N + j] = s;
}

struct Shape { int M, N, K, at, bt; const char* name; };

static float *dA, *dB, *dC, *dR;
static std::vector<float> hC, hR;

template <class F>
static void run(const char* tag, const Shape& s, F&& launch, bool check = true) {
    hipMemset(dC, 0, (size_t)s.M * s.N * 4);
    launch();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%-34s %-22s LAUNCH ERROR %s\n", tag, s.name, hipGetErrorString(e)); (void)hipGetLastError(); return; }
    double maxerr = 0, maxref = 0;
    if (check) {
        hipMemcpy(hC.data(), dC, (size_t)s.M * s.N * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < (size_t)s.M * s.N; ++i) {
            maxerr = fmax(maxerr, fabs((double)hC[i] - hR[i]));
            maxref = fmax(maxref, fabs((double)hR[i]));
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 100.0, tf_s = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
    printf("%-34s %-22s %8.1f us %7.1f TF/s  relerr %.1e%s\n", tag, s.name, us, tf_s, maxerr / (maxref + 1e-30), maxerr > 2e-4 * maxref ? "  <-- MISMATCH" : "");
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

static PlainOp plain(const float* p, long ld, int rows, int cols) {
    PlainOp o; o.p = p; o.ld = ld; o.rows = rows; o.cols = cols; o.vec = 1; o.s_outer = 0; o.s_inner = 0; o.inner = 1; return o;
}

template <bool A_KC, bool B_KC>
static void bench_shape(const Shape& s) {
    const int M = s.M, N = s.N, K = s.K;
    // A: KC stored [M][K]; IC stored [K][M].  B: KC stored [N][K]; IC stored [K][N].
    const long lda = A_KC ? K : M, ldb = B_KC ? K : N;
    PlainOp A = A_KC ? plain(dA, lda, M, K) : plain(dA, lda, K, M);
    PlainOp B = B_KC ? plain(dB, ldb, N, K) : plain(dB, ldb, K, N);
    GemmEpi ep; ep.C = dC; ep.ldc = N; ep.ldcj = 1; ep.sc_outer = 0; ep.sc_inner = 0; ep.inner = 1; ep.bias = nullptr; ep.sbias = 0;
    ep.res = nullptr; ep.ldres = 0; ep.alpha = 1.f; ep.relu = 0; ep.mode = 0;
    dim3 rg((N + 255) / 256, M);
    ref_gemm<<<rg, 256>>>(dA, dB, dR, M, N, K, lda, ldb, !A_KC, !B_KC);
    hipDeviceSynchronize();
    hipMemcpy(hR.data(), dR, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    printf("---- %s  M=%d N=%d K=%d  A %s  B %s\n", s.name, M, N, K, A_KC ? "[m][k]" : "[k][m]", B_KC ? "[n][k]" : "[k][n]");
#define OLD(BM_, BN_, WM_, BK_) run("r01 engine " #BM_ "x" #BN_ " bk" #BK_, s, [&] { launch_cfg<BM_, BN_, WM_, BK_, PlainOp, A_KC, PlainOp, B_KC>(A, B, ep, M, N, K, 1, 1, nullptr); })
    OLD(64, 64, 2, 16);
    OLD(128, 64, 2, 16);
    OLD(128, 128, 2, 16);
#define DMA(TM_, TN_, WMM_, WNN_, BK_, ST_, OCC_) \
    run("dma w" #WMM_ "x" #WNN_ " t" #TM_ "x" #TN_ " bk" #BK_ " st" #ST_ " occ" #OCC_, s, [&] { launch_dma_cfg<TM_, TN_, WMM_, WNN_, BK_, ST_, A_KC, B_KC, OCC_>(A, B, ep, M, N, K, 1, 1, nullptr); })
#ifndef LAB_FEW
    // single-wave workgroups
    DMA(2, 2, 1, 1, 16, 2, 2);
    DMA(2, 2, 1, 1, 16, 3, 2);
    DMA(2, 2, 1, 1, 32, 2, 2);
    DMA(2, 3, 1, 1, 16, 3, 2);
    DMA(3, 2, 1, 1, 16, 3, 2);
    DMA(2, 4, 1, 1, 16, 2, 2);
    DMA(4, 2, 1, 1, 16, 2, 2);
    DMA(3, 3, 1, 1, 16, 2, 2);
    // two waves
    DMA(2, 2, 2, 1, 16, 3, 2);
    DMA(2, 2, 1, 2, 16, 3, 2);
    DMA(2, 3, 2, 1, 16, 3, 2);
    // four waves
    DMA(2, 1, 2, 2, 16, 3, 3);
    DMA(1, 2, 2, 2, 16, 3, 3);
    DMA(2, 2, 2, 2, 16, 4, 2);
    DMA(2, 2, 2, 2, 32, 2, 2);
#endif
    DMA(1, 1, 2, 2, 16, 4, 4);
    DMA(2, 2, 2, 2, 16, 3, 2);
}

int main(int argc, char** argv) {
    const Shape shapes[] = {
        {1740, 6048, 1512, 0, 0, "gpt4 fc1 fwd"},        {1740, 1512, 6048, 0, 0, "gpt4 fc2 fwd"},     {1740, 4536, 1512, 0, 0, "gpt4 qkv fwd"},
        {1740, 1512, 6048, 0, 1, "gpt4 fc1 dgrad"},      {6048, 1512, 1740, 1, 1, "gpt4 fc1 wgrad"},   {7040, 576, 576, 0, 0, "s3 1x1 fwd"},
        {576, 576, 7040, 1, 1, "s3 1x1 wgrad"},          {2560, 576, 576, 0, 0, "s3 lidar 1x1 fwd"},   {28160, 216, 216, 0, 0, "s2 1x1 fwd"},
        {1740, 576, 2304, 0, 0, "gpt3 fc2 fwd"},         {1741, 1508, 1516, 0, 0, "ragged"},          {4096, 4096, 4096, 0, 0, "4096^3"},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const int zero = argc > 2 ? atoi(argv[2]) : 0;       // 1: zero-filled operands (DVFS probe: same instruction stream, no data toggling)
    size_t maxA = 0, maxB = 0, maxC = 0;
    for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxB = std::max(maxB, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N); }
    hipMalloc(&dA, maxA * 4 + 64); hipMalloc(&dB, maxB * 4 + 64); hipMalloc(&dC, maxC * 4); hipMalloc(&dR, maxC * 4);
    std::vector<float> h(std::max(maxA, maxB));
    srand(1);
    for (auto& v : h) v = zero ? 0.f : (float)(rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(dA, h.data(), maxA * 4, hipMemcpyHostToDevice);
    for (auto& v : h) v = zero ? 0.f : (float)(rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(dB, h.data(), maxB * 4, hipMemcpyHostToDevice);
    hC.resize(maxC); hR.resize(maxC);
    int idx = 0;
    for (auto& s : shapes) {
        if (only >= 0 && only != idx++) continue;
        if (!s.at && !s.bt) bench_shape<true, true>(s);
        else if (!s.at && s.bt) bench_shape<true, false>(s);
        else if (s.at && s.bt) bench_shape<false, false>(s);
        else bench_shape<false, true>(s);
    }
    return 0;
}

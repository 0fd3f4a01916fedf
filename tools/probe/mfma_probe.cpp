// Micro-probe (diagnostics, not part of the library): v_mfma_f32_32x32x2_f32 issue ceilings on gfx950.
//   acc = independent accumulators per wave (1, 2, 4), waves/SIMD = resident 256-thread blocks per CU,
//   mode 0: operands from registers; mode 1: operands re-read from LDS each step (software pipelined);
//   mode 2: as 1 plus a __syncthreads() every 8 steps (the engine's k-step structure).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/probe/mfma_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC, int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, int rnd) {
    __shared__ float lds[2][16][132];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    for (int i = threadIdx.x; i < 2 * 16 * 132; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        (&lds[0][0][0])[i] = rnd ? (float)(int)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f : (float)(i & 7) * 0.001f; }
    __syncthreads();
    f32x16 acc[ACC];
    for (int a = 0; a < ACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = rnd ? lds[0][lane & 15][lane] : lane * 0.01f, b0 = rnd ? lds[1][lane & 15][lane + 64] : lane * 0.02f;
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        if (MODE == 0) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
        } else {
            float fa[2][2], fb[2][2];
            fa[0][0] = lds[cur][hi][l31]; fa[0][1] = lds[cur][hi][32 + l31]; fb[0][0] = lds[cur][hi][64 + l31]; fb[0][1] = lds[cur][hi][96 + l31];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int s = kk & 1;
                if (kk < 7) {
                    fa[s ^ 1][0] = lds[cur][kk * 2 + 2 + hi][l31]; fa[s ^ 1][1] = lds[cur][kk * 2 + 2 + hi][32 + l31];
                    fb[s ^ 1][0] = lds[cur][kk * 2 + 2 + hi][64 + l31]; fb[s ^ 1][1] = lds[cur][kk * 2 + 2 + hi][96 + l31];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][a & 1], fb[s][(a >> 1) & 1], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int a = 0; a < ACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}

template <int ACC, int MODE>
void run(int wps, float* d, int rnd) {
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<ACC, MODE><<<blocks, 256>>>(d, 10, rnd);
    hipEventRecord(e0);
    probe<ACC, MODE><<<blocks, 256>>>(d, iters, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * ACC * 4096.0;
    printf("acc=%d mode=%d waves/SIMD=%d %s: %.1f TFLOP/s (%.2f ms)\n", ACC, MODE, wps, rnd ? "random operands" : "constant operands", flops / ms / 1e9, ms);
}

int main() {
    float* d; hipMalloc(&d, 1024);
    for (int rnd = 0; rnd <= 1; ++rnd)      // DVFS: identical instruction stream, constant vs random operand bits
        for (int wps = 1; wps <= 4; ++wps) {
            run<1, 0>(wps, d, rnd); run<2, 0>(wps, d, rnd); run<4, 0>(wps, d, rnd);
            run<1, 1>(wps, d, rnd); run<4, 1>(wps, d, rnd);
            run<1, 2>(wps, d, rnd); run<4, 2>(wps, d, rnd);
        }
    return 0;
}

// Device primitives for the TransFuser gfx950 kernels.
//
// Production build: hipcc --offload-arch=gfx950 (CDNA4, wave64).  With -DTF_EMU (tests/emu only)
// the same kernels compile for the host against a fiber emulator; that build is test
// infrastructure and is never part of libtransfuser_hip.so.
#pragma once
#ifdef TF_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace tf {

constexpr int kWave = 64;  // CDNA wavefront

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// instruction-scheduling fence for hipcc (nothing may be moved across it); no-op in the host emulation
#ifdef TF_EMU
#define TF_SCHED_FENCE() ((void)0)
#else
#define TF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifdef TF_EMU
__forceinline__ int lane_id() { return emu::cur_lane(); }

// v_mfma_f32_32x32x2_f32: A lane l holds A[i=l&31][k=l>>5]; B lane l holds B[k=l>>5][j=l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5); k-ordered fmaf chain (exact f32).
__forceinline__ void mfma_32x32x2(float a, float b, f32x16& acc) {
    float* s = emu::wave_scratch();
    const int l = lane_id();
    s[l] = a;
    s[64 + l] = b;
    emu::wave_barrier();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float c = acc[r];
        c = fmaf(s[i], s[64 + j], c);
        c = fmaf(s[32 + i], s[64 + 32 + j], c);
        acc[r] = c;
    }
    emu::wave_barrier();
}
// v_mfma_f32_32x32x16_bf16 with fp32 inputs rounded to bf16 (RNE): A lane l holds A[i=l&31][k=8*(l>>5)+j], B lane l holds B[k=8*(l>>5)+j][j'=l&31],
// j = 0..7; D layout as mfma_32x32x2.  Emulation: exact bf16 x bf16 products accumulated in fp32 in k order (hardware order may differ).
__forceinline__ float emu_bf16_round(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return f;
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}
__forceinline__ void mfma_32x32x16_bf16(const float (&a)[8], const float (&b)[8], f32x16& acc) {
    float* s = emu::wave_scratch();
    const int l = lane_id(), jc = l & 31, hi = l >> 5;
    for (int j = 0; j < 8; ++j) {        // one k pair (k = j of lane half 0, k = 8 + j of lane half 1) per round through the wave scratch
        s[l] = emu_bf16_round(a[j]);
        s[64 + l] = emu_bf16_round(b[j]);
        emu::wave_barrier();
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float c = acc[r];
            c = fmaf(s[i], s[64 + jc], c);
            c = fmaf(s[32 + i], s[64 + 32 + jc], c);
            acc[r] = c;
        }
        emu::wave_barrier();
    }
}
__forceinline__ float shfl(float v, int src) {
    float* s = emu::wave_scratch();
    s[128 + lane_id()] = v;
    emu::wave_barrier();
    float r = s[128 + (src & 63)];
    emu::wave_barrier();
    return r;
}
__forceinline__ float shfl_xor(float v, int m) { return shfl(v, lane_id() ^ m); }
__forceinline__ float shfl_down(float v, int d) { int s = lane_id() + d; return shfl(v, s < 64 ? s : lane_id()); }
#else
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ void mfma_32x32x2(float a, float b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// fp32 operands rounded to bf16 in registers (v_cvt_pk_bf16_f32, RNE) -> v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate), fp32 accumulate
__device__ __forceinline__ void mfma_32x32x16_bf16(const float (&a)[8], const float (&b)[8], f32x16& acc) {
    f32x8 x, y;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = a[j]; y[j] = b[j]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_convertvector(x, bf16x8), __builtin_convertvector(y, bf16x8), acc, 0, 0, 0);
}
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

// Block-wide sum for blocks of NW waves (blockDim.x == NW*64); result valid in every thread.
// ``red`` is caller-provided LDS of >= NW floats.  Fixed summation order => deterministic.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// Counter-based RNG for dropout (our own stream: torch's Philox sequence is not reproduced;
// parity tests run with p = 0).  One 32-bit hash per element, keyed by (seed, site, index).
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(uint32_t seed, uint32_t site, uint32_t idx, uint32_t thresh) {
    // keep iff hash >= thresh, thresh = p * 2^32
    return hash32(idx * 0x9E3779B9U + hash32(seed ^ (site * 0x85ebca6bU))) >= thresh;
}

}  // namespace tf

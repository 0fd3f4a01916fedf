// CPU emulator for the HIP kernels under transfuser_amd/csrc.  TEST INFRASTRUCTURE ONLY.
//
// The product library (libtransfuser_hip.so) is built by hipcc for gfx950 and never contains
// this code.  tests/emu builds the SAME kernel sources with the host clang++ and -DTF_EMU so
// that indexing / tiling / barrier logic can be checked against the oracle on a machine
// without a GPU (and under ASan).  A workgroup's threads run as cooperative fibers on one OS
// thread; __syncthreads() and the wave-level collectives (shuffles, MFMA) are rendezvous
// points.  Wave64 lane layouts of the MFMA forms follow /opt/skills/guides/cdna_hip_programming.md.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef void* hipStream_t;

namespace emu {
struct Fiber;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern Fiber* g_cur;
const dim3& cur_tid();
int cur_lane();  // 0..63
void block_barrier();
void wave_barrier();
float* wave_scratch();  // 4*64 floats per wave, shared by the wave's lanes
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::cur_tid())
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

static inline void __syncthreads() { emu::block_barrier(); }

// single OS thread => plain RMW is atomic
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
static inline int atomicCAS(int* p, int cmp, int v) { int o = *p; if (o == cmp) *p = v; return o; }

static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

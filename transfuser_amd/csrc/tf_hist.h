// H1: the 2-bin LiDAR height histogram (team_code_transfuser/data.py:446-470), single pass over the cloud.
//
// Three launches on the caller's stream without a workspace (two with one: the *_ws entry points keep the counters in a caller-owned buffer that
// every call leaves zeroed, hist_finish_ws_kernel): (1) clear the (B, 2, 256, 256) output, viewed as int32 counters; (2) one thread
// per point: one 16-byte load, bin, one return-less int32 atomic on the counter of its cell (the counters of a step's clouds are 5 MB:
// they live in L2 / the Infinity Cache); (3) counters -> min(cnt, 5) / 5 in place.  The cloud is read ONCE (the round-3 kernel had every
// one of 32 slab blocks per sample re-scan it).  Same-address atomics queue (~0.1 us each), so a wave first merges RUNS of equal cells
// (a spinning LiDAR emits neighbours back to back): only the first lane of a run issues the atomic, with min(run length, 5) - exact,
// because only min(total, 5) is ever read and a term capped at 5 leaves that minimum unchanged.  Integer-exact by construction.
#pragma once
#include "tf_common.h"

namespace tf {

// cell of a point in the OUTPUT layout (channel 0 = above -2.3 m, 1 = below; rot90(-1) of the histogramdd grid): -1 = outside
template <typename T>
__device__ __forceinline__ int hist_cell(T x, T y, T z) {
    if (!(x >= (T)-16 && x <= (T)16 && y >= (T)-32 && y <= (T)0)) return -1;
    int xb = (int)floor(x * (T)8) + 128; if (xb > 255) xb = 255;
    int yb = (int)floor(y * (T)8) + 256; if (yb > 255) yb = 255;
    return (((z <= (T)-2.3) ? 1 : 0) * 256 + yb) * 256 + (255 - xb);
}

// wave-level run merge + atomic; every lane of the wave must call it (cell = -1 for lanes without a point)
__device__ __forceinline__ void hist_add(int* __restrict__ counters, int cell) {
    const int lane = lane_id();
    const float kf = __int_as_float(cell);
    const int prev = __float_as_int(shfl(kf, lane - 1));
    int run = 1;
    bool open = true;
#pragma unroll
    for (int d = 1; d <= 4; ++d) {
        const int nxt = __float_as_int(shfl(kf, lane + d));
        open = open && (lane + d < 64) && nxt == cell;
        run += open ? 1 : 0;
    }
    if (cell >= 0 && (lane == 0 || prev != cell)) atomicAdd(counters + cell, run);
}

static __global__ void __launch_bounds__(256) hist_clear_kernel(float4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static __global__ void __launch_bounds__(256) hist_finish_kernel(float4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = out[i];
    const int a = __float_as_int(v.x), b = __float_as_int(v.y), c = __float_as_int(v.z), d = __float_as_int(v.w);
    out[i] = make_float4((float)(a < 5 ? a : 5) / 5.0f, (float)(b < 5 ? b : 5) / 5.0f, (float)(c < 5 ? c : 5) / 5.0f, (float)(d < 5 ? d : 5) / 5.0f);
}

// two-launch form: the counters live in a caller-owned int32 workspace that is ALL ZERO between calls - this kernel turns them into the output
// and re-zeroes them, so no clear launch is needed in front of the next call
static __global__ void __launch_bounds__(256) hist_finish_ws_kernel(int4* __restrict__ counters, float4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int4 v = counters[i];
    int4 zero; zero.x = zero.y = zero.z = zero.w = 0;
    counters[i] = zero;
    out[i] = make_float4((float)(v.x < 5 ? v.x : 5) / 5.0f, (float)(v.y < 5 ? v.y : 5) / 5.0f, (float)(v.z < 5 ? v.z : 5) / 5.0f, (float)(v.w < 5 ? v.w : 5) / 5.0f);
}

}  // namespace tf

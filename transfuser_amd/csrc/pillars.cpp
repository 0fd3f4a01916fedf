// H2 - PointPillars front-end (team_code_transfuser/point_pillar.py:37-122): index-exact pillar ids without a sort.
//
// The reference builds (batch, x_idx, y_idx) rows, calls torch.unique(dim=0, return_inverse=True) (a lexicographic SORT) and
// uses torch_scatter for the per-pillar mean / max.  On the GPU the sorted-unique rank of a pillar is simply the number of
// occupied grid cells with a smaller (b, x_idx, y_idx) key, so: mark an occupancy grid, exclusive-scan it, and every point reads
// its pillar id from the scanned grid - integer-exact by construction, HBM-bound (12 B/point + 4 B/cell), no sort.
// Everything is fp32 / int32; kept points are compacted in their original order (stable), like points[keep].
#include "tf_common.h"
#include <stdlib.h>
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

inline int pl_blocks(long n, int cap = 8192) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// point_pillar.py:70-83: keep iff min <= x < max (both axes); coords = ((xy - min) * ppm).long()  [fp32 sub, fp32 mul, trunc]
__global__ void __launch_bounds__(256) pillar_keys_kernel(const float* __restrict__ pts, const int32_t* __restrict__ npts, int B, int Nmax, int F,
                                                          float min_x, float max_x, float min_y, float max_y, float ppm, int GX, int GY,
                                                          int32_t* __restrict__ keys, int32_t* __restrict__ keep, int32_t* __restrict__ occ) {
    const long total = (long)B * Nmax;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Nmax), j = (int)(i % Nmax);
        const int n = npts ? (npts[b] < Nmax ? npts[b] : Nmax) : Nmax;
        int key = -1;
        if (j < n) {
            const float x = pts[i * F], y = pts[i * F + 1];
            if (x >= min_x && x < max_x && y >= min_y && y < max_y) {
                const float fx = (x - min_x) * ppm, fy = (y - min_y) * ppm;
                int cx = (int)fx, cy = (int)fy;                 // >= 0; may reach nx / ny when (x - min) rounds up to the range
                cx = cx < GX ? cx : GX - 1;
                cy = cy < GY ? cy : GY - 1;
                key = (b * GX + cx) * GY + cy;
            }
        }
        keys[i] = key;
        keep[i] = key >= 0;
        if (key >= 0) occ[key] = 1;
    }
}

// ---- exclusive scan of int32 flags / counts: 1024 elements per block, three small kernels
__global__ void __launch_bounds__(256) scan_block_sums_kernel(const int32_t* __restrict__ in, long n, int32_t* __restrict__ bsum) {
    __shared__ int sm[256];
    const long base = (long)blockIdx.x * 1024 + threadIdx.x * 4;
    int s = 0;
    for (int k = 0; k < 4; ++k)
        if (base + k < n) s += in[base + k];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0];
}

__global__ void __launch_bounds__(256) scan_carry_kernel(int32_t* __restrict__ bsum, int nb, int32_t* __restrict__ total) {
    __shared__ int sm[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {       // inclusive Hillis-Steele
            const int t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
            __syncthreads();
            sm[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) bsum[i] = carry + sm[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += sm[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(256) scan_final_kernel(const int32_t* __restrict__ in, long n, const int32_t* __restrict__ boff, int32_t* __restrict__ out) {
    __shared__ int sm[256];
    const long base = (long)blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
    for (int k = 0; k < 4; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += t;
        __syncthreads();
    }
    int run = boff[blockIdx.x] + sm[threadIdx.x] - s;
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// ---- both exclusive scans of the pillar index (kept-point positions over the keys, sorted-unique ranks over the occupancy grid) in TWO launches:
// one pass of 1024-element block sums over the concatenation [keys >= 0 | occ], then every block adds up the sums in front of it inside its
// own array (<= ~700 values: one strided pass of the block), scans its elements and writes; the occupancy blocks also emit
// cellkey[rank] = cell (round 3: 2 x 3 scan launches + a cell-key launch).  totals[0] = kept points, totals[1] = pillars.
__global__ void __launch_bounds__(256) scan2_block_sums_kernel(const int32_t* __restrict__ keys, long nA, const int32_t* __restrict__ occ, long nB, int nbA,
                                                               int32_t* __restrict__ bsum) {
    __shared__ int sm[4];
    const bool isA = (int)blockIdx.x < nbA;
    const long n = isA ? nA : nB, base = (long)(isA ? blockIdx.x : blockIdx.x - nbA) * 1024 + threadIdx.x * 4;
    const int32_t* in = isA ? keys : occ;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = in[base + k < n ? base + k : n - 1];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (base + k < n) ? (isA ? (v[k] >= 0) : v[k]) : 0;
    s = (int)wave_sum((float)s);                   // <= 256 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void __launch_bounds__(256) scan2_final_kernel(const int32_t* __restrict__ keys, long nA, const int32_t* __restrict__ occ, long nB, int nbA, int nbB,
                                                          const int32_t* __restrict__ bsum, int32_t* __restrict__ pos, int32_t* __restrict__ rank,
                                                          int32_t* __restrict__ cellkey, int32_t* __restrict__ totals) {
    __shared__ int sm[256];
    __shared__ int red[4];
    const bool isA = (int)blockIdx.x < nbA;
    const int b = isA ? blockIdx.x : blockIdx.x - nbA, first = isA ? 0 : nbA, nb = isA ? nbA : nbB;
    // offset of this block = sum of the block sums in front of it (same array): eight loads in flight per trip
    int pre = 0;
    for (int j0 = 0; j0 < b; j0 += 8 * 256) {
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + threadIdx.x + 256 * u; t[u] = bsum[first + (j < b ? j : 0)]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) pre += (j0 + (int)threadIdx.x + 256 * u < b) ? t[u] : 0;
    }
    // (block totals stay far below 2^24: fp32 wave sums are exact)
    pre = (int)wave_sum((float)(pre & 0xffff)) + ((int)wave_sum((float)(pre >> 16)) << 16);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
    __syncthreads();
    const int boff = (red[0] + red[1]) + (red[2] + red[3]);
    const long n = isA ? nA : nB, base = (long)b * 1024 + threadIdx.x * 4;
    const int32_t* in = isA ? keys : occ;
    int raw[4], v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[k] = in[base + k < n ? base + k : n - 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? (isA ? (raw[k] >= 0) : raw[k]) : 0; s += v[k]; }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {       // inclusive Hillis-Steele over the 256 thread sums
        const int t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
        __syncthreads();
        sm[threadIdx.x] += t;
        __syncthreads();
    }
    int run = boff + sm[threadIdx.x] - s;
    int32_t* out = isA ? pos : rank;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) {
            out[base + k] = run;
            if (!isA && v[k]) cellkey[run] = (int32_t)(base + k);
        }
        run += v[k];
    }
    if (b == nb - 1 && threadIdx.x == 255) totals[isA ? 0 : 1] = boff + sm[255];
}

// rank -> cell key of every occupied cell (= torch.unique's sorted unique_coords, point_pillar.py:88)
__global__ void __launch_bounds__(256) pillar_cells_kernel(const int32_t* __restrict__ occ, const int32_t* __restrict__ rank, long ncells,
                                                           int32_t* __restrict__ cellkey) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ncells; i += (long)gridDim.x * 256)
        if (occ[i]) cellkey[rank[i]] = (int32_t)i;
}

// stable compaction of the kept points + inverse indices + per-pillar xyz sums / counts (scatter_mean numerator, :61).
// The sums are FIXED-POINT (2^-24 m) 64-bit integers: integer atomics commute, so the result does not depend on the order the atomics land
// in (run-to-run reproducible; torch_scatter's fp32 atomics - and this file's until round 4 - are not), and x 2^24 is exact for every
// |coordinate| >= 0.5 m (fp32 has <= 23 fraction bits there), within 6e-8 m below.  A thread owns FOUR consecutive points: a spinning
// LiDAR emits a pillar's points back to back, so runs of equal pillars inside the quad are merged in registers and cost one set of atomics.
constexpr float kPillarFix = 16777216.0f;      // 2^24
__global__ void __launch_bounds__(256) pillar_gather_kernel(const float* __restrict__ pts, int F, const int32_t* __restrict__ keys,
                                                            const int32_t* __restrict__ pos, const int32_t* __restrict__ rank, long n_all,
                                                            float* __restrict__ pts4, int32_t* __restrict__ inv, long long* __restrict__ sums) {
    const long nq = (n_all + 3) / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
        int cur = -1;
        long long sx = 0, sy = 0, sz = 0, sn = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long i = q * 4 + j;
            const int key = i < n_all ? keys[i] : -1;
            int r = -1;
            if (key >= 0) {
                r = rank[key];
                const int row = pos[i];
                const float x = pts[i * F], y = pts[i * F + 1], z = pts[i * F + 2], w = pts[i * F + 3];
                *reinterpret_cast<float4*>(pts4 + (long)row * 4) = make_float4(x, y, z, w);
                inv[row] = r;
                if (r != cur && cur >= 0) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4), (unsigned long long)sx);
                    atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4 + 1), (unsigned long long)sy);
                    atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4 + 2), (unsigned long long)sz);
                    atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4 + 3), (unsigned long long)sn);
                    sx = sy = sz = sn = 0;
                }
                cur = r;
                sx += (long long)llrintf(x * kPillarFix); sy += (long long)llrintf(y * kPillarFix); sz += (long long)llrintf(z * kPillarFix); sn += 1;
            }
        }
        if (cur >= 0) {
            atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4), (unsigned long long)sx);
            atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4 + 1), (unsigned long long)sy);
            atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4 + 2), (unsigned long long)sz);
            atomicAdd(reinterpret_cast<unsigned long long*>(sums + (long)cur * 4 + 3), (unsigned long long)sn);
        }
    }
}

// decorate (:54-67): [x, y, z, i, xyz - pillar mean, x - x_center, y - y_center]; quirk Q15: x_center comes from the y index
// (column 2 of the (b, x_idx, y_idx) rows) + min_x and y_center from the x index + min_y - reproduced literally.
__global__ void __launch_bounds__(256) pillar_decorate_kernel(const float* __restrict__ pts4, const int32_t* __restrict__ inv,
                                                              const long long* __restrict__ sums, const int32_t* __restrict__ cellkey, long N, int GX,
                                                              int GY, float ppm, float min_x, float min_y, float* __restrict__ feat) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
        const float4 p = *reinterpret_cast<const float4*>(pts4 + i * 4);
        const int r = inv[i];
        const long long* sr = sums + (long)r * 4;
        const double inv_fix = 1.0 / (double)kPillarFix;
        float4 s;       // the pillar's coordinate SUMS rounded to fp32 once (the reference accumulates them in fp32, in an unspecified order)
        s.x = (float)((double)sr[0] * inv_fix); s.y = (float)((double)sr[1] * inv_fix); s.z = (float)((double)sr[2] * inv_fix); s.w = (float)sr[3];
        const float cnt = s.w < 1.f ? 1.f : s.w;
        const int key = cellkey[r];
        const int cy = key % GY, cx = (key / GY) % GX;
        const float xc = (float)cy / ppm + min_x, yc = (float)cx / ppm + min_y;
        float* f = feat + i * 9;
        f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = p.w;
        f[4] = p.x - s.x / cnt; f[5] = p.y - s.y / cnt; f[6] = p.z - s.z / cnt;
        f[7] = p.x - xc; f[8] = p.y - yc;
    }
}

// ======================================================================================================================================
// Round 6: the same index in THREE launches, no fill, no global atomic (was: occupancy fill, keys, 2 scan launches, sums fill, gather with four 64-bit
// global atomics per point run, decorate - and three more fills in the static-shape mode).
//   pillar_slab_kernel             H1's slab form (misc.cpp): a 1024-thread block owns a SLAB of one sample's grid (<= 4096 cells: [cell][x, y, z sums, count]
//                                  as 64-bit fixed-point integers = 128 KB of LDS), walks the sample's cloud (all slab blocks of a sample on one XCD: the
//                                  cloud leaves HBM once) and accumulates its cells with return-less LDS atomics - integer adds, so the sums are the ones
//                                  the global atomics produced.  It then writes the sums of its OCCUPIED cells to a cell-indexed table (plain stores;
//                                  never read for an empty cell, so the table needs no initialisation), its words of the occupancy BITMAP (every word
//                                  written: no zeroed workspace), and - for the 1024-point chunks it owns (chunk % S == slab) - the keys and the chunk's
//                                  kept-point count.  Cells are numbered with each sample padded to S slabs of SC cells (SC % 128 == 0): the padded id
//                                  b * S * SC + x_idx * GY + y_idx orders like the (b, x_idx, y_idx) rows torch.unique sorts.
//   pillar_scan_kernel (1 block)   exclusive scans of the bitmap's per-word popcounts (16-byte loads, 4096 words per block scan) and of the chunk counts
//   pillar_gather_decorate_kernel  point blocks: row of a kept point = chunk offset + in-block scan of the keep flags, pillar id = word prefix +
//                                  popcount of the lower bits, the 9 features straight from the cell's sums - compacted cloud, inverse indices and
//                                  features in ONE pass over the points; in the static-shape mode every dropped point writes one zero row of the tail
//                                  (the d-th dropped point owns row N + d), so the capacity buffers need no fill either.
//                                  cell blocks: cellkey[rank] = cell for the set bits (+ the -1 tail in the static-shape mode)
// Integer results are what the seven-launch form produces (torch.unique's, tests/kernel_cases.py check_pillars / check_pillar_index_forms).
constexpr int kPlBlock = 1024;      // points per chunk: one block of the gather kernel (256 threads x 4 consecutive points), one trip row of the slab kernel
constexpr int kPlSlabMax = 4096;    // cells per slab (12 bits of a queue entry)
constexpr int kPlQueue = 4096;      // queued points per trip of 16 x 1024 points (14 bits of a queue entry); the expected load is 16384 / S
struct PlGeom { int S, SC, CP; };   // slabs per sample, cells per slab, padded cells per sample
inline PlGeom pl_geom(int GX, int GY) {
    const long cs = (long)GX * GY;
    const int S = cdiv(cs, kPlSlabMax), SC = (cdiv(cs, S) + 127) / 128 * 128;
    return PlGeom{S, SC, S * SC};
}
#ifdef TF_EMU
static inline int pl_popc(unsigned v) { return __builtin_popcount(v); }
static inline int wave_count(bool p) { return (int)wave_sum(p ? 1.f : 0.f); }
#else
__device__ __forceinline__ int pl_popc(unsigned v) { return __popc(v); }
__device__ __forceinline__ int wave_count(bool p) { return __popcll(__ballot(p)); }
#endif
// exclusive scan of one int per thread over the NW waves of the block (block total < 2^24: the shuffles carry fp32); sm: NW ints
template <int NW>
__device__ __forceinline__ int block_excl_scan(int v, int* sm, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x = (float)v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float y = shfl(x, (lane - off) & 63);
        if (lane >= off) x += y;
    }
    if (lane == 63) sm[wave] = (int)x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int t = sm[w];
        base += w < wave ? t : 0;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + (int)x - v;
}
__device__ __forceinline__ void pillar_accumulate(unsigned long long* c, float x, float y, float z) {      // LDS; the sums pillar_gather_kernel forms with global atomics
    atomicAdd(c, (unsigned long long)(long long)llrintf(x * kPillarFix));
    atomicAdd(c + 1, (unsigned long long)(long long)llrintf(y * kPillarFix));
    atomicAdd(c + 2, (unsigned long long)(long long)llrintf(z * kPillarFix));
    atomicAdd(c + 3, 1ull);
}
__global__ void __launch_bounds__(1024) pillar_slab_kernel(const float* __restrict__ pts, const int32_t* __restrict__ npts, int B, int Nmax, int F, int vec4,
                                                           float min_x, float max_x, float min_y, float max_y, float ppm, int GX, int GY, int S, int SC,
                                                           int32_t* __restrict__ keys, unsigned* __restrict__ bitmap, long long* __restrict__ cellsums,
                                                           int32_t* __restrict__ blockcnt, int dbg) {
    __shared__ __attribute__((aligned(16))) unsigned long long acc[kPlSlabMax * 4];
    __shared__ unsigned char occ[kPlSlabMax];
    __shared__ unsigned queue[kPlQueue];
    __shared__ int cnt[64];
    __shared__ int qn;
    const int tid = threadIdx.x;
    // blocks of a sample sit on ONE XCD where they can (the cloud leaves HBM once), but no XCD gets more than its share of the B * S blocks: a block
    // needs a whole CU's LDS, so 34 blocks on a 32-CU XCD would run in two rounds
    const int per = (B * S + 7) / 8, lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || lin >= B * S) return;
    const int b = lin / S, slab = lin - b * S;
    const int n = npts ? (npts[b] < Nmax ? npts[b] : Nmax) : Nmax;
    const int CP = S * SC, lo = slab * SC, NCH = (Nmax + kPlBlock - 1) / kPlBlock;
    const float* base = pts + (long)b * Nmax * F;
    constexpr int U = 16;
    bool first = true;
    int own = slab, ownk = 0;                      // the next 1024-point chunk this block owns (slab, slab + S, ...) and its slot in cnt
    for (int i0 = 0; i0 < n || first; i0 += U * 1024) {
        float px[U], py[U];
        if (i0 < n) {
#pragma unroll
            for (int u = 0; u < U; ++u) {          // clamped addresses: the loads are unconditional, the predicate applies to the cell
                const int i = i0 + tid + 1024 * u;
                const float* p = base + (long)(i < n ? i : n - 1) * F;
                if (vec4) { const float4 v = *reinterpret_cast<const float4*>(p); px[u] = v.x; py[u] = v.y; }
                else { px[u] = p[0]; py[u] = p[1]; }
            }
        }
        if (first) {                               // the accumulators are cleared while the first trip's points travel
            float4* a4 = reinterpret_cast<float4*>(acc);
            for (int k = tid; k < 2 * SC; k += 1024) a4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < 64) cnt[tid] = 0;
            first = false;
        }
        if (tid == 0) qn = 0;
        __syncthreads();
        // phase A: the cell of every point of the trip; the few that fall into this slab (1 / S of the kept ones) are QUEUED - accumulating them here
        // would run the fixed-point conversions and the four atomics for two or three live lanes of every wave and every u
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + tid + 1024 * u, chunk = (i0 >> 10) + u;
            int local = -1;
            if (dbg & 1) { if (px[u] == 123.f) keys[0] = 1; continue; }
            if (i < n) {                           // point_pillar.py:70-83, as pillar_keys_kernel
                const float x = px[u], y = py[u];
                if (x >= min_x && x < max_x && y >= min_y && y < max_y) {
                    const float fx = (x - min_x) * ppm, fy = (y - min_y) * ppm;
                    int cx = (int)fx, cy = (int)fy;
                    cx = cx < GX ? cx : GX - 1;
                    cy = cy < GY ? cy : GY - 1;
                    local = cx * GY + cy;
                    if (local >= lo && local < lo + SC) {
                        const int slot = atomicAdd(&qn, 1);
                        if (slot < kPlQueue) queue[slot] = ((unsigned)(tid + 1024 * u) << 12) | (unsigned)(local - lo);
                        else pillar_accumulate(acc + (long)(local - lo) * 4, x, y, base[(long)i * F + 2]);      // queue full (a cloud concentrated in one slab): in place
                    }
                }
            }
            if (!(dbg & 8) && chunk == own) {                // block-uniform: this block owns the chunk's keys and its kept-point count
                if (i < Nmax) keys[(long)b * Nmax + i] = local >= 0 ? b * CP + local : -1;
                const int c = wave_count(local >= 0);
                if ((tid & 63) == 0 && c) atomicAdd(&cnt[ownk], c);
                own += S; ++ownk;
            }
        }
        __syncthreads();
        // phase B: dense - one queued point per thread (its coordinates come back from L2)
        const int nq = qn < kPlQueue ? qn : kPlQueue;
        for (int e = tid; e < nq && !(dbg & 2); e += 1024) {
            const unsigned q = queue[e];
            const float* p = base + (long)(i0 + (int)(q >> 12)) * F;
            pillar_accumulate(acc + (long)(q & 4095u) * 4, p[0], p[1], p[2]);
        }
        __syncthreads();
    }
    for (; own < NCH; own += S)                              // owned chunks behind the sample's last point: every key is -1, the count stays 0
        if (own * kPlBlock + tid < Nmax) keys[(long)b * Nmax + own * kPlBlock + tid] = -1;
    if (dbg & 4) return;
    for (int q = tid; q < SC; q += 1024) {
        const bool on = acc[(long)q * 4 + 3] != 0ull;
        occ[q] = on ? 1 : 0;
        if (on) {
            const float4* a4 = reinterpret_cast<const float4*>(acc + (long)q * 4);
            float4* o4 = reinterpret_cast<float4*>(cellsums + ((long)b * CP + lo + q) * 4);
            o4[0] = a4[0];
            o4[1] = a4[1];
        }
    }
    __syncthreads();
    if (tid < SC / 32) {
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) w |= (unsigned)occ[tid * 32 + k] << k;
        bitmap[((long)b * CP + lo) / 32 + tid] = w;
    }
    if (tid < 64 && slab + S * tid < NCH) blockcnt[b * NCH + slab + S * tid] = cnt[tid];
}
__global__ void __launch_bounds__(1024) pillar_scan_kernel(const unsigned* __restrict__ bitmap, int nwords, const int32_t* __restrict__ blockcnt, int nblocks,
                                                           int32_t* __restrict__ wordprefix, int32_t* __restrict__ blockoff, int32_t* __restrict__ totals) {
    __shared__ int sm[16];
    const int t = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < nwords; base += 8 * 4096) {        // eight 16-byte loads in flight per thread, then eight block scans (nwords % 4 == 0)
        int4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int w = base + r * 4096 + 4 * t;
            v[r] = *reinterpret_cast<const int4*>(bitmap + (w < nwords ? w : 0));
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int w = base + r * 4096 + 4 * t;
            if (base + r * 4096 >= nwords) break;                // block-uniform
            const bool in = w < nwords;
            const int c0 = in ? pl_popc((unsigned)v[r].x) : 0, c1 = in ? pl_popc((unsigned)v[r].y) : 0, c2 = in ? pl_popc((unsigned)v[r].z) : 0,
                      c3 = in ? pl_popc((unsigned)v[r].w) : 0;
            int tt;
            const int ex = carry + block_excl_scan<16>((c0 + c1) + (c2 + c3), sm, tt);
            if (in) { int4 o; o.x = ex; o.y = ex + c0; o.z = ex + c0 + c1; o.w = ex + c0 + c1 + c2; *reinterpret_cast<int4*>(wordprefix + w) = o; }
            carry += tt;
        }
    }
    if (t == 0) totals[1] = carry;
    carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + t, v = i < nblocks ? blockcnt[i] : 0;
        int tt;
        const int ex = block_excl_scan<16>(v, sm, tt);
        if (i < nblocks) blockoff[i] = carry + ex;
        carry += tt;
    }
    if (t == 0) totals[0] = carry;
}
__global__ void __launch_bounds__(256) pillar_gather_decorate_kernel(const float* __restrict__ pts, int F, int vec4, const int32_t* __restrict__ keys, int B, int Nmax,
                                                                     const int32_t* __restrict__ blockoff, const int32_t* __restrict__ wordprefix,
                                                                     const unsigned* __restrict__ bm, const long long* __restrict__ cellsums,
                                                                     const int32_t* __restrict__ totals, int GX, int GY, int CP, float ppm, float min_x,
                                                                     float min_y, float* __restrict__ pts4, int32_t* __restrict__ inv, float* __restrict__ feat,
                                                                     int32_t* __restrict__ cellkey, long cell_cap, int fill_tail) {
    __shared__ int sm[4];
    __shared__ float lf[kPlBlock * 9];
    const int NCH = (Nmax + kPlBlock - 1) / kPlBlock, nbA = B * NCH, nwords = (int)((long)B * CP / 32);
    if ((int)blockIdx.x >= nbA) {          // cell blocks: one bitmap word (32 padded cells) per thread
        const int w = ((int)blockIdx.x - nbA) * 256 + threadIdx.x;
        if (w >= nwords) return;
        const unsigned bits = bm[w];
        if (bits == 0u && !fill_tail) return;
        int r = wordprefix[w];
        const int P = totals[1];
        const long pc0 = (long)w * 32;
        const int b = (int)(pc0 / CP), l0 = (int)(pc0 - (long)b * CP);        // CP % 32 == 0: a word never straddles two samples
        for (int k = 0; k < 32; ++k) {
            if ((bits >> k) & 1u) cellkey[r++] = b * GX * GY + l0 + k;
            else if (fill_tail) {                                            // the u-th empty (padded) cell owns tail slot P + u
                const long slot = (long)P + (pc0 + k - r);
                if (slot < cell_cap) cellkey[slot] = -1;
            }
        }
        return;
    }
    const int b = blockIdx.x / NCH, c = blockIdx.x - b * NCH;
    const int j0 = c * kPlBlock + threadIdx.x * 4;
    const long f0 = (long)b * Nmax + j0;                 // flat index of the thread's first point
    int key[4], kept = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { key[j] = j0 + j < Nmax ? keys[f0 + j] : -1; kept += key[j] >= 0; }
    float4 p[4];
    int rk[4];
    long long s[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {           // every gather of the four points before any store
        const int k = key[j] >= 0 ? key[j] : b * CP;
        const long i = j0 + j < Nmax ? f0 + j : (long)b * Nmax + Nmax - 1;
        if (vec4) p[j] = *reinterpret_cast<const float4*>(pts + i * 4);
        else p[j] = make_float4(pts[i * F], pts[i * F + 1], pts[i * F + 2], pts[i * F + 3]);
        rk[j] = wordprefix[k >> 5] + pl_popc(bm[k >> 5] & ((1u << (k & 31)) - 1u));
        if (key[j] >= 0) {
            const long long* sr = cellsums + (long)k * 4;
            s[j][0] = sr[0]; s[j][1] = sr[1]; s[j][2] = sr[2]; s[j][3] = sr[3];
        } else { s[j][0] = s[j][1] = s[j][2] = s[j][3] = 0; }
    }
    int tot;
    const int ex = block_excl_scan<4>(kept, sm, tot);
    const long row0 = (long)blockoff[blockIdx.x];        // the block's kept points are rows [row0, row0 + tot): their features leave through LDS as contiguous
    const long N = totals[0];                            // dwords (nine 4-byte stores per lane at a 36-byte stride are address-rate bound: 15.8 -> measured below)
    int r = ex;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (key[j] >= 0) {
            // decorate (:54-67) exactly as pillar_decorate_kernel: the pillar's coordinate SUMS rounded to fp32 once, quirk Q15 kept
            const double inv_fix = 1.0 / (double)kPillarFix;
            const float mx = (float)((double)s[j][0] * inv_fix), my = (float)((double)s[j][1] * inv_fix), mz = (float)((double)s[j][2] * inv_fix);
            const float cw = (float)s[j][3], cnt = cw < 1.f ? 1.f : cw;
            const int local = key[j] - b * CP;
            const int cy = local % GY, cx = local / GY;
            const float xc = (float)cy / ppm + min_x, yc = (float)cx / ppm + min_y;
            *reinterpret_cast<float4*>(pts4 + (row0 + r) * 4) = p[j];
            inv[row0 + r] = rk[j];
            float* f = lf + r * 9;
            f[0] = p[j].x; f[1] = p[j].y; f[2] = p[j].z; f[3] = p[j].w;
            f[4] = p[j].x - mx / cnt; f[5] = p[j].y - my / cnt; f[6] = p[j].z - mz / cnt;
            f[7] = p[j].x - xc; f[8] = p[j].y - yc;
            ++r;
        }
    }
    __syncthreads();
    float* fo = feat + row0 * 9;
    for (int k = threadIdx.x; k < tot * 9; k += 256) fo[k] = lf[k];
    if (fill_tail) {
        // static shapes: the block's dropped points own the tail rows [N + d0, N + d0 + nd), d0 = dropped points in front of the block
        const int npt = Nmax - c * kPlBlock < kPlBlock ? Nmax - c * kPlBlock : kPlBlock, nd = npt - tot;
        const long t0 = N + ((long)b * Nmax + (long)c * kPlBlock - row0);
        for (int k = threadIdx.x; k < nd; k += 256) { *reinterpret_cast<float4*>(pts4 + (t0 + k) * 4) = make_float4(0.f, 0.f, 0.f, 0.f); inv[t0 + k] = 0; }
        float* ft = feat + t0 * 9;
        for (int k = threadIdx.x; k < nd * 9; k += 256) ft[k] = 0.f;
    }
}

__global__ void __launch_bounds__(256) fill_i32_kernel(int32_t* __restrict__ p, long n, int32_t v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}

// scatter_max (:32) of post-ReLU features (>= 0, so the uint bit pattern orders like the float and 0 is the identity)
__global__ void __launch_bounds__(256) pillar_max_kernel(const float* __restrict__ z, const int32_t* __restrict__ inv, long N, int C,
                                                         float* __restrict__ pf) {
    const long total = N * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float v = z[i];
        if (v > 0.f) atomicMax(reinterpret_cast<unsigned*>(pf) + (long)inv[i / C] * C + (i % C), __float_as_uint(v));
    }
}

// arg of the max: the LOWEST point row attaining it (torch_scatter's CPU kernel updates on strict '>' in row order)
__global__ void __launch_bounds__(256) pillar_arg_kernel(const float* __restrict__ z, const int32_t* __restrict__ inv, const float* __restrict__ pf,
                                                         long N, int C, int32_t* __restrict__ arg) {
    const long total = N * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float v = z[i];
        const long o = (long)inv[i / C] * C + (i % C);
        if (v > 0.f && v == pf[o]) atomicMin(arg + o, (int32_t)(i / C));
    }
}

// canvas cell of a pillar after scatter_points (:94-95) AND rot90(-1) (model.py:738): row = clamp(y_idx), col = clamp(x_idx)
__device__ __forceinline__ long pillar_cell(int key, int GX, int GY, int H, int W) {
    const int cy = key % GY, cx = (key / GY) % GX, b = key / (GY * GX);
    const int row = cy < H ? cy : H - 1, col = cx < W ? cx : W - 1;
    return ((long)b * H + row) * W + col;
}

__global__ void __launch_bounds__(256) pillar_owner_kernel(const int32_t* __restrict__ cellkey, int P, int GX, int GY, int H, int W,
                                                           int32_t* __restrict__ owner) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < P; r += gridDim.x * 256) {
        const int key = cellkey[r];                                       // < 0: an unused slot of a fixed-capacity key list (static-shape mode)
        if (key >= 0) atomicMax(owner + pillar_cell(key, GX, GY, H, W), r);      // index_put: the last (highest-rank) duplicate wins
    }
}

__global__ void __launch_bounds__(256) pillar_canvas_kernel(const float* __restrict__ pf, const int32_t* __restrict__ owner, long ncell, int C,
                                                            const float* __restrict__ extra, int Ce, long HW, float* __restrict__ out) {
    const int Cs = C + Ce;
    const long total = ncell * Cs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long cell = i / Cs;
        const int c = (int)(i % Cs);
        float v;
        if (c < C) {
            const int r = owner[cell];
            v = r >= 0 ? pf[(long)r * C + c] : 0.f;
        } else {
            const long b = cell / HW, px = cell % HW;
            v = extra[(b * Ce + (c - C)) * HW + px];
        }
        out[i] = v;
    }
}

__global__ void __launch_bounds__(256) pillar_canvas_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ owner,
                                                                const int32_t* __restrict__ cellkey, const int32_t* __restrict__ inv,
                                                                const int32_t* __restrict__ arg, long N, int C, int Cs, int GX, int GY, int H, int W,
                                                                float* __restrict__ dz) {
    const long total = N * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int row = (int)(i / C), c = (int)(i % C);
        const int r = inv[row];
        float g = 0.f;
        if (arg[(long)r * C + c] == row) {
            const long cell = pillar_cell(cellkey[r], GX, GY, H, W);
            if (owner[cell] == r) g = dout[cell * Cs + c];
        }
        dz[i] = g;
    }
}

// ---- BatchNorm1d (+ ReLU) over the FIRST n rows of a fixed-capacity (rows_cap, C) matrix, n read on the DEVICE (the kept-point count of the pillar
// index, totals[0]): what lets --use_point_pillars run under a captured hipGraph (the point net's nn.BatchNorm1d, point_pillar.py:17-27, sees a
// data-dependent number of rows).  Rows >= n are IGNORED by the statistics and written as ZERO by the apply passes, so everything downstream
// (Linear, scatter-max, weight gradients) can run over the whole capacity.  C <= 256, 256 % C == 0.  Two statistics passes (sum, then centred
// squares: every block re-reduces the first pass's <= kBnrBlocks partials in its prologue) + apply = 3 launches; backward 2.  Deterministic.
constexpr int kBnrBlocks = 256;
template <int PASS>      // 1: sum x   2: sum (x - mean)^2   3 (backward): sum g, sum g * xhat with g = dz * [z > 0]
__global__ void __launch_bounds__(256) bnr_stats_kernel(const float* __restrict__ x, const float* __restrict__ dz, const float* __restrict__ z,
                                                        const int32_t* __restrict__ nrows, int rows_cap, int C, const float* __restrict__ part1,
                                                        const float* __restrict__ mean_in, const float* __restrict__ invstd_in, float* __restrict__ part) {
    __shared__ float red[2][256];
    __shared__ float mu[256];
    const int tid = threadIdx.x, c = tid % C, rl = tid / C, rpb = 256 / C;
    int n = nrows[0];
    n = n < 0 ? 0 : (n > rows_cap ? rows_cap : n);
    if (PASS == 2) {
        if (tid < C) {
            float s = 0.f;
            for (int b = 0; b < (int)gridDim.x; ++b) s += part1[(long)b * C + tid];
            mu[tid] = s / (float)(n > 0 ? n : 1);
        }
        __syncthreads();
    }
    const float m = PASS == 2 ? mu[c] : (PASS == 3 ? mean_in[c] : 0.f), is = PASS == 3 ? invstd_in[c] : 0.f;
    float a0 = 0.f, a1 = 0.f;
    for (long r = (long)blockIdx.x * rpb + rl; r < n; r += (long)gridDim.x * rpb) {
        const float v = x[r * C + c];
        if (PASS == 1) a0 += v;
        else if (PASS == 2) { const float d = v - m; a0 += d * d; }
        else { float g = dz[r * C + c]; if (!(z[r * C + c] > 0.f)) g = 0.f; a0 += g; a1 += g * ((v - m) * is); }
    }
    red[0][tid] = a0; red[1][tid] = a1;
    __syncthreads();
    if (rl == 0) {
        float t0 = 0.f, t1 = 0.f;
        for (int j = 0; j < rpb; ++j) { t0 += red[0][j * C + c]; t1 += red[1][j * C + c]; }
        if (PASS == 3) { part[((long)blockIdx.x * 2 + 0) * C + c] = t0; part[((long)blockIdx.x * 2 + 1) * C + c] = t1; }
        else part[(long)blockIdx.x * C + c] = t0;
    }
}
// y = relu(x * sc + sh) for rows < n, 0 beyond; block 0 also publishes mean / invstd and updates the running statistics (unbiased variance)
__global__ void __launch_bounds__(256) bnr_apply_kernel(const float* __restrict__ x, const int32_t* __restrict__ nrows, int rows_cap, int C, int nblk,
                                                        const float* __restrict__ part1, const float* __restrict__ part2, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                        float eps, int relu, float* __restrict__ y, float* __restrict__ save_mean,
                                                        float* __restrict__ save_invstd) {
    __shared__ float sc[256], sh[256];
    int n = nrows[0];
    n = n < 0 ? 0 : (n > rows_cap ? rows_cap : n);
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        float s = 0.f, q = 0.f;
        for (int b = 0; b < nblk; ++b) { s += part1[(long)b * C + c]; q += part2[(long)b * C + c]; }
        const float fn = (float)(n > 0 ? n : 1), mean = s / fn, var = q / fn, inv = 1.0f / sqrtf(var + eps);
        sc[c] = gamma[c] * inv; sh[c] = beta[c] - mean * (gamma[c] * inv);
        if (blockIdx.x == 0) {
            save_mean[c] = mean; save_invstd[c] = inv;
            if (rmean && n > 0) {
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * (n > 1 ? var * (fn / (fn - 1.f)) : var);
            }
        }
    }
    __syncthreads();
    const long total = (long)rows_cap * C, live = (long)n * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = x[i < live ? i : 0] * sc[c] + sh[c];
        if (relu) v = fmaxf(v, 0.f);
        y[i] = i < live ? v : 0.f;
    }
}
// dx = A g + Bc x + Cc (coefficients as bn_bwd_finalize_kernel's) for rows < n, 0 beyond; block 0 accumulates dgamma / dbeta
__global__ void __launch_bounds__(256) bnr_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ x,
                                                            const int32_t* __restrict__ nrows, int rows_cap, int C, int nblk, const float* __restrict__ part,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float cA[256], cB[256], cC[256];
    int n = nrows[0];
    n = n < 0 ? 0 : (n > rows_cap ? rows_cap : n);
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        float sg = 0.f, sgx = 0.f;
        for (int b = 0; b < nblk; ++b) { sg += part[((long)b * 2 + 0) * C + c]; sgx += part[((long)b * 2 + 1) * C + c]; }
        const float fn = (float)(n > 0 ? n : 1), A = gamma[c] * invstd[c], Bc = -A * invstd[c] * (sgx / fn);
        cA[c] = A; cB[c] = Bc; cC[c] = -A * (sg / fn) - Bc * mean[c];
        if (blockIdx.x == 0) { if (dgamma) dgamma[c] += sgx; if (dbeta) dbeta[c] += sg; }
    }
    __syncthreads();
    const long total = (long)rows_cap * C, live = (long)n * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long j = i < live ? i : 0;
        float g = dz[j];
        if (!(z[j] > 0.f)) g = 0.f;
        dx[i] = i < live ? cA[c] * g + cB[c] * x[j] + cC[c] : 0.f;
    }
}

}  // namespace

extern "C" int tf_pillar_keys_f32(const float* points, const int32_t* num_points, int B, int max_points, int point_stride, float min_x, float max_x,
                                  float min_y, float max_y, float pixels_per_meter, int GX, int GY, int32_t* keys, int32_t* keep, int32_t* occ,
                                  void* stream) {
    TF_REQUIRE(points && keys && keep && occ && B > 0 && max_points > 0 && point_stride >= 4 && GX > 0 && GY > 0 &&
                   (long)B * GX * GY < 2147483647L, "tf_pillar_keys_f32: bad arguments");
    TF_LAUNCH(fill_i32_kernel, dim3(pl_blocks((long)B * GX * GY)), dim3(256), stream, occ, (long)B * GX * GY, 0);
    TF_LAUNCH(pillar_keys_kernel, dim3(pl_blocks((long)B * max_points)), dim3(256), stream, points, num_points, B, max_points, point_stride, min_x,
              max_x, min_y, max_y, pixels_per_meter, GX, GY, keys, keep, occ);
    return launch_status("tf_pillar_keys_f32");
}

extern "C" int tf_pillar_index_scan_i32(const int32_t* keys, int64_t n_points, const int32_t* occ, int64_t ncells, int32_t* pos, int32_t* rank, int32_t* cellkey,
                                        int32_t* totals, int32_t* ws, void* stream) {
    TF_REQUIRE(keys && occ && pos && rank && cellkey && totals && ws && n_points > 0 && ncells > 0, "tf_pillar_index_scan_i32: bad arguments (ws needs (n_points + ncells) / 1024 + 2 ints)");
    const int nbA = cdiv(n_points, 1024), nbB = cdiv(ncells, 1024);
    TF_LAUNCH(scan2_block_sums_kernel, dim3(nbA + nbB), dim3(256), stream, keys, (long)n_points, occ, (long)ncells, nbA, ws);
    TF_LAUNCH(scan2_final_kernel, dim3(nbA + nbB), dim3(256), stream, keys, (long)n_points, occ, (long)ncells, nbA, nbB, (const int32_t*)ws, pos, rank, cellkey, totals);
    return launch_status("tf_pillar_index_scan_i32");
}

extern "C" int tf_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* total, int32_t* ws, void* stream) {
    TF_REQUIRE(in && out && total && ws && n > 0, "tf_exclusive_scan_i32: bad arguments (ws needs n/1024 + 1 ints)");
    const int nb = cdiv(n, 1024);
    TF_LAUNCH(scan_block_sums_kernel, dim3(nb), dim3(256), stream, in, (long)n, ws);
    TF_LAUNCH(scan_carry_kernel, dim3(1), dim3(256), stream, ws, nb, total);
    TF_LAUNCH(scan_final_kernel, dim3(nb), dim3(256), stream, in, (long)n, (const int32_t*)ws, out);
    return launch_status("tf_exclusive_scan_i32");
}

extern "C" int tf_pillar_gather_f32(const float* points, int point_stride, const int32_t* keys, const int32_t* pos, const int32_t* occ,
                                    const int32_t* rank, int64_t n_all, int64_t ncells, int P, float* pts4, int32_t* inv, int64_t* sums,
                                    int32_t* cellkey, void* stream) {
    TF_REQUIRE(points && keys && pos && rank && pts4 && inv && sums && cellkey && n_all > 0 && ncells > 0 && P >= 0,
               "tf_pillar_gather_f32: bad arguments");
    if (P == 0) return 0;
    TF_LAUNCH(fill_i32_kernel, dim3(pl_blocks((long)P * 8)), dim3(256), stream, reinterpret_cast<int32_t*>(sums), (long)P * 8, 0);
    if (occ) TF_LAUNCH(pillar_cells_kernel, dim3(pl_blocks(ncells)), dim3(256), stream, occ, rank, (long)ncells, cellkey);      // NULL: tf_pillar_index_scan_i32 already wrote the cell keys
    TF_LAUNCH(pillar_gather_kernel, dim3(pl_blocks((n_all + 3) / 4)), dim3(256), stream, points, point_stride, keys, pos, rank, (long)n_all, pts4, inv,
              reinterpret_cast<long long*>(sums));
    return launch_status("tf_pillar_gather_f32");
}

/* ---- round 6: the three-launch pillar index (kernels above) ---- */
static int pl_dbg() { static const int d = [] { const char* e = getenv("TF_PL_DBG"); return e ? atoi(e) : 0; }(); return d; }      // timing diagnosis: phases of the slab kernel off (results are garbage)
extern "C" long tf_pillar_padded_cells(int GX, int GY) { return GX > 0 && GY > 0 ? (long)pl_geom(GX, GY).CP : -1; }
extern "C" int tf_pillar_mark_f32(const float* points, const int32_t* num_points, int B, int max_points, int point_stride, float min_x, float max_x, float min_y,
                                  float max_y, float pixels_per_meter, int GX, int GY, int32_t* keys, int32_t* bitmap, int64_t* cellsums, int32_t* blockcnt,
                                  void* stream) {
    TF_REQUIRE(points && keys && bitmap && cellsums && blockcnt && B > 0 && max_points > 0 && point_stride >= 4 && GX > 0 && GY > 0 && aligned16(cellsums) &&
                   aligned16(bitmap), "tf_pillar_mark_f32: bad arguments (bitmap / cellsums 16-byte aligned)");
    const PlGeom g = pl_geom(GX, GY);
    TF_REQUIRE((long)B * g.CP < 16777216L && (long)B * max_points < 16777216L && cdiv(max_points, kPlBlock) <= 64 * g.S,
               "tf_pillar_mark_f32: padded cells and points must stay below 2^24, max_points below 65536 x slabs");
    const int vec4 = point_stride == 4 && aligned16(points);
    TF_LAUNCH(pillar_slab_kernel, dim3(8 * cdiv((long)B * g.S, 8)), dim3(1024), stream, points, num_points, B, max_points, point_stride, vec4, min_x, max_x, min_y, max_y,
              pixels_per_meter, GX, GY, g.S, g.SC, keys, reinterpret_cast<unsigned*>(bitmap), reinterpret_cast<long long*>(cellsums), blockcnt, pl_dbg());
    return launch_status("tf_pillar_mark_f32");
}
extern "C" int tf_pillar_rank_scan_i32(const int32_t* bitmap, int64_t padded_cells, const int32_t* blockcnt, int nblocks, int32_t* wordprefix, int32_t* blockoff,
                                       int32_t* totals, void* stream) {
    TF_REQUIRE(bitmap && blockcnt && wordprefix && blockoff && totals && padded_cells > 0 && padded_cells % 128 == 0 && padded_cells < 16777216L && nblocks > 0 &&
                   aligned16(bitmap) && aligned16(wordprefix), "tf_pillar_rank_scan_i32: bad arguments (padded_cells = B x tf_pillar_padded_cells(), 16-byte aligned arrays)");
    TF_LAUNCH(pillar_scan_kernel, dim3(1), dim3(1024), stream, reinterpret_cast<const unsigned*>(bitmap), (int)(padded_cells / 32), blockcnt, nblocks, wordprefix,
              blockoff, totals);
    return launch_status("tf_pillar_rank_scan_i32");
}
extern "C" int tf_pillar_gather_decorate_f32(const float* points, int point_stride, const int32_t* keys, int B, int max_points, const int32_t* blockoff,
                                             const int32_t* wordprefix, const int32_t* bitmap, const int64_t* cellsums, const int32_t* totals, int GX, int GY,
                                             float pixels_per_meter, float min_x, float min_y, float* pts4, int32_t* inv, float* feat, int32_t* cellkey,
                                             int64_t cell_cap, int fill_tail, void* stream) {
    TF_REQUIRE(points && keys && blockoff && wordprefix && bitmap && cellsums && totals && pts4 && inv && feat && cellkey && B > 0 && max_points > 0 && GX > 0 &&
                   GY > 0 && point_stride >= 4 && aligned16(pts4) && cell_cap >= 0, "tf_pillar_gather_decorate_f32: bad arguments");
    const PlGeom g = pl_geom(GX, GY);
    TF_REQUIRE(!fill_tail || cell_cap <= (long)B * g.CP, "tf_pillar_gather_decorate_f32: cell_cap exceeds the number of cells");
    const int nbA = B * cdiv(max_points, kPlBlock), nbB = cdiv((long)B * g.CP / 32, 256);
    const int vec4 = point_stride == 4 && aligned16(points);
    TF_LAUNCH(pillar_gather_decorate_kernel, dim3(nbA + nbB), dim3(256), stream, points, point_stride, vec4, keys, B, max_points, blockoff, wordprefix,
              reinterpret_cast<const unsigned*>(bitmap), reinterpret_cast<const long long*>(cellsums), totals, GX, GY, g.CP, pixels_per_meter, min_x, min_y, pts4, inv,
              feat, cellkey, (long)cell_cap, fill_tail);
    return launch_status("tf_pillar_gather_decorate_f32");
}

extern "C" int tf_pillar_decorate_f32(const float* pts4, const int32_t* inv, const int64_t* sums, const int32_t* cellkey, int64_t N, int GX, int GY,
                                      float pixels_per_meter, float min_x, float min_y, float* feat, void* stream) {
    TF_REQUIRE(pts4 && inv && sums && cellkey && feat && N >= 0, "tf_pillar_decorate_f32: bad arguments");
    if (N == 0) return 0;
    TF_LAUNCH(pillar_decorate_kernel, dim3(pl_blocks(N)), dim3(256), stream, pts4, inv, reinterpret_cast<const long long*>(sums), cellkey, (long)N, GX, GY,
              pixels_per_meter, min_x, min_y, feat);
    return launch_status("tf_pillar_decorate_f32");
}

extern "C" int tf_pillar_scatter_max_f32(const float* z, const int32_t* inv, int64_t N, int C, int P, float* pillar_feat, int32_t* arg, void* stream) {
    TF_REQUIRE(z && inv && pillar_feat && arg && N >= 0 && C > 0 && P >= 0, "tf_pillar_scatter_max_f32: bad arguments");
    if (P == 0) return 0;
    TF_LAUNCH(fill_i32_kernel, dim3(pl_blocks((long)P * C)), dim3(256), stream, reinterpret_cast<int32_t*>(pillar_feat), (long)P * C, 0);
    TF_LAUNCH(fill_i32_kernel, dim3(pl_blocks((long)P * C)), dim3(256), stream, arg, (long)P * C, 2147483647);
    if (N > 0) {
        TF_LAUNCH(pillar_max_kernel, dim3(pl_blocks(N * C)), dim3(256), stream, z, inv, (long)N, C, pillar_feat);
        TF_LAUNCH(pillar_arg_kernel, dim3(pl_blocks(N * C)), dim3(256), stream, z, inv, (const float*)pillar_feat, (long)N, C, arg);
    }
    return launch_status("tf_pillar_scatter_max_f32");
}

extern "C" int tf_pillar_canvas_f32(const float* pillar_feat, const int32_t* cellkey, int P, int C, int B, int H, int W, int GX, int GY,
                                    const float* extra_nchw, int Ce, int32_t* owner, float* out_nhwc, void* stream) {
    TF_REQUIRE(owner && out_nhwc && B > 0 && H > 0 && W > 0 && C > 0 && Ce >= 0 && (Ce == 0 || extra_nchw) && (P == 0 || (pillar_feat && cellkey)),
               "tf_pillar_canvas_f32: bad arguments");
    const long ncell = (long)B * H * W;
    TF_LAUNCH(fill_i32_kernel, dim3(pl_blocks(ncell)), dim3(256), stream, owner, ncell, -1);
    if (P > 0) TF_LAUNCH(pillar_owner_kernel, dim3(pl_blocks(P)), dim3(256), stream, cellkey, P, GX, GY, H, W, owner);
    TF_LAUNCH(pillar_canvas_kernel, dim3(pl_blocks(ncell * (C + Ce))), dim3(256), stream, pillar_feat, (const int32_t*)owner, ncell, C, extra_nchw, Ce,
              (long)H * W, out_nhwc);
    return launch_status("tf_pillar_canvas_f32");
}

extern "C" int tf_pillar_canvas_bwd_f32(const float* dout_nhwc, const int32_t* owner, const int32_t* cellkey, const int32_t* inv, const int32_t* arg,
                                        int64_t N, int C, int Cs, int GX, int GY, int H, int W, float* dz, void* stream) {
    TF_REQUIRE(dout_nhwc && owner && cellkey && inv && arg && dz && N >= 0 && C > 0 && Cs >= C, "tf_pillar_canvas_bwd_f32: bad arguments");
    if (N == 0) return 0;
    TF_LAUNCH(pillar_canvas_bwd_kernel, dim3(pl_blocks(N * C)), dim3(256), stream, dout_nhwc, owner, cellkey, inv, arg, (long)N, C, Cs, GX, GY, H, W, dz);
    return launch_status("tf_pillar_canvas_bwd_f32");
}

// BatchNorm1d + ReLU over the first *nrows_dev rows of x (rows_cap, C) - see bnr_stats_kernel.  ws: 2 * 256 * C floats (the two passes' block partials).
extern "C" long tf_bn_rows_dev_ws_floats(int C) { return 2L * kBnrBlocks * C; }
extern "C" int tf_bn_rows_dev_fwd_f32(const float* x, const int32_t* nrows_dev, int rows_cap, int C, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd, float* ws,
                                      void* stream) {
    TF_REQUIRE(x && nrows_dev && gamma && beta && y && save_mean && save_invstd && ws && rows_cap > 0 && C > 0 && C <= 256 && 256 % C == 0,
               "tf_bn_rows_dev_fwd_f32: needs C <= 256 dividing 256 (got %d)", C);
    // only the BatchNorm1d + ReLU form of DynamicPointNet exists: tf_bn_rows_dev_bwd_f32 always applies the z > 0 mask, and the static-shape front-end relies
    // on the post-ReLU zeros of the padding rows (they alias pillar 0 in the scatter-max) - a call without the ReLU would be silently wrong downstream
    TF_REQUIRE(relu != 0, "tf_bn_rows_dev_fwd_f32: relu = 0 is not implemented (the backward and the static-shape pillar scatter assume the ReLU)");
    int nblk = cdiv(rows_cap, (256 / C) * 16);
    nblk = nblk < 1 ? 1 : (nblk > kBnrBlocks ? kBnrBlocks : nblk);
    float* p1 = ws; float* p2 = ws + (long)kBnrBlocks * C;
    const float* nof = nullptr;
    TF_LAUNCH(bnr_stats_kernel<1>, dim3(nblk), dim3(256), stream, x, nof, nof, nrows_dev, rows_cap, C, nof, nof, nof, p1);
    TF_LAUNCH(bnr_stats_kernel<2>, dim3(nblk), dim3(256), stream, x, nof, nof, nrows_dev, rows_cap, C, (const float*)p1, nof, nof, p2);
    TF_LAUNCH(bnr_apply_kernel, dim3(pl_blocks((long)rows_cap * C)), dim3(256), stream, x, nrows_dev, rows_cap, C, nblk, (const float*)p1, (const float*)p2, gamma, beta,
              running_mean, running_var, momentum, eps, relu, y, save_mean, save_invstd);
    return launch_status("tf_bn_rows_dev_fwd_f32");
}
// its backward: dz = gradient of the (post-ReLU) output z; dgamma / dbeta are ACCUMULATED; dx rows >= *nrows_dev are written as zero
extern "C" int tf_bn_rows_dev_bwd_f32(const float* dz, const float* z, const float* x, const int32_t* nrows_dev, int rows_cap, int C, const float* gamma,
                                      const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    TF_REQUIRE(dz && z && x && nrows_dev && gamma && save_mean && save_invstd && dx && ws && rows_cap > 0 && C > 0 && C <= 256 && 256 % C == 0,
               "tf_bn_rows_dev_bwd_f32: needs C <= 256 dividing 256 (got %d)", C);
    int nblk = cdiv(rows_cap, (256 / C) * 16);
    nblk = nblk < 1 ? 1 : (nblk > kBnrBlocks ? kBnrBlocks : nblk);
    const float* nof = nullptr;
    TF_LAUNCH(bnr_stats_kernel<3>, dim3(nblk), dim3(256), stream, x, dz, z, nrows_dev, rows_cap, C, nof, save_mean, save_invstd, ws);
    TF_LAUNCH(bnr_bwd_apply_kernel, dim3(pl_blocks((long)rows_cap * C)), dim3(256), stream, dz, z, x, nrows_dev, rows_cap, C, nblk, (const float*)ws, gamma, save_mean,
              save_invstd, dx, dgamma, dbeta);
    return launch_status("tf_bn_rows_dev_bwd_f32");
}

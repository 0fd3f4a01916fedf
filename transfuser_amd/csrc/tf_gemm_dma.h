// Direct-to-LDS (LDS-DMA) fp32 MFMA GEMM for gfx950: the plain-operand fast path of the engine.
//
//   C[z](i, j) (op)= alpha * sum_k A[z](i, k) * B[z](k, j)  (+ bias[j]) (+ res(i, j)) (relu)        (same contract / epilogue as gemm_kernel)
//
// What differs from tf_gemm_engine.h's gemm_kernel (global -> VGPR -> 4 x ds_write_b32 per float4 -> LDS, prefetch distance 1):
//  * operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (1 KiB per wave-instruction, no staging registers, no ds_write
//    pass); out-of-range rows / ragged K tails are ZERO-FILLED by the buffer bounds check (per-lane offset forced past num_records),
//    so every tile is handled by the same branch-free loop;
//  * a STAGES-deep LDS ring with counted `s_waitcnt vmcnt(N)` (never 0 inside the loop) and ONE raw s_barrier per k-tile: STAGES-1
//    tiles stay in flight across the barriers (hipcc's __syncthreads would drain them: cdna_hip_programming.md "glds vs register staging");
//  * KC operands (k contiguous in memory: X[m][k], W[n][k]) keep their row-major order in LDS.  LDS-DMA writes lane-linearly, so the
//    bank-conflict-free layout is obtained by permuting the SOURCE chunk each lane fetches (XOR swizzle of the 16-byte chunk index
//    with the row) and reading fragments with ds_read_b128 through the same involution.  One b128 read feeds FOUR MFMA k-steps:
//    the k order inside a tile is permuted (lane half `hi` takes chunk 2q+hi), identically for A and B - a sum over k does not care;
//  * IC operands (k rows in memory: A^T, B) are copied as they are ([k][m] tiles) and read with conflict-free ds_read_b32;
//  * workgroups of 1, 2 or 4 waves; every wave owns a (32 TM) x (32 TN) accumulator block (TM x TN MFMA tiles, 4 = the sweet spot of
//    tools/probe/mfma_probe.cpp).  Single-wave workgroups need no barrier at all and give a fine tile menu (64x64 ... 96x96 ... 128x64):
//    the GPT shapes (M = 1740) lose 12-17 % to tile-count quantisation with 128-row tiles.
//
// Requirements (checked by the host dispatcher, otherwise the call stays on gemm_kernel): both operands 16-byte aligned with ld % 4 == 0,
// cols % 4 == 0, and every operand slab (incl. batch offset) < 2 GiB so a 32-bit byte offset addresses it.
#pragma once
#include "tf_gemm_engine.h"

namespace tf {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned DMA_OOB = 0x80000000u;   // per-lane byte offset beyond any num_records (< 2 GiB): the load returns / writes zeros

#ifdef TF_EMU
struct DmaSrc { const float* base; unsigned bytes; };
__forceinline__ DmaSrc dma_make_src(const float* p, unsigned bytes) { return DmaSrc{p, bytes}; }
// host emulation: the copy happens at issue time (the pipeline's waits / barriers are still executed, races are not modelled)
__forceinline__ void dma_b128(float* lds_wave_dst, const DmaSrc& s, unsigned voff) {
    float* d = lds_wave_dst + emu::cur_lane() * 4;
    if (voff >= s.bytes || voff + 16u > s.bytes) { d[0] = d[1] = d[2] = d[3] = 0.f; }
    else memcpy(d, reinterpret_cast<const char*>(s.base) + voff, 16);
}
template <int N> __forceinline__ void dma_wait() {}
__forceinline__ void lds_wait() {}
template <int NW> __forceinline__ void dma_barrier() { if (NW > 1) __syncthreads(); else emu::wave_barrier(); }
#else
typedef i32x4 DmaSrc;
__device__ __forceinline__ DmaSrc dma_make_src(const float* p, unsigned bytes) {
    // raw buffer descriptor (stride 0): base[47:0], num_records = bytes, dword3 = 0x00020000 (cdna_hip_programming.md T8)
    DmaSrc r;
    r.x = (int)(unsigned)(size_t)p; r.y = (int)(unsigned)((size_t)p >> 32) & 0xffff; r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
// one LDS-DMA piece: every lane's 16 bytes at (base + voff) land at lds_wave_dst + 16 * lane.  M0 carries the LDS destination and is
// written in the same asm statement that uses it (cdna_hip_programming.md 5.7); the load is invisible to hipcc's vmcnt bookkeeping
// (no register destination): completion is counted by dma_wait<N>().
__device__ __forceinline__ void dma_b128(float* lds_wave_dst, const DmaSrc& s, unsigned voff) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(s) : "memory");
}
template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int NW> __device__ __forceinline__ void dma_barrier() {
    if (NW > 1) asm volatile("s_barrier" ::: "memory");
}
#endif

// ---- stream-K (SK instantiations): cross-workgroup hand-over of a partial accumulator tile.  Only the words that cross workgroups are
// accessed at agent scope (relaxed atomic store / load = global_store / global_load ... sc1: written through to / read from the memory
// side, the XCDs' L2s are not coherent with each other); the producer drains its stores (vmcnt(0)) before its flag goes up.  No
// __threadfence(): on gfx950 that is a write-back + invalidate of the XCD's whole L2 (reduce.cpp, "last block finishes").
// hipcc turns the 64 hand-over addresses of a 2 x 2 accumulator block into 64 loop-invariant 64-bit registers, hoists them out of the work loop
// and spills them (a scratch-using kernel: fewer resident waves, the persistent launch degenerates into rounds); an address that passes through
// an empty asm cannot be hoisted, and the 16 accesses behind it fold their offsets into the instruction's 12-bit immediate
#ifdef TF_EMU
__forceinline__ float* sk_launder(float* p) { return p; }
#else
__device__ __forceinline__ float* sk_launder(float* p) { asm volatile("" : "+v"(p)); return p; }
#endif
#ifdef TF_EMU
__forceinline__ void sk_store(float* p, float v) { *p = v; }
__forceinline__ float sk_load(const float* p) { return *p; }
__forceinline__ void sk_drain() {}
__forceinline__ void sk_flag_set(int* f, int v) { *f = v; }
__forceinline__ void sk_flag_wait(int* f) { if (*f == 0) { fprintf(stderr, "stream-K emulation: partial tile not ready (block order)\n"); abort(); } }
#else
__device__ __forceinline__ void sk_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sk_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sk_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void sk_flag_set(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sk_flag_wait(int* f) {
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(4);
}
#endif

// swizzle of the 16-byte chunk index inside a row of CH chunks (CH = BK / 4 = 4 or 8): with it the 16 lanes of every ds_read_b128 lane
// group (MI355X_MICROARCH.md, LDS table) hit 16 distinct 16-byte slots of the 256-byte bank row.  Depends on (row mod 32) only.
template <int CH> __device__ __forceinline__ int dma_swz(int row) { return CH == 4 ? ((row >> 2) & 3) : ((row >> 1) & 7); }

template <int TM, int TN, int WAVES_M, int WAVES_N, int BK, int STAGES, bool A_KC, bool B_KC>
struct DmaCfg {
    static constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
    static constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N, CH = BK / 4;
    static constexpr int A_FL = BM * BK, B_FL = BN * BK, STAGE_FL = A_FL + B_FL;        // floats per stage
    static constexpr int NA = A_FL / 256, NB = B_FL / 256;                              // 1 KiB DMA pieces per stage
    static constexpr int DA = NA / NW, DB = NB / NW, DPW = DA + DB;                     // pieces per wave per stage
    static constexpr int LDS_BYTES = STAGES * STAGE_FL * 4;
    static_assert(BK == 16 || BK == 32, "BK");
    static_assert(NA % NW == 0 && NB % NW == 0, "every wave must issue the same number of DMA pieces per stage (vmcnt is counted)");
    static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 1) * DPW <= 60, "ring depth / vmcnt range");
    static_assert(LDS_BYTES <= 64 * 1024, "M0 LDS base is kept within 64 KiB");
};

template <int TM, int TN, int WAVES_M, int WAVES_N, int BK, int STAGES, bool A_KC, bool B_KC, bool X3, bool SK = false>
__device__ __forceinline__ void gemm_dma_tile(const PlainOp& la_in, const PlainOp& lb_in, const GemmEpi& ep, int M, int N, int K, int tiles_m,
                                              int tiles_n, int kchunk, float* smem) {
    typedef DmaCfg<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC> C;
    constexpr int NW = C::NW, BM = C::BM, BN = C::BN, CH = C::CH, DA = C::DA, DB = C::DB, DPW = C::DPW;
    constexpr int QN = BK / 8;                       // b128 fragment reads per 32-row sub-tile per stage (KC); 4 k-steps each

    PlainOp la = la_in, lb = lb_in;
    const int tid = threadIdx.x, z = blockIdx.z;
    la.set_batch(z);
    lb.set_batch(z);
    const int wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    // tile index -> (i0, j0): grouped rasterisation (identical to gemm_kernel)
    auto tile_origin = [&](int tile, int& i0, int& j0) {
        int tm, tn;
        if (ep.group_m > 1) {
            const int per = ep.group_m * tiles_n, sr = tile / per, rem = tile - sr * per;
            int gsz = tiles_m - sr * ep.group_m;
            gsz = gsz < ep.group_m ? gsz : ep.group_m;
            tn = rem / gsz; tm = sr * ep.group_m + (rem - tn * gsz);
        } else { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
        i0 = tm * BM; j0 = tn * BN;
    };
    int i0 = 0, j0 = 0, kbeg = 0, kend = 0, nkt = 0;      // the current work item: output tile origin and k range

    // ---- DMA sources.  Piece j of an operand covers LDS floats [256 j, 256 j + 256) of the stage tile; wave w issues pieces w, w + NW, ...
    //  KC tile [rows][CH chunks]:  slot = 64 j + lane -> row = slot / CH, physical chunk p = slot % CH holds LOGICAL chunk p ^ swz(row).
    //  IC tile [BK][cols]:         slot -> k row = slot / (cols / 4), column chunk = slot % (cols / 4).
    const DmaSrc sa = dma_make_src(la.p, (unsigned)(((long)(la.rows - 1) * la.ld + la.cols) * 4));
    const DmaSrc sb = dma_make_src(lb.p, (unsigned)(((long)(lb.rows - 1) * lb.ld + lb.cols) * 4));
    unsigned voff[DPW];      // byte offset of this lane's 16 bytes in k-tile 0
    int klim[DPW];           // the chunk is inside the matrix while (k-tile index * BK) < klim (INT_MIN: row / column out of range)
    const unsigned astep = (unsigned)((A_KC ? (long)BK : (long)BK * la.ld) * 4), bstep = (unsigned)((B_KC ? (long)BK : (long)BK * lb.ld) * 4);   // byte step per k-tile
    auto seek = [&]() {      // DMA source offsets of the current work item (i0, j0, kbeg, kend)
        auto setup = [&](const PlainOp& op, bool kc, int r0, int extent, int piece, unsigned& vo, int& kl) {
            const int slot = piece * 64 + lane;
            if (kc) {
                const int row = slot / CH, c = (slot % CH) ^ dma_swz<CH>(row);
                const bool ok = r0 + row < op.rows;
                vo = (unsigned)(((long)(r0 + row) * op.ld + kbeg + c * 4) * 4);
                kl = ok ? (kend - kbeg - c * 4) : (int)0x80000000;
            } else {
                const int cq_n = extent / 4, kr = slot / cq_n, cq = slot % cq_n;
                const bool ok = r0 + cq * 4 < op.cols;
                vo = (unsigned)(((long)(kbeg + kr) * op.ld + r0 + cq * 4) * 4);
                kl = ok ? (kend - kbeg - kr) : (int)0x80000000;
            }
        };
#pragma unroll
        for (int i = 0; i < DA; ++i) setup(la, A_KC, i0, BM, wave + i * NW, voff[i], klim[i]);
#pragma unroll
        for (int i = 0; i < DB; ++i) setup(lb, B_KC, j0, BN, wave + i * NW, voff[DA + i], klim[DA + i]);
        nkt = (kend - kbeg + BK - 1) / BK;
    };
    auto issue = [&](int kt, int slot) {     // k-tile kt -> ring slot (all-zero tile when kt >= nkt)
        float* base = smem + slot * C::STAGE_FL + wave * 256;
        const int kpos = kt * BK;
#pragma unroll
        for (int i = 0; i < DA; ++i) {
            const unsigned o = voff[i] + (unsigned)kt * astep;
            dma_b128(base + i * NW * 256, sa, kpos < klim[i] ? o : DMA_OOB);
        }
#pragma unroll
        for (int i = 0; i < DB; ++i) {
            const unsigned o = voff[DA + i] + (unsigned)kt * bstep;
            dma_b128(base + C::A_FL + i * NW * 256, sb, kpos < klim[DA + i] ? o : DMA_OOB);
        }
    };

    // ---- fragment addresses
    const int wm0 = (wave % WAVES_M) * 32 * TM, wn0 = (wave / WAVES_M) * 32 * TN;
    const int sw = dma_swz<CH>(l31);
    // KC: float index of (row, logical chunk c) = row * BK + ((c ^ sw) * 4);  lane reads chunk 2q + hi
    // IC: float index of (k, col) = k * EXTENT + col;                         lane reads k = 8q + 4hi + e, e = 0..3
    int a_off, b_off;
    a_off = A_KC ? (wm0 + l31) * BK : (4 * hi) * BM + wm0 + l31;
    b_off = B_KC ? (wn0 + l31) * BK : (4 * hi) * BN + wn0 + l31;

    f32x16 acc[TM][TN];

    auto compute = [&](int slot) {
        const float* As = smem + slot * C::STAGE_FL;
        const float* Bs = As + C::A_FL;
        float4 a[2][TM], b[2][TN];
        auto load = [&](int q, int s) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (A_KC) a[s][t] = *reinterpret_cast<const float4*>(As + a_off + t * 32 * BK + (((2 * q + hi) ^ sw) * 4));
                else {
                    const float* p = As + a_off + (8 * q) * BM + t * 32;
                    a[s][t] = make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (B_KC) b[s][t] = *reinterpret_cast<const float4*>(Bs + b_off + t * 32 * BK + (((2 * q + hi) ^ sw) * 4));
                else {
                    const float* p = Bs + b_off + (8 * q) * BN + t * 32;
                    b[s][t] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
                }
            }
        };
        load(0, 0);
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int s = q & 1;
            if (q + 1 < QN) load(q + 1, s ^ 1);
            TF_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) {
                        const float av = e == 0 ? a[s][t].x : e == 1 ? a[s][t].y : e == 2 ? a[s][t].z : a[s][t].w;
                        const float bv = e == 0 ? b[s][u].x : e == 1 ? b[s][u].y : e == 2 ? b[s][u].z : b[s][u].w;
                        mfma_32x32x2(av, bv, acc[t][u]);
                    }
            TF_SCHED_FENCE();
        }
    };

    // bf16-MFMA variants (ep.prec 1: rounded operands; X3 instantiation: bf16x3 split, fp32-accurate - tf_prims.h): lane half hi owns k = 8 hi .. 8 hi + 7 of every 16-deep group = logical chunks 2 hi, 2 hi + 1 of a
    // KC row (two b128 reads) or 8 k-rows of an IC tile; fp32 values are rounded to bf16 in registers, one MFMA per 32x32 tile and group.
    auto compute_bf16 = [&](int slot) {
        const float* As = smem + slot * C::STAGE_FL;
        const float* Bs = As + C::A_FL;
#pragma unroll
        for (int g = 0; g < BK / 16; ++g) {
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (A_KC) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 v = *reinterpret_cast<const float4*>(As + a_off + t * 32 * BK + (((4 * g + 2 * hi + h) ^ sw) * 4));
                        a[t][4 * h] = v.x; a[t][4 * h + 1] = v.y; a[t][4 * h + 2] = v.z; a[t][4 * h + 3] = v.w;
                    }
                } else {
                    const float* p = As + (wm0 + l31) + (16 * g + 8 * hi) * BM + t * 32;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[t][j] = p[j * BM];
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (B_KC) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 v = *reinterpret_cast<const float4*>(Bs + b_off + t * 32 * BK + (((4 * g + 2 * hi + h) ^ sw) * 4));
                        b[t][4 * h] = v.x; b[t][4 * h + 1] = v.y; b[t][4 * h + 2] = v.z; b[t][4 * h + 3] = v.w;
                    }
                } else {
                    const float* p = Bs + (wn0 + l31) + (16 * g + 8 * hi) * BN + t * 32;
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[t][j] = p[j * BN];
                }
            }
            if constexpr (X3) mfma_tiles_x3<TM, TN>(a, b, acc);
            else {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) mfma_32x32x16_lp(a[t], b[u], acc[t][u], ep.prec);
            }
        }
    };
    const bool lowp = ep.prec != 0;
    // 16-bit STORAGE operands (ep.packed16; both operands K-contiguous): the tile rows are 2 BK halves long, every b128 fragment read IS the
    // lane's 8-deep operand of one v_mfma_f32_32x32x16_{bf16,f16} (chunk 2q + hi = k 16 q + 8 hi .. + 7: natural order) - no conversion,
    // half the L2 / LDS bytes of the fp32-storage modes per MFMA flop
    auto compute16 = [&](int slot) {
        if constexpr (A_KC && B_KC) {
            const float* As = smem + slot * C::STAGE_FL;
            const float* Bs = As + C::A_FL;
            const bool f16 = ep.packed16 == 2;
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const float4*>(As + a_off + t * 32 * BK + (((2 * q + hi) ^ sw) * 4));
#pragma unroll
                for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + b_off + t * 32 * BK + (((2 * q + hi) ^ sw) * 4));
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u) mfma_packed16(a[t], b[u], acc[t][u], f16);
            }
        }
    };

    // ---- pipeline over the current item's k-tiles: STAGES-1 tiles in flight; iteration t: wait for tile t (counted), barrier (also frees
    // slot (t-1) % STAGES for everyone), request tile t + STAGES - 1 into that slot, multiply tile t.
    auto run = [&]() {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        seek();
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) issue(s, s);
        int cur = 0, nxt = STAGES - 1;
        for (int kt = 0; kt < nkt; ++kt) {
            dma_wait<(STAGES - 2) * DPW>();
            lds_wait();
            dma_barrier<NW>();
            issue(kt + STAGES - 1, nxt);
            if constexpr (X3) compute_bf16(cur);
            else if constexpr (A_KC && B_KC) { if (ep.packed16) compute16(cur); else if (lowp) compute_bf16(cur); else compute(cur); }
            else { if (lowp) compute_bf16(cur); else compute(cur); }
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        }
        dma_wait<0>();      // drain the (all-zero) tail requests before the epilogue's own loads / the next item / the end of the block
    };

    if constexpr (!SK) {
        // XCD-aware bijective block -> tile map: the 8 XCDs (private L2 each) get contiguous tile ranges
        int tile;
        {
            const int nt = tiles_m * tiles_n, bid = blockIdx.x;
            const int q = nt >> 3, r = nt & 7, xcd = bid & 7, loc = bid >> 3;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        }
        tile_origin(tile, i0, j0);
        kbeg = blockIdx.y * kchunk;
        kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
        run();
        GemmEpi epo = ep;
        epo.C += (long)blockIdx.y * ep.sk_stride;       // two-pass split-K: this k-slice's partial tile (sk_stride = 0 otherwise)
        gemm_epilogue<TM, TN>(acc, epo, M, N, i0, j0, BM, BN, wm0, wn0, z);
    } else {
        // ---- stream-K: the launch has P persistent workgroups (all resident: P <= CUs x blocks per CU).  XCD x (workgroups with bid % 8 == x)
        // owns the same contiguous tile range as in the data-parallel launch; its tiles x nktT (tile, k-tile) units are laid out tile-major and
        // its workgroup `loc` takes the contiguous range [loc U / Pb, (loc + 1) U / Pb) - every CU gets the same number of MFMA steps whatever
        // the tile count mod the slot count is (the data-parallel launch of the GPT-4 shapes runs 672 tiles on 512 slots: a full round plus a
        // 31 % one).  The host guarantees U / Pb >= nktT, so a tile is cut into at most two parts: its k-HEAD is the LAST item of workgroup loc,
        // its k-TAIL the FIRST item of workgroup loc + 1 OF THE SAME XCD.  The tail's accumulators go to the scratch slot of that workgroup
        // right at the start of its life, with plain stores: both workgroups sit behind the same L2, the hand-over never leaves the XCD
        // (a first version exchanged them at agent scope across XCDs - 67 MB of write-through / L2-bypassing traffic per launch - and
        // gained nothing).  The head's workgroup, which reaches the tile at the very end of its range, adds them (fixed order: head +
        // tail, bitwise reproducible) and runs the normal epilogue.  Only the flag is an agent-scope word; it is all but always already up
        // and is reset by its reader for the next launch.
        const int P = gridDim.x, bid = blockIdx.x, Pb = P >> 3, xcd = bid & 7;
#ifdef TF_EMU
        const int loc = Pb - 1 - (bid >> 3);            // the emulator runs the blocks in ascending order: producers (loc + 1) first
#else
        const int loc = bid >> 3;
#endif
        const int nktT = (K + BK - 1) / BK;
        int t_first, t_count;                            // this XCD's tiles (the bijective map of the data-parallel launch)
        {
            const int nt = tiles_m * tiles_n, q = nt >> 3, r = nt & 7;
            t_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            t_count = xcd < r ? q + 1 : q;
        }
        const long U = (long)t_count * nktT;
        long u = (long)loc * U / Pb;
        const long u1 = (long)(loc + 1) * U / Pb;
        constexpr int SLOT = NW * TM * TN * 16 * 64;      // floats per partial tile: [wave][t][u][r][lane]
        const int pos = xcd * Pb + loc;                   // scratch slot / flag of this workgroup; its consumer is (xcd, loc - 1), its producer (xcd, loc + 1)
        float* part = ep.sk_ws + (long)wave * (TM * TN * 16 * 64) + lane;
        while (u < u1) {
            const int trel = (int)(u / nktT), kt0 = (int)(u - (long)trel * nktT), tile = t_first + trel;
            const long left = u1 - u;
            const int kt1 = kt0 + left < nktT ? (int)(kt0 + left) : nktT;
            tile_origin(tile, i0, j0);
            kbeg = kt0 * BK;
            kend = kt1 * BK < K ? kt1 * BK : K;
            lds_wait();
            dma_barrier<NW>();                          // every wave is done reading the previous item's last tiles before the ring is refilled
            run();
            if (kt0 != 0) {                             // k-tail of a tile whose head belongs to workgroup pos - 1: hand the accumulators over
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int v = 0; v < TN; ++v) {
                        volatile float* dst = sk_launder(part + (long)pos * SLOT + (t * TN + v) * 1024);      // 16 stores = one base + 12-bit immediates
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[r * 64] = acc[t][v][r];
                    }
                sk_drain();
                dma_barrier<NW>();
                if (tid == 0) sk_flag_set(ep.sk_flags + pos, 1);
            } else {
                if (kt1 != nktT) {                      // k-head: the tail was computed by workgroup pos + 1 as its first item
                    if (tid == 0) sk_flag_wait(ep.sk_flags + pos + 1);
                    dma_barrier<NW>();
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int v = 0; v < TN; ++v) {
                            const volatile float* src = sk_launder(part + (long)(pos + 1) * SLOT + (t * TN + v) * 1024);
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[t][v][r] += src[r * 64];
                        }
                    dma_barrier<NW>();
                    if (tid == 0) sk_flag_set(ep.sk_flags + pos + 1, 0);
                }
                gemm_epilogue<TM, TN>(acc, ep, M, N, i0, j0, BM, BN, wm0, wn0, z);
            }
            u += kt1 - kt0;
        }
    }
}

// X3 = the bf16x3-split instantiation (tf_set_precision(2)): a separate kernel, so the fp32 / bf16 binary keeps its register allocation
template <int TM, int TN, int WAVES_M, int WAVES_N, int BK, int STAGES, bool A_KC, bool B_KC, int OCC, bool X3 = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, OCC)
gemm_dma_kernel(PlainOp la, PlainOp lb, GemmEpi ep, int M, int N, int K, int tiles_m, int tiles_n, int kchunk) {
    typedef DmaCfg<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC> C;
    __shared__ __attribute__((aligned(1024))) float smem[STAGES * C::STAGE_FL];
    gemm_dma_tile<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC, X3>(la, lb, ep, M, N, K, tiles_m, tiles_n, kchunk, smem);
}
// the stream-K instantiation of the same configuration (persistent workgroups, see gemm_dma_tile): its own kernel, so the data-parallel one
// keeps its code size and register allocation
template <int TM, int TN, int WAVES_M, int WAVES_N, int BK, int STAGES, bool A_KC, bool B_KC, int OCC, bool X3 = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, OCC)
gemm_dma_sk_kernel(PlainOp la, PlainOp lb, GemmEpi ep, int M, int N, int K, int tiles_m, int tiles_n) {
    typedef DmaCfg<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC> C;
    __shared__ __attribute__((aligned(1024))) float smem[STAGES * C::STAGE_FL];
    gemm_dma_tile<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC, X3, true>(la, lb, ep, M, N, K, tiles_m, tiles_n, 0, smem);
}

// can this problem run on the DMA kernels?  (vector-aligned plain operands, 32-bit addressable)
inline bool dma_eligible_impl(const PlainOp& a, const PlainOp& b) {
    auto ok = [](const PlainOp& o) {
        return o.vec && o.rows > 0 && o.cols > 0 && ((long)(o.rows - 1) * o.ld + o.cols) * 4 < 0x7ffffff0L;
    };
    return ok(a) && ok(b);
}

template <int TM, int TN, int WAVES_M, int WAVES_N, int BK, int STAGES, bool A_KC, bool B_KC, int OCC>
inline void launch_dma_cfg(const PlainOp& la, const PlainOp& lb, const GemmEpi& ep, int M, int N, int K, int batch, int splitk, void* stream) {
    typedef DmaCfg<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC> C;
    const int tiles_m = cdiv(M, C::BM), tiles_n = cdiv(N, C::BN);
    if (splitk == kStreamK) {                       // eligibility was checked by launch_gemm (streamk_blocks > 0)
        const int P = streamk_blocks(ep, M, N, K, batch, C::BM, C::BN, BK, C::NW, OCC, C::LDS_BYTES);
        streamk_count(1);
        GemmEpi eps = ep;
        eps.prec = ep.packed16 ? 0 : gemm_precision();
        {
            long panel = (long)C::BM * K * 4;
            int g = (int)((2L << 20) / (panel > 0 ? panel : 1));
            if (g > 8) g = 8;
            if (g > tiles_m) g = tiles_m;
            eps.group_m = (g >= 2 && tiles_n >= 4) ? g : 1;
        }
        if (eps.stat_nparts) *eps.stat_nparts = eps.stat ? cdiv(M, 32 * TM) : 0;
        if (eps.prec == 2 && !ep.packed16)
            TF_LAUNCH((gemm_dma_sk_kernel<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC, OCC, true>), dim3(P), dim3(C::NT), stream, la, lb, eps, M, N, K, tiles_m, tiles_n);
        else
            TF_LAUNCH((gemm_dma_sk_kernel<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC, OCC, false>), dim3(P), dim3(C::NT), stream, la, lb, eps, M, N, K, tiles_m, tiles_n);
        return;
    }
    const bool twopass = splitk >= kTwoPass;       // eligibility was checked by launch_gemm (twopass_ok)
    if (twopass) splitk -= kTwoPass;
    int kchunk = cdiv(cdiv(K, splitk), BK) * BK;
    if (kchunk < BK) kchunk = BK;
    const int nsplit = cdiv(K, kchunk);
    dim3 grid(tiles_m * tiles_n, nsplit > 0 ? nsplit : 1, batch);
    GemmEpi epg = ep;
    epg.prec = ep.packed16 ? 0 : gemm_precision();
    const int ldws = twopass_ldws(N);
    if (twopass) {      // slices store raw partial sums [M][ldws] into the caller's scratch; the epilogue proper runs in the fix-up pass
        epg.C = ep.sk_ws; epg.ldc = ldws; epg.ldcj = 1; epg.sc_outer = epg.sc_inner = 0; epg.inner = 1; epg.bias = nullptr; epg.res = nullptr;
        epg.mask = nullptr; epg.alpha = 1.f; epg.relu = 0; epg.mode = 0; epg.sk_stride = (long)M * ldws;
        epg.drop_seed = nullptr;
    }
    {
        long panel = (long)C::BM * (kchunk < K ? kchunk : K) * 4;
        int g = (int)((2L << 20) / (panel > 0 ? panel : 1));
        if (g > 8) g = 8;
        if (g > tiles_m) g = tiles_m;
        epg.group_m = (g >= 2 && tiles_n >= 4) ? g : 1;
    }
    if (twopass || nsplit > 1) epg.stat = nullptr;          // (launch_gemm never sends a statistics request down a k-split plan)
    if (epg.stat_nparts) *epg.stat_nparts = epg.stat ? cdiv(M, 32 * TM) : 0;
    if (epg.prec == 2 && !ep.packed16)
        TF_LAUNCH((gemm_dma_kernel<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC, OCC, true>), grid, dim3(C::NT), stream, la, lb, epg, M, N, K,
                  tiles_m, tiles_n, kchunk);
    else
        TF_LAUNCH((gemm_dma_kernel<TM, TN, WAVES_M, WAVES_N, BK, STAGES, A_KC, B_KC, OCC, false>), grid, dim3(C::NT), stream, la, lb, epg, M, N, K,
                  tiles_m, tiles_n, kchunk);
    if (twopass) launch_splitk_fixup(ep.sk_ws, nsplit, (long)M * ldws, ldws, ep, M, N, stream);
}

}  // namespace tf

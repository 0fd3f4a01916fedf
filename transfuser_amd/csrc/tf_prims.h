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

// a value every lane of the wave agrees on (wave index, a row pointer's row): keeps it - and the addresses derived from it - in SGPRs
#ifdef TF_EMU
__forceinline__ int wave_uniform(int v) { return v; }
#else
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
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
// bf16x3 split ("f32x3" precision): x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (both residuals are exact in fp32,
// |x - (h + m + l)| <= 2^-27 |x|); a product keeps the six terms of weight >= 2^-18: hh + hm + mh + mm + hl + lh (dropped: <= 2^-26 |xy|).
struct Bf16x3 { float h[8], m[8], l[8]; };
__forceinline__ Bf16x3 split_bf16x3(const float (&a)[8]) {
    Bf16x3 f;
    for (int j = 0; j < 8; ++j) {
        f.h[j] = emu_bf16_round(a[j]);
        const float r1 = a[j] - f.h[j];
        f.m[j] = emu_bf16_round(r1);
        f.l[j] = emu_bf16_round(r1 - f.m[j]);
    }
    return f;
}
__forceinline__ void mfma_bf16_raw(const float (&a)[8], const float (&b)[8], f32x16& acc) { mfma_32x32x16_bf16(a, b, acc); }   // operands are bf16 values: re-rounding is the identity
__forceinline__ uint32_t emu_bf16_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u >> 16; }
__forceinline__ float emu_bf16_from_bits(uint32_t b) { uint32_t u = b << 16; float f; memcpy(&f, &u, 4); return f; }
// IEEE half <-> float (software, round to nearest even) for the fp16 compute / storage mode (tf_set_precision(3))
__forceinline__ uint16_t emu_f16_bits(float f) { const _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }
__forceinline__ float emu_f16_from_bits(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
__forceinline__ float emu_f16_round(float f) { return emu_f16_from_bits(emu_f16_bits(f)); }
// exact products of the given (already 16-bit representable) operand values, fp32 accumulation in k order: the common tail of the 16-deep MFMAs
__forceinline__ void emu_mfma16_values(const float (&a)[8], const float (&b)[8], f32x16& acc) {
    float* s = emu::wave_scratch();
    const int l = lane_id(), jc = l & 31, hi = l >> 5;
    for (int j = 0; j < 8; ++j) {
        s[l] = a[j];
        s[64 + l] = b[j];
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
// v_mfma_f32_32x32x16_f16 with fp32 inputs rounded to half (RNE); layouts as mfma_32x32x16_bf16
__forceinline__ void mfma_32x32x16_f16(const float (&a)[8], const float (&b)[8], f32x16& acc) {
    float x[8], y[8];
    for (int j = 0; j < 8; ++j) { x[j] = emu_f16_round(a[j]); y[j] = emu_f16_round(b[j]); }
    emu_mfma16_values(x, y, acc);
}
// PACKED 16-bit operands (16-bit STORAGE path): a float4 carries the lane's 8 consecutive k elements, element 0 in the low half of .x
__forceinline__ void mfma_packed16(const float4& a, const float4& b, f32x16& acc, bool f16) {
    uint32_t wa[4], wb[4];
    memcpy(wa, &a, 16); memcpy(wb, &b, 16);
    float x[8], y[8];
    for (int d = 0; d < 4; ++d) {
        if (f16) {
            x[2 * d] = emu_f16_from_bits((uint16_t)(wa[d] & 0xffffu)); x[2 * d + 1] = emu_f16_from_bits((uint16_t)(wa[d] >> 16));
            y[2 * d] = emu_f16_from_bits((uint16_t)(wb[d] & 0xffffu)); y[2 * d + 1] = emu_f16_from_bits((uint16_t)(wb[d] >> 16));
        } else {
            x[2 * d] = emu_bf16_from_bits(wa[d] & 0xffffu); x[2 * d + 1] = emu_bf16_from_bits(wa[d] >> 16);
            y[2 * d] = emu_bf16_from_bits(wb[d] & 0xffffu); y[2 * d + 1] = emu_bf16_from_bits(wb[d] >> 16);
        }
    }
    emu_mfma16_values(x, y, acc);
}
__forceinline__ uint16_t cvt16_bits(float f, bool f16) { return f16 ? emu_f16_bits(f) : (uint16_t)emu_bf16_bits(emu_bf16_round(f)); }
// packed bf16 planes (pre-split operands kept in LDS): one dword = two consecutive k elements, element 0 in the low half
__forceinline__ void split_pair_bf16x3(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    const float ha = emu_bf16_round(a), hb = emu_bf16_round(b);
    const float ra = a - ha, rb = b - hb;
    const float ma = emu_bf16_round(ra), mb = emu_bf16_round(rb);
    const float la = emu_bf16_round(ra - ma), lb = emu_bf16_round(rb - mb);
    h = emu_bf16_bits(ha) | (emu_bf16_bits(hb) << 16);
    m = emu_bf16_bits(ma) | (emu_bf16_bits(mb) << 16);
    l = emu_bf16_bits(la) | (emu_bf16_bits(lb) << 16);
}
__forceinline__ Bf16x3 frag_from_planes(const uint32_t* h, const uint32_t* m, const uint32_t* l) {   // 4 dwords (8 elements) per plane
    Bf16x3 f;
    for (int d = 0; d < 4; ++d) {
        f.h[2 * d] = emu_bf16_from_bits(h[d] & 0xffffu); f.h[2 * d + 1] = emu_bf16_from_bits(h[d] >> 16);
        f.m[2 * d] = emu_bf16_from_bits(m[d] & 0xffffu); f.m[2 * d + 1] = emu_bf16_from_bits(m[d] >> 16);
        f.l[2 * d] = emu_bf16_from_bits(l[d] & 0xffffu); f.l[2 * d + 1] = emu_bf16_from_bits(l[d] >> 16);
    }
    return f;
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
// bf16x3 split ("f32x3" precision, tf_set_precision(2)): fp32-accurate contraction on the bf16 MFMA pipe (16x the fp32 MFMA rate on gfx950).
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round-to-nearest-even (both residuals are exact in fp32,
// |x - (h + m + l)| <= 2^-27 |x|); a product keeps the six terms of weight >= 2^-18: hh + hm + mh + mm + hl + lh (dropped: <= 2^-26 |xy|,
// below the fp32 half-ulp), accumulated in fp32 by the MFMA.  v_cvt_pk_bf16_f32 + shift/and + v_pk_add_f32: 4.5 VALU ops per element.
struct Bf16x3 { bf16x8 h, m, l; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t bf16_pack_rne(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }   // one v_cvt_pk_bf16_f32
__device__ __forceinline__ f32x2 bf16_unpack(uint32_t p) {
    f32x2 r;
    r.x = __builtin_bit_cast(float, p << 16);
    r.y = __builtin_bit_cast(float, p & 0xffff0000u);
    return r;
}
__device__ __forceinline__ Bf16x3 split_bf16x3(const float (&a)[8]) {
    u32x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 x;
        x.x = a[2 * i]; x.y = a[2 * i + 1];
        h[i] = bf16_pack_rne(x);
        const f32x2 r1 = x - bf16_unpack(h[i]);
        m[i] = bf16_pack_rne(r1);
        const f32x2 r2 = r1 - bf16_unpack(m[i]);
        l[i] = bf16_pack_rne(r2);
    }
    Bf16x3 f;
    f.h = __builtin_bit_cast(bf16x8, h);
    f.m = __builtin_bit_cast(bf16x8, m);
    f.l = __builtin_bit_cast(bf16x8, l);
    return f;
}
// packed bf16 planes (pre-split operands kept in LDS): one dword = two consecutive k elements, element 0 in the low half
__device__ __forceinline__ void split_pair_bf16x3(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    f32x2 x;
    x.x = a; x.y = b;
    h = bf16_pack_rne(x);
    const f32x2 r1 = x - bf16_unpack(h);
    m = bf16_pack_rne(r1);
    const f32x2 r2 = r1 - bf16_unpack(m);
    l = bf16_pack_rne(r2);
}
__device__ __forceinline__ Bf16x3 frag_from_planes(const uint32_t* h, const uint32_t* m, const uint32_t* l) {   // 16-byte aligned: one ds_read_b128 per plane
    Bf16x3 f;
    f.h = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(h));
    f.m = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(m));
    f.l = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(l));
    return f;
}
__device__ __forceinline__ void mfma_bf16_raw(const bf16x8& a, const bf16x8& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
// fp16 compute mode (tf_set_precision(3)): fp32 operands rounded to IEEE half in registers (v_cvt_f16_f32 / pack, RNE) -> v_mfma_f32_32x32x16_f16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_32x32x16_f16(const float (&a)[8], const float (&b)[8], f32x16& acc) {
    f32x8 x, y;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = a[j]; y[j] = b[j]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_convertvector(x, f16x8), __builtin_convertvector(y, f16x8), acc, 0, 0, 0);
}
// PACKED 16-bit operands (16-bit STORAGE path): a float4 (one ds_read_b128) carries the lane's 8 consecutive k elements
__device__ __forceinline__ void mfma_packed16(const float4& a, const float4& b, f32x16& acc, bool f16) {
    if (f16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ uint16_t cvt16_bits(float f, bool f16) {
    if (f16) return __builtin_bit_cast(uint16_t, (_Float16)f);
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
#endif

// 16-deep low-precision MFMA of the compute modes: prec 3 = IEEE half operands, anything else = bf16 (block-uniform branch)
__device__ __forceinline__ void mfma_32x32x16_lp(const float (&a)[8], const float (&b)[8], f32x16& acc, int prec) {
    if (prec == 3) mfma_32x32x16_f16(a, b, acc); else mfma_32x32x16_bf16(a, b, acc);
}

// TM x TN tiles of one 16-deep k group in bf16x3-split precision: fragments split once, six bf16 MFMAs per tile issued term-major so that
// consecutive MFMAs hit different accumulators (smallest terms first).
template <int TM, int TN>
__device__ __forceinline__ void mfma_tiles_x3(const float (&a)[TM][8], const float (&b)[TN][8], f32x16 (&acc)[TM][TN]) {
    Bf16x3 fa[TM], fb[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fa[t] = split_bf16x3(a[t]);
#pragma unroll
    for (int u = 0; u < TN; ++u) fb[u] = split_bf16x3(b[u]);
#define TF_X3_TERM(P, Q)                                                       \
    _Pragma("unroll") for (int t = 0; t < TM; ++t)                             \
        _Pragma("unroll") for (int u = 0; u < TN; ++u) mfma_bf16_raw(fa[t].P, fb[u].Q, acc[t][u]);
    TF_X3_TERM(l, h) TF_X3_TERM(h, l) TF_X3_TERM(m, m) TF_X3_TERM(m, h) TF_X3_TERM(h, m) TF_X3_TERM(h, h)
#undef TF_X3_TERM
}

// single-tile forms for the direct convolution kernels: both operands split in registers / a pre-split A fragment shared by several tiles
__device__ __forceinline__ void mfma_x3_presplit(const Bf16x3& fa, const Bf16x3& fb, f32x16& acc) {
    mfma_bf16_raw(fa.l, fb.h, acc); mfma_bf16_raw(fa.h, fb.l, acc); mfma_bf16_raw(fa.m, fb.m, acc);
    mfma_bf16_raw(fa.m, fb.h, acc); mfma_bf16_raw(fa.h, fb.m, acc); mfma_bf16_raw(fa.h, fb.h, acc);
}
__device__ __forceinline__ void mfma_32x32x16_x3(const float (&a)[8], const float (&b)[8], f32x16& acc) {
    mfma_x3_presplit(split_bf16x3(a), split_bf16x3(b), acc);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
// Sum over the 32 lanes of a wave HALF (the lanes that share lane >> 5), delivered to every lane of the half.  On the device the first four butterfly
// steps are DPP adds (quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror: VALU, no trip through the LDS crossbar) and only the 16-lane
// exchange is a ds_bpermute; after steps 1..k every group of 2^k lanes is uniform, so the mirrored partner of a lane carries exactly what its xor
// partner carries and the result is BIT-IDENTICAL to the shfl_xor(1, 2, 4, 8, 16) butterfly the emulator runs.
#ifdef TF_EMU
__forceinline__ float half_sum(float v) {
    for (int m = 1; m <= 16; m <<= 1) v += shfl_xor(v, m);
    return v;
}
#else
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_mov<0xB1>(v);       // quad_perm [1, 0, 3, 2]
    v += dpp_mov<0x4E>(v);       // quad_perm [2, 3, 0, 1]
    v += dpp_mov<0x141>(v);      // row_half_mirror
    v += dpp_mov<0x140>(v);      // row_mirror
    v += shfl_xor(v, 16);
    return v;
}
#endif
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

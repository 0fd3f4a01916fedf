// 16-bit operand copies for the packed-16 GEMM path (tf_gemm16_nt_f32; BASELINE configs[2] "bf16" and configs[4] "fp16 MFMA"):
// fp32 activations / weights / gradients are written ONCE as bf16 or IEEE-half matrices - row-major and / or TRANSPOSED - so that every
// contraction of a linear layer is an "NT" GEMM with both operands K-contiguous:
//   y  = x W^T          : x16  [M][K]   . W16  [N][K]
//   dx = dy W           : dy16 [M][N]   . W16T [K][N]
//   dW = dy^T x         : dy16T [N][M]  . x16T [K][M]       (contraction over the M rows: the transposed copies, zero-padded to M % 8 == 0)
// One pass over the fp32 source produces both copies (64 x 64 tiles, the transpose goes through LDS, 16-byte stores on both sides).
// Round to nearest even in both formats; the reference trains fp32 (config.py:55) - the 16-bit modes are this framework's own.
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int TS = 64;            // tile side
constexpr int TP = TS + 8;        // LDS row pitch in halves (16-byte aligned rows, bank spread)

// y16[r][c] (ld ldy) and / or y16t[c][r] (ld ldyt) from x[r][c] (ld ldx); pad columns of y16 (cols .. ldy) and pad rows of y16t (rows .. rows8) are zeroed
__device__ __forceinline__ void cast16_tile(const float* __restrict__ x, int rows, int cols, long ldx, uint16_t* __restrict__ y, long ldy,
                                            uint16_t* __restrict__ yt, long ldyt, int f16, int vec, int bx, int by, uint16_t* T) {
    const int tid = threadIdx.x, r0 = by * TS, c0 = bx * TS;
    const int c4 = (tid & 15) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 16 + (tid >> 4), r = r0 + row, c = c0 + c4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (vec && c + 3 < cols) { const float4 q = *reinterpret_cast<const float4*>(x + (long)r * ldx + c); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] = x[(long)r * ldx + c + e];
            }
        }
        uint16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = cvt16_bits(v[e], f16 != 0);
        if (y && r < rows) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < ldy) y[(long)r * ldy + c + e] = (c + e < cols) ? h[e] : (uint16_t)0;
        }
        if (yt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) T[(c4 + e) * TP + row] = h[e];
        }
    }
    if (!yt) return;
    __syncthreads();
    const int rows8 = (rows + 7) & ~7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int id = p * 256 + tid, col = id >> 3, rc = (id & 7) * 8;
        if (c0 + col < cols && r0 + rc < rows8) {
            const float4 q = *reinterpret_cast<const float4*>(T + col * TP + rc);      // 8 halves = 16 bytes
            *reinterpret_cast<float4*>(yt + (long)(c0 + col) * ldyt + r0 + rc) = q;
        }
    }
}
__global__ void __launch_bounds__(256) cast16_kernel(const float* __restrict__ x, int rows, int cols, long ldx, uint16_t* __restrict__ y, long ldy,
                                                     uint16_t* __restrict__ yt, long ldyt, int f16, int vec) {
    __shared__ __attribute__((aligned(16))) uint16_t T[TS * TP];
    cast16_tile(x, rows, cols, ldx, y, ldy, yt, ldyt, f16, vec, blockIdx.x, blockIdx.y, T);
}
// Many matrices in ONE launch (round 5): the 16-bit copies of every cached linear weight after AdamW were one launch per weight (64 per step in the
// TransFuser configuration).  Block b works on tile b - items[i].tile0 of the item whose tile range holds b (the table is sorted by tile0).
__global__ void __launch_bounds__(256) cast16_multi_kernel(const tf_cast16_item* __restrict__ items, int n_items, int f16) {
    __shared__ __attribute__((aligned(16))) uint16_t T[TS * TP];
    const int b = blockIdx.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {                          // last item with tile0 <= b (block-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const tf_cast16_item it = items[lo];
    const int tx = (it.cols + TS - 1) / TS, t = b - it.tile0;
    if (t >= tx * ((it.rows + TS - 1) / TS)) return;
    const int vec = ((((uintptr_t)it.x) & 15) == 0 && it.ldx % 4 == 0) ? 1 : 0;
    cast16_tile(it.x, it.rows, it.cols, (long)it.ldx, (uint16_t*)it.y16, (long)it.ldy, (uint16_t*)it.y16t, (long)it.ldyt, f16, vec, t % tx, t / tx, T);
}

// dst[b][2 i][2 j][:] += src[b][i][j][:]  (input gradient of a 1x1 / stride-2 convolution = a plain GEMM + this scatter)
__global__ void __launch_bounds__(256) add_strided2_kernel(const float* __restrict__ src, float* __restrict__ dst, int Ho, int Wo, int C4, int Hi, int Wi, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long p = i / C4;
        const int j = (int)(p % Wo); p /= Wo;
        const int ii = (int)(p % Ho);
        const long b = p / Ho;
        float4* d = reinterpret_cast<float4*>(dst) + ((b * Hi + 2 * ii) * Wi + 2 * j) * C4 + c;
        const float4 s = reinterpret_cast<const float4*>(src)[i];
        float4 v = *d;
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        *d = v;
    }
}

// cols[(b, h, w)][(kh * 3 + kw) * C + c] = x[b][h + kh - 1][w + kw - 1][c] (zero outside): the im2col matrix of a dense 3x3 / stride 1 / pad 1 convolution,
// K ordered like the (Cout, kh, kw, Cin) weight.  For the FEW-ROW, DEEP-K convolutions at the head of the decoders (SegDecoder / DepthDecoder
// deconv1, transfuser.py:221-225: 512 -> 128 at 8 x 22, 1760 output rows = 56 tiles with K = 4608: 178 us at 11.6 TFLOP/s through the implicit
// GEMM, one k-chain per tile) the product then runs as a PLAIN GEMM whose deterministic two-pass split-K fills the chip.
__global__ void __launch_bounds__(256) im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W, int C) {
    const int c4n = C >> 2;
    const long total = (long)B * H * W * 9 * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        long r = i / c4n;
        const int tap = (int)(r % 9); r /= 9;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
        const bool ok = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh), wc = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
        const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * H + hc) * W + wc) * C + 4 * c4);
        *reinterpret_cast<float4*>(cols + i * 4) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

}  // namespace

extern "C" int tf_cast16_f32(const float* x, int rows, int cols, int ldx, void* y16, int ldy, void* y16t, int ldyt, int dtype, void* stream) {
    TF_REQUIRE(x && rows > 0 && cols > 0 && ldx >= cols && (y16 || y16t) && (dtype == 1 || dtype == 2), "tf_cast16_f32: bad arguments (dtype 1 = bf16, 2 = fp16)");
    TF_REQUIRE(!y16 || ldy >= cols, "tf_cast16_f32: ldy < cols");
    TF_REQUIRE(!y16t || (ldyt >= ((rows + 7) & ~7) && ldyt % 8 == 0 && aligned16(y16t)), "tf_cast16_f32: the transposed copy needs ldyt %% 8 == 0, ldyt >= rows rounded up to 8, 16-byte aligned");
    const int vec = (aligned16(x) && ldx % 4 == 0) ? 1 : 0;
    TF_LAUNCH(cast16_kernel, dim3(cdiv(cols, TS), cdiv(rows, TS)), dim3(256), stream, x, rows, cols, (long)ldx, (uint16_t*)y16, (long)ldy, (uint16_t*)y16t, (long)ldyt,
              dtype == 2 ? 1 : 0, vec);
    return launch_status("tf_cast16_f32");
}

extern "C" int tf_cast16_multi_f32(const tf_cast16_item* items_dev, int n_items, int total_tiles, int dtype, void* stream) {
    TF_REQUIRE(items_dev && n_items > 0 && total_tiles > 0 && (dtype == 1 || dtype == 2), "tf_cast16_multi_f32: bad arguments (dtype 1 = bf16, 2 = fp16)");
    TF_LAUNCH(cast16_multi_kernel, dim3(total_tiles), dim3(256), stream, items_dev, n_items, dtype == 2 ? 1 : 0);
    return launch_status("tf_cast16_multi_f32");
}

extern "C" int tf_im2col3x3_f32(const float* x, float* cols, int B, int H, int W, int C, void* stream) {
    TF_REQUIRE(x && cols && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(cols), "tf_im2col3x3_f32: needs C %% 4 == 0 and 16-byte aligned tensors");
    const long total = (long)B * H * W * 9 * (C / 4);
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    TF_LAUNCH(im2col3x3_kernel, dim3((int)nb), dim3(256), stream, x, cols, B, H, W, C);
    return launch_status("tf_im2col3x3_f32");
}

extern "C" int tf_add_strided2_f32(const float* src, float* dst, int B, int Ho, int Wo, int C, int Hi, int Wi, void* stream) {
    TF_REQUIRE(src && dst && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && 2 * Ho - 1 <= Hi && 2 * Wo - 1 <= Wi && aligned16(src) && aligned16(dst),
               "tf_add_strided2_f32: needs C %% 4 == 0, 16-byte aligned tensors and (2 Ho - 1, 2 Wo - 1) <= (Hi, Wi)");
    const long n4 = (long)B * Ho * Wo * (C / 4);
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    TF_LAUNCH(add_strided2_kernel, dim3((int)blocks), dim3(256), stream, src, dst, Ho, Wo, C / 4, Hi, Wi, n4);
    return launch_status("tf_add_strided2_f32");
}

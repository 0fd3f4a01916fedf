/* C ABI of libtransfuser_hip.so - the MI355X (gfx950) kernels of the TransFuser training hot path.
 *
 * The reference (autonomousvision/transfuser) has no FFI / plugin layer: its native work is done
 * by third-party wheels (cuDNN/cuBLAS through torch 1.11, mmcv, torch_scatter).  Each entry point
 * below names the reference call site(s) whose arithmetic it replaces (file:line relative to the
 * reference repo).  Conventions:
 *   - every function returns 0 on success, non-zero on error; tf_last_error() gives the text;
 *   - all pointers are device pointers borrowed for the duration of the call; nothing is
 *     allocated, freed or retained; launches are asynchronous on `stream` (a hipStream_t);
 *     no entry point synchronises the device, so all of them are hipGraph-capturable;
 *   - activations are NHWC ("channels-last") fp32; conv weights are (Cout, kh, kw, Cin/groups),
 *     i.e. the channels_last physical layout of a torch (Cout, Cin/g, kh, kw) parameter;
 *   - token / linear tensors are row-major (rows, features).
 */
#ifndef TRANSFUSER_HIP_H
#define TRANSFUSER_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int tf_version(void);
const char* tf_last_error(void);

/* ---- dense contractions (fp32 MFMA, LDS-tiled) --------------------------------------------- */

/* C[z] (op)= alpha * A[z] . B[z] (+bias[col]) (+res) (relu).  Replaces nn.Linear fwd/dgrad/wgrad
 * (transfuser.py:500-507,539-541; model.py:592-605), the attention score / context bmm
 * (transfuser.py:519-523) and every 1x1 convolution on NHWC maps (timm RegNetY conv1/conv3,
 * transfuser.py:92-109).
 *   a_trans = 0: A(i,k) = a[i*lda + k]   a_trans = 1: A(i,k) = a[k*lda + i]
 *   b_trans = 0: B(k,j) = b[j*ldb + k]  (torch Linear / conv weight [out][in])
 *   b_trans = 1: B(k,j) = b[k*ldb + j]
 *   batch z in [0,batch): offset = (z / inner) * s?_outer + (z % inner) * s?_inner
 *   accumulate: 0 store, 1 C += (split-K with fp32 atomics may be used) */
typedef struct {
    const float* a; const float* b; float* c; const float* bias; const float* res;
    int m, n, k;
    int a_trans, b_trans;
    int64_t lda, ldb, ldc, ldres;
    int batch, inner;
    int64_t sa_outer, sa_inner, sb_outer, sb_inner, sc_outer, sc_inner;
    float alpha; int relu; int accumulate;
} tf_gemm_desc;
int tf_gemm_f32(const tf_gemm_desc* d, void* stream);

/* 2-D convolution as implicit GEMM (im2col gather -> LDS -> MFMA): kernel 1x1 or 3x3, stride 1/2,
 * pad, groups.  Replaces cuDNN conv fwd/bwd of the RegNetY trunks (timm regnety_032 via
 * transfuser.py:380,442), Seg/Depth decoders (transfuser.py:221-237,256-272), CenterNet heads
 * (model.py:93-99) and pred_bev (model.py:581-585). */
typedef struct {
    int B, Hi, Wi, Cin, Ho, Wo, Cout, ksize, stride, pad, groups;
} tf_conv_geom;
int tf_conv2d_fwd_f32(const tf_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int relu, void* stream);
int tf_conv2d_dgrad_f32(const tf_conv_geom* g, const float* dy, const float* w, float* dx, int accumulate, void* stream);
int tf_conv2d_wgrad_f32(const tf_conv_geom* g, const float* dy, const float* x, float* dw, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif

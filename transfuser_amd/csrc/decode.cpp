// Inference side of the CenterNet head: LidarCenterNetHead.decode_heatmap (model.py:436-497) with mmdet 2.25's
// get_local_maximum (3x3 max-pool NMS), get_topk_from_heatmap and transpose_and_gather_feat, fused into one launch per batch:
// one block per image keeps the (sigmoid) heat map in LDS, suppresses non-maxima, selects the k best cells by repeated block-wide
// arg-max (ties: lowest cell index) and gathers / decodes the box attributes of each selected cell.
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

constexpr int DEC_MAXCELLS = 16384;   // 128 x 128 feature map

// pred (B, fh, fw, 9 + nbins) logits: [hm, wh(2), off(2), yaw_cls(nbins), yaw_res, vel, brake(2)]  (HeadsFn's packing)
// out (B, k, 8) = [x, y, w, h (x ratio), yaw, velocity, brake, score]  (model.py:489-493)
__global__ void __launch_bounds__(256) centernet_decode_kernel(const float* __restrict__ pred, int fh, int fw, int nbins, int k, int kernel, float ratio,
                                                               float* __restrict__ out) {
    __shared__ float heat[DEC_MAXCELLS];
    __shared__ float sc[DEC_MAXCELLS];
    __shared__ float rv[4];
    __shared__ int ri[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cells = fh * fw, P = 9 + nbins, pad = (kernel - 1) / 2;
    const float* pb = pred + (long)b * cells * P;
    for (int i = tid; i < cells; i += 256) heat[i] = 1.f / (1.f + expf(-pb[(long)i * P]));
    __syncthreads();
    for (int i = tid; i < cells; i += 256) {      // get_local_maximum: keep where max_pool2d(heat, kernel, 1, pad) == heat, else score 0
        const int y = i / fw, x = i - y * fw;
        const float v = heat[i];
        float m = v;
        for (int dy = -pad; dy <= pad; ++dy)
            for (int dx = -pad; dx <= pad; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < fh && xx >= 0 && xx < fw) m = fmaxf(m, heat[yy * fw + xx]);
            }
        sc[i] = (m == v) ? v : 0.f;
    }
    __syncthreads();
    const float apc = (float)(2.0 * 3.141592653589793 / (double)nbins);
    for (int t = 0; t < k; ++t) {
        float bv = -1.f; int bi = 0x7fffffff;    // scores are >= 0; taken cells are marked -2
        for (int i = tid; i < cells; i += 256) { const float v = sc[i]; if (v > bv) { bv = v; bi = i; } }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = shfl(bv, lane ^ o); const int oi = (int)__float_as_uint(shfl(__uint_as_float((unsigned)bi), lane ^ o));
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = rv[0]; int i = ri[0];
            for (int w = 1; w < 4; ++w) if (rv[w] > v || (rv[w] == v && ri[w] < i)) { v = rv[w]; i = ri[w]; }
            rv[0] = v; ri[0] = i;
            float* o = out + ((long)b * k + t) * 8;
            if (i < cells) {
                sc[i] = -2.f;
                const float* q = pb + (long)i * P;
                const int y = i / fw, x = i - y * fw;
                int cls = 0; float cm = q[5];
                for (int c = 1; c < nbins; ++c) if (q[5 + c] > cm) { cm = q[5 + c]; cls = c; }   // torch.argmax: first maximum
                float yaw = (float)cls * apc + q[5 + nbins];                                      // class2angle (model.py:270-284)
                if (yaw > 3.14159265358979323846f) yaw -= (float)(2.0 * 3.141592653589793);
                o[0] = ((float)x + q[3]) * ratio; o[1] = ((float)y + q[4]) * ratio;
                o[2] = q[1] * ratio; o[3] = q[2] * ratio;
                o[4] = yaw; o[5] = q[6 + nbins];
                o[6] = (q[8 + nbins] > q[7 + nbins]) ? 1.f : 0.f;
                o[7] = v;
            } else {
                for (int c = 0; c < 8; ++c) o[c] = 0.f;     // k > number of cells
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int tf_centernet_decode_f32(const float* pred, int B, int fh, int fw, int num_dir_bins, int k, int kernel, float ratio, float* out, void* stream) {
    TF_REQUIRE(pred && out && B > 0 && fh > 0 && fw > 0 && fh * fw <= DEC_MAXCELLS && num_dir_bins > 0 && k > 0 && kernel >= 1 && (kernel & 1),
               "tf_centernet_decode_f32: bad arguments (feature map <= 16384 cells, odd kernel)");
    TF_LAUNCH(centernet_decode_kernel, dim3(B), dim3(256), stream, pred, fh, fw, num_dir_bins, k, kernel, ratio, out);
    return launch_status("tf_centernet_decode_f32");
}

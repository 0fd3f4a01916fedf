// GPU-side batch preparation: everything team_code_transfuser/data.py does per sample AFTER file decoding, for a whole
// batch at once on the device, so the CPU DataLoader workers only decode PNG / npy / json:
//   * align (data.py:411-444) + lidar_to_histogram_features (data.py:446-470) fused: 4x4 rigid transform of every point in fp64 (the
//     reference multiplies a float64 matrix with the float32 cloud) and the integer-exact 2-bin histogram of the transformed points;
//   * crop_image_cv2 (data.py:536-553) + cast: HWC uint8 -> CHW float32 centre crop with the augmentation's x shift;
//   * get_depth (data.py:358-372): 24-bit depth decode, clip at 50 m, rescale to [0, 1] (evaluated in double, stored as float);
//   * crop_seg + converter LUT (data.py:176-177, 555-571, config.py:88-115): class re-mapping to int64 labels;
//   * decode_pil_to_npy + load_crop_bev_npy (data.py:844-856, 586-612): bit-unpack of the encoded top-down map, 7-row shift, rotation by
//     the augmentation angle (bilinear, skimage.transform.rotate's centre / direction convention), 160 x 160 crop, 3-class argmax.
#include "tf_common.h"
#include "tf_hist.h"
#include "../../include/transfuser_hip.h"

using namespace tf;

namespace {

// pts (B, max_pts, stride >= 4) fp32 as loaded by data.py:166-170 (y already negated at load time); T (B, 16) row-major fp64 =
// degree_matrix @ Tr_vehicle_to_lidar @ inv(M1) @ M0 @ Tr_lidar_to_vehicle.  align(): p = (x, -y, z, 1); q = T p; (x', y', z') = (q0, -q1, q2).
__global__ void __launch_bounds__(256) lidar_align_hist_kernel(const float* __restrict__ pts, const int32_t* __restrict__ npts, int max_pts, int stride,
                                                               int vec4, const double* __restrict__ T, int* __restrict__ counters, float* __restrict__ aligned) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = npts ? (npts[b] < max_pts ? npts[b] : max_pts) : max_pts;
    const int ic = i < max_pts ? i : max_pts - 1;          // clamped address, unconditional loads (tf_hist.h)
    const float* p = pts + ((long)b * max_pts + ic) * stride;
    float fx, fy, fz, fw;
    if (vec4) { const float4 v = *reinterpret_cast<const float4*>(p); fx = v.x; fy = v.y; fz = v.z; fw = v.w; }
    else { fx = p[0]; fy = p[1]; fz = p[2]; fw = p[3]; }
    const double* t = T + (long)b * 16;                    // wave-uniform: scalar loads
    const double px = fx, py = -(double)fy, pz = fz;
    // numpy evaluates the row-times-vector dot products left to right in double: ((t0 x + t1 y) + t2 z) + t3
    const double x = ((t[0] * px + t[1] * py) + t[2] * pz) + t[3];
    const double y = -(((t[4] * px + t[5] * py) + t[6] * pz) + t[7]);
    const double z = ((t[8] * px + t[9] * py) + t[10] * pz) + t[11];
    if (aligned && i < n) {             // optional: the aligned cloud itself (PointPillars input, data.py:247-251), fp32 like the collated batch
        float* q = aligned + ((long)b * max_pts + i) * 4;
        q[0] = (float)x; q[1] = (float)y; q[2] = (float)z; q[3] = fw;
    }
    hist_add(counters + (long)b * 2 * 256 * 256, i < n ? hist_cell<double>(x, y, z) : -1);
}

// mode 0: rgb  -> out f32 (B, C, ch, cw) = src[b, sy + y, sx_b + x, c]
// mode 1: depth -> out f32 (B, ch, cw) = clip((R 65536 + G 256 + B) / (2^24 - 1), 0, 0.05) * 20     (src has 3 channels, RGB order)
// mode 2: seg  -> out i64 (B, ch, cw) = lut[src[b, sy + y, sx_b + x, 0]]
__global__ void __launch_bounds__(256) image_prep_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, int C, int ch, int cw, int sy, const int32_t* __restrict__ sx,
                                                         int mode, const uint8_t* __restrict__ lut, void* __restrict__ out, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % cw);
        long r = i / cw;
        const int y = (int)(r % ch);
        r /= ch;
        if (mode == 0) {
            const int c = (int)(r % C), b = (int)(r / C);
            const int xs = sx[b] + x, ys = sy + y;
            const bool ok = (unsigned)xs < (unsigned)Ws && (unsigned)ys < (unsigned)Hs;
            ((float*)out)[i] = ok ? (float)src[(((long)b * Hs + ys) * Ws + xs) * C + c] : 0.f;
        } else {
            const int b = (int)r;
            const int xs = sx[b] + x, ys = sy + y;
            const bool ok = (unsigned)xs < (unsigned)Ws && (unsigned)ys < (unsigned)Hs;
            const uint8_t* p = src + (((long)b * Hs + ys) * Ws + xs) * C;
            if (mode == 1) {
                double v = ok ? ((double)p[0] * 65536.0 + (double)p[1] * 256.0 + (double)p[2]) : 0.0;
                v /= (double)(256 * 256 * 256 - 1);
                v = v < 0.0 ? 0.0 : (v > 0.05 ? 0.05 : v);
                ((float*)out)[i] = (float)(v * 20.0);
            } else {
                ((int64_t*)out)[i] = ok ? (int64_t)lut[p[0]] : 0;
            }
        }
    }
}

// enc (B, S, S, 3) uint8 RGB (the encoded top-down image, S = 500): c0 = bit 7, c1 = bit 6 of the third channel (decode_pil_to_npy keeps
// rows 10:12 of the 15-plane unpacking); shifted down 7 rows; rotated by deg_b about the centre; out[b, y, x] over the crop
// [90:250, 170:330]: argmax(0, c0, c0 + c1).
__global__ void __launch_bounds__(256) bev_prep_kernel(const uint8_t* __restrict__ enc, int S, const float* __restrict__ deg, int64_t* __restrict__ out, long total) {
    constexpr int P = 160;
    const int start_x = 250 - P / 2, start_y = 250 - P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % P), y = (int)((i / P) % P), b = (int)(i / (P * P));
        const int ox = start_x + x, oy = start_y + y;            // position in the (shifted, rotated) S x S map
        auto plane = [&](int yy, int xx, int bit) -> float {     // shifted map value at integer position (0 outside)
            const int ys = yy - 7;
            if ((unsigned)xx >= (unsigned)S || ys < 0 || yy >= S) return 0.f;
            return (float)((enc[(((long)b * S + ys) * S + xx) * 3 + 2] >> bit) & 1);
        };
        float c0, c1;
        const float d = deg ? deg[b] : 0.f;
        if (d == 0.f) { c0 = plane(oy, ox, 7); c1 = plane(oy, ox, 6); }
        else {
            // skimage.transform.rotate(image, angle): counter-clockwise by `angle` degrees about ((cols - 1) / 2, (rows - 1) / 2), order 1,
            // mode 'constant': output(o) = input(R(o)) with the inverse map below
            const float a = d * 0.017453292519943295f, cs = cosf(a), sn = sinf(a);
            const float cx = 0.5f * (S - 1), cy = 0.5f * (S - 1);
            const float dx = ox - cx, dy = oy - cy;
            const float sxf = cs * dx - sn * dy + cx, syf = sn * dx + cs * dy + cy;
            const int x0 = (int)floorf(sxf), y0 = (int)floorf(syf);
            const float fx = sxf - x0, fy = syf - y0;
            auto lerp = [&](int bit) {
                return (1.f - fy) * ((1.f - fx) * plane(y0, x0, bit) + fx * plane(y0, x0 + 1, bit)) +
                       fy * ((1.f - fx) * plane(y0 + 1, x0, bit) + fx * plane(y0 + 1, x0 + 1, bit));
            };
            c0 = lerp(7); c1 = lerp(6);
        }
        const float v1 = c0, v2 = c0 + c1;                       // np.argmax: first maximum of (0, c0, c0 + c1)
        int64_t lab = 0;
        float best = 0.f;
        if (v1 > best) { best = v1; lab = 1; }
        if (v2 > best) { lab = 2; }
        out[i] = lab;
    }
}

inline int blocks_for(long total) { long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

extern "C" int tf_lidar_align_hist_f64(const float* points, const int32_t* num_points, int B, int max_points, int point_stride, const double* transforms,
                                       float* out, float* aligned_or_null, void* stream) {
    TF_REQUIRE(points && transforms && out && B > 0 && max_points >= 0 && point_stride >= 4, "tf_lidar_align_hist_f64: bad arguments");
    TF_REQUIRE(aligned16(out), "tf_lidar_align_hist_f64: out must be 16-byte aligned");
    const long n4 = (long)B * 2 * 256 * 256 / 4;
    TF_LAUNCH(hist_clear_kernel, dim3(cdiv(n4, 256)), dim3(256), stream, reinterpret_cast<float4*>(out), n4);
    if (max_points > 0) {
        const int vec4 = (point_stride == 4 && aligned16(points)) ? 1 : 0;
        TF_LAUNCH(lidar_align_hist_kernel, dim3(cdiv(max_points, 256), B), dim3(256), stream, points, num_points, max_points, point_stride, vec4, transforms,
                  reinterpret_cast<int*>(out), aligned_or_null);
    }
    TF_LAUNCH(hist_finish_kernel, dim3(cdiv(n4, 256)), dim3(256), stream, reinterpret_cast<float4*>(out), n4);
    return launch_status("tf_lidar_align_hist_f64");
}
// two launches with the zeroed counter workspace of tf_lidar_hist_ws_f32 (tf_lidar_hist_ws_bytes(B))
extern "C" int tf_lidar_align_hist_ws_f64(const float* points, const int32_t* num_points, int B, int max_points, int point_stride, const double* transforms,
                                          void* zero_ws, float* out, float* aligned_or_null, void* stream) {
    TF_REQUIRE(points && transforms && out && zero_ws && B > 0 && max_points >= 0 && point_stride >= 4, "tf_lidar_align_hist_ws_f64: bad arguments");
    TF_REQUIRE(aligned16(out) && aligned16(zero_ws), "tf_lidar_align_hist_ws_f64: out / workspace must be 16-byte aligned");
    const long n4 = (long)B * 2 * 256 * 256 / 4;
    if (max_points > 0) {
        const int vec4 = (point_stride == 4 && aligned16(points)) ? 1 : 0;
        TF_LAUNCH(lidar_align_hist_kernel, dim3(cdiv(max_points, 256), B), dim3(256), stream, points, num_points, max_points, point_stride, vec4, transforms,
                  reinterpret_cast<int*>(zero_ws), aligned_or_null);
    }
    TF_LAUNCH(hist_finish_ws_kernel, dim3(cdiv(n4, 256)), dim3(256), stream, reinterpret_cast<int4*>(zero_ws), reinterpret_cast<float4*>(out), n4);
    return launch_status("tf_lidar_align_hist_ws_f64");
}

extern "C" int tf_image_prep_u8(const uint8_t* src, int B, int Hs, int Ws, int C, int crop_h, int crop_w, int start_y, const int32_t* start_x, int mode,
                                const uint8_t* lut, void* out, void* stream) {
    TF_REQUIRE(src && out && start_x && B > 0 && Hs > 0 && Ws > 0 && C >= 1 && crop_h > 0 && crop_w > 0 && mode >= 0 && mode <= 2 && (mode != 2 || lut) &&
               (mode != 1 || C == 3), "tf_image_prep_u8: bad arguments");
    const long total = (long)B * (mode == 0 ? C : 1) * crop_h * crop_w;
    TF_LAUNCH(image_prep_kernel, dim3(blocks_for(total)), dim3(256), stream, src, Hs, Ws, C, crop_h, crop_w, start_y, start_x, mode, lut, out, total);
    return launch_status("tf_image_prep_u8");
}

extern "C" int tf_bev_prep_u8(const uint8_t* encoded, int B, int S, const float* degrees_or_null, int64_t* out, void* stream) {
    TF_REQUIRE(encoded && out && B > 0 && S >= 330, "tf_bev_prep_u8: needs the S x S x 3 encoded top-down image (S >= 330)");
    const long total = (long)B * 160 * 160;
    TF_LAUNCH(bev_prep_kernel, dim3(blocks_for(total)), dim3(256), stream, encoded, S, degrees_or_null, out, total);
    return launch_status("tf_bev_prep_u8");
}

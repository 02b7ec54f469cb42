// Device-side pre-processing on the e2e path:
//  * detection input: uint8 HWC -> /255 (float32) -> (x - mean)/std in float64 -> float32 NCHW  (pipeline2.py:312-314)
//  * recognition input: crop (src/det/test.py:123-130) -> resize to the target height keeping aspect, squash if wider than
//    the target, right-pad with 255 -> /255, ImageNet normalise in float32 -> NCHW  (pipeline2.py:92-128)
// The resize restates OpenCV's 8-bit INTER_LINEAR (cv2.resize default): 11-bit fixed-point coefficients, horizontal pass
// in int32, vertical pass ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2, and the exact-2x-downscale special case that
// OpenCV routes to the 2x2 box filter.  cv2 is absent from the build container: parity with cv2 itself is UNPINNED; the
// CPU oracle (oracle/preproc_cpu.py) restates the same published algorithm independently in numpy.
#include "common.h"

namespace ocrvi {

__constant__ double c_mean[3] = {0.485, 0.456, 0.406};
__constant__ double c_std[3] = {0.229, 0.224, 0.225};

__global__ void normalize_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int N, int H, int W) {
    const size_t plane = (size_t)H * W, total = (size_t)N * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, p = i % plane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)img[i * 3 + c] / 255.0f;
            out[(n * 3 + c) * plane + p] = (float)(((double)v - c_mean[c]) / c_std[c]);
        }
    }
}

struct AxisCoef { int s0, s1; int a0, a1; };
// OpenCV resizeLinear coefficient for destination index d on an axis of src length `ssize`, dst length `dsize`.
__device__ __forceinline__ AxisCoef axis_coef(int d, int ssize, int dsize) {
    const double scale = (double)ssize / (double)dsize;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    AxisCoef c;
    c.s0 = s;
    c.s1 = min(s + 1, ssize - 1);
    c.a0 = __float2int_rn((1.f - f) * 2048.f);
    c.a1 = __float2int_rn(f * 2048.f);
    return c;
}

__global__ void crop_resize_normalize_kernel(const uint8_t* __restrict__ images, int n_img, int H, int W, const int32_t* __restrict__ boxes,
                                             int B, int oh, int ow, float* __restrict__ out) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const size_t total = (size_t)B * oh * ow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        const size_t t = i / ow;
        const int y = (int)(t % oh), b = (int)(t / oh);
        const int32_t* bx = boxes + (size_t)b * 5;
        const int img = bx[0];
        // clamp the rectangle to the page exactly as crop_image does (src/det/test.py:126-129: x = max(0, x); bw = min(bw, w - x) --
        // the width is NOT reduced by the shift).  Boxes live in HBM, so the host cannot validate them; a rectangle that ends up
        // empty takes the zero-tensor path below
        const int cx = max(bx[1], 0), cy = max(bx[2], 0);
        const int cw = min(bx[3], W - cx), ch = min(bx[4], H - cy);
        float* o = out + ((size_t)b * 3 * oh + y) * ow + x;
        const size_t plane = (size_t)oh * ow;
        if (cw <= 0 || ch <= 0 || img < 0 || img >= n_img) {  // empty crop -> zeros tensor (pipeline2.py:154-156)
            o[0] = o[plane] = o[2 * plane] = 0.f;
            continue;
        }
        // new_w = int(w * (target_h / h))  (pipeline2.py:101-102), computed in double like Python floats
        int new_w = (int)((double)cw * ((double)oh / (double)ch));
        if (new_w > ow) new_w = ow;   // squash (pipeline2.py:104-105)
        if (new_w < 1) new_w = 1;
        int v[3] = {255, 255, 255};
        if (x < new_w) {
            const uint8_t* src = images + ((size_t)img * H + cy) * W * 3 + (size_t)cx * 3;
            const size_t rs = (size_t)W * 3;
            if (cw == 2 * new_w && ch == 2 * oh) {  // OpenCV: exact 2x decimation with INTER_LINEAR runs the 2x2 area filter
                const uint8_t* p0 = src + (size_t)(2 * y) * rs + (size_t)(2 * x) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p0[rs + c] + p0[rs + 3 + c] + 2) >> 2;
            } else {
                const AxisCoef ax = axis_coef(x, cw, new_w), ay = axis_coef(y, ch, oh);
                const uint8_t *r0 = src + (size_t)ay.s0 * rs, *r1 = src + (size_t)ay.s1 * rs;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
                    const int h1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
                    v[c] = (((ay.a0 * (h0 >> 4)) >> 16) + ((ay.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c * plane] = ((float)v[c] / 255.0f - mean[c]) / stdv[c];
    }
}

// cv2.resize(img, (dw, dh)) with the default INTER_LINEAR on uint8 HWC (resize_image_for_det, pipeline2.py:33-40)
__global__ void resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh, int dw) {
    const size_t total = (size_t)dh * dw;
    const bool area2 = (sw == 2 * dw && sh == 2 * dh);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % dw), y = (int)(i / dw);
        const size_t rs = (size_t)sw * 3;
        int v[3];
        if (area2) {
            const uint8_t* p0 = src + (size_t)(2 * y) * rs + (size_t)(2 * x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p0[rs + c] + p0[rs + 3 + c] + 2) >> 2;
        } else {
            const AxisCoef ax = axis_coef(x, sw, dw), ay = axis_coef(y, sh, dh);
            const uint8_t *r0 = src + (size_t)ay.s0 * rs, *r1 = src + (size_t)ay.s1 * rs;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
                const int h1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
                v[c] = (((ay.a0 * (h0 >> 4)) >> 16) + ((ay.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (uint8_t)v[c];
    }
}

}  // namespace ocrvi

using namespace ocrvi;

extern "C" int ocrvi_resize_u8(int device, const uint8_t* src, int src_h, int src_w, uint8_t* dst, int dst_h, int dst_w, void* stream) {
    OCRVI_CHECK(src && dst && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, OCRVI_EINVAL, "resize_u8: bad argument");
    DeviceGuard dg(device);  // the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    const size_t total = (size_t)dst_h * dst_w;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_h, src_w, dst, dst_h, dst_w);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

extern "C" int ocrvi_normalize_u8(int device, const uint8_t* images, int N, int H, int W, float* out, void* stream) {
    OCRVI_CHECK(images && out && N > 0 && H > 0 && W > 0, OCRVI_EINVAL, "normalize_u8: bad argument");
    DeviceGuard dg(device);  // the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    const size_t total = (size_t)N * H * W;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, out, N, H, W);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

extern "C" int ocrvi_crop_resize_normalize(int device, const uint8_t* images, int n_img, int H, int W, const int32_t* boxes, int B, int out_h,
                                           int out_w, float* out, void* stream) {
    OCRVI_CHECK(images && boxes && out && n_img > 0 && B > 0 && out_h > 0 && out_w > 0, OCRVI_EINVAL, "crop_resize_normalize: bad argument");
    DeviceGuard dg(device);  // the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    const size_t total = (size_t)B * out_h * out_w;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(crop_resize_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, n_img, H, W, boxes, B, out_h, out_w, out);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

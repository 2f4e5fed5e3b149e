// Person-crop extraction for the pose net: the step right before POSE in the tracking
// pipeline.  Reference: lib/pose/utils/transforms.py:231-240 `transform_image` = cv2.warpAffine(img, t[:2],
// (res_w, res_h)) with t = get_transform(center, scale, res) (transforms.py:173-184), called per box on the
// host from lib/tracking/net_utils.py:49-57 (then an H2D copy of every crop).  Here: one launch for all boxes
// of a frame, frame resident in HBM as HWC uint8, inverse-mapped bilinear sampling with constant-0 border
// (cv2 defaults), optional per-channel normalisation, output already in the net's NCHW fp32 input layout.
// Parity: cv2 is not available in this environment and cv2 quantises the interpolation weights to 1/32 px;
// this kernel interpolates in fp32 => parity with the reference's crops is UNPINNED (oracle/tracking_ref.py
// restates the same exact-bilinear definition).
#include "ft_common.h"

namespace ft {

__global__ __launch_bounds__(256) void crop_affine_kernel(const uint8_t* __restrict__ img, int H, int W, int C,
                                                          const float* __restrict__ boxes, int rh, int rw,
                                                          const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                          float pre_scale, float* __restrict__ out, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % rw);
    size_t t = i / rw;
    const int y = (int)(t % rh);
    const size_t b = t / rh;
    const float cx = boxes[b * 3 + 0], cy = boxes[b * 3 + 1], scale = boxes[b * 3 + 2];
    // inverse of t = [[rh/scale, 0, -rh*cx/scale + rw/2], [0, rh/scale, -rh*cy/scale + rh/2]]
    const float g = scale / (float)rh;
    const float sx = ((float)x - 0.5f * (float)rw) * g + cx;
    const float sy = ((float)y - 0.5f * (float)rh) * g + cy;
    const float fx0 = floorf(sx), fy0 = floorf(sy);
    const float ax = sx - fx0, ay = sy - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    for (int c = 0; c < C; ++c) {
      const float p00 = (vx0 && vy0) ? (float)img[((size_t)y0 * W + x0) * C + c] : 0.f;
      const float p01 = (vx1 && vy0) ? (float)img[((size_t)y0 * W + x0 + 1) * C + c] : 0.f;
      const float p10 = (vx0 && vy1) ? (float)img[((size_t)(y0 + 1) * W + x0) * C + c] : 0.f;
      const float p11 = (vx1 && vy1) ? (float)img[((size_t)(y0 + 1) * W + x0 + 1) * C + c] : 0.f;
      float v = (1.f - ay) * ((1.f - ax) * p00 + ax * p01) + ay * ((1.f - ax) * p10 + ax * p11);
      v *= pre_scale;
      if (mean) v -= mean[c];
      if (inv_std) v *= inv_std[c];
      out[((b * C + c) * rh + y) * rw + x] = v;
    }
  }
}

}  // namespace ft

using namespace ft;

extern "C" int ft_crop_affine_fwd(const uint8_t* img, int H, int W, int C, const float* boxes, int nb, int rh, int rw,
                                  const float* mean, const float* inv_std, float pre_scale, float* out,
                                  ft_stream_t stream) {
  if (!img || !boxes || !out || H <= 0 || W <= 0 || C <= 0 || C > 4 || nb <= 0 || rh <= 0 || rw <= 0) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)nb * rh * rw;
  size_t g = (total + 255) / 256;
  g = g > 16384 ? 16384 : g;
  hipLaunchKernelGGL(crop_affine_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), img, H, W, C, boxes, rh, rw,
                     mean, inv_std, pre_scale, out, total);
  FT_LAUNCH_CHECK("crop_affine_kernel");
  return FT_OK;
}

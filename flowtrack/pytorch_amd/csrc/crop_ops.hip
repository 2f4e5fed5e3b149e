// Person-crop extraction for the pose net: the step right before POSE in the tracking
// pipeline.  Reference: lib/pose/utils/transforms.py:231-240 `transform_image` = cv2.warpAffine(img, t[:2],
// (res_w, res_h)) with t = get_transform(center, scale, res) (transforms.py:173-184), called per box on the
// host from lib/tracking/net_utils.py:49-57 (then an H2D copy of every crop).  Here: one launch for all boxes
// of a frame, frame resident in HBM as HWC uint8, inverse-mapped bilinear sampling with constant-0 border
// (cv2 defaults), optional per-channel normalisation, output already in the net's NCHW fp32 input layout.
// Two forms.  `ft_crop_affine_fwd`: the ideal bilinear crop in fp32 (fast default; differs from cv2's uint8 result by
// up to 0.5 grey level + cv2's 1/32-px coordinate snap).  `ft_crop_affine_cv2_fwd` (round 5): what cv2.warpAffine RETURNS
// for a uint8 frame — bit-exact to the RESTATED classic OpenCV path (oracle/tracking_ref.py; parity with a real cv2 build is unpinned:
// none is installed here) — OpenCV's fixed-point INTER_LINEAR (imgwarp.cpp: cv::warpAffine, WarpAffineInvoker,
// initInterTab2D, remapBilinear<FixedPtCast<int, uchar, 15>>): inverse map in AB_BITS = 10 fixed point with
// round_delta = 16, 1/32-px fractional index, 15-bit weights a * b * 32, (sum + 2^14) >> 15, constant-0 border taps.
// Integer work => the test bar is bit-exact against oracle/tracking_ref.py::warp_affine_cv2_ref.
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

// cvRound(double) -> int as x86's cvtsd2si does it: round half to even, INT_MIN when out of range / NaN.
__device__ __forceinline__ int cv_round_i32(double v) {
  const double r = rint(v);
  return (fabs(r) < 2147483648.0) ? (int)r : (int)0x80000000;
}

// One thread per output pixel.  All coordinate arithmetic mirrors WarpAffineInvoker::operator(): doubles multiplied and
// added with explicit round-to-nearest intrinsics (no FMA contraction: OpenCV's set-up code is plain SSE2 double math),
// then 32-bit wrapping integer adds and arithmetic shifts.
__global__ __launch_bounds__(256) void crop_affine_cv2_kernel(const uint8_t* __restrict__ img, int H, int W, int C,
                                                              const double* __restrict__ minv, int rh, int rw,
                                                              const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                              float pre_scale, uint8_t* __restrict__ out_u8,
                                                              float* __restrict__ out, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % rw);
    size_t t = i / rw;
    const int y = (int)(t % rh);
    const size_t b = t / rh;
    const double* M = minv + b * 6;
    const double dx = (double)x, dy = (double)y;
    const int adelta = cv_round_i32(__dmul_rn(__dmul_rn(M[0], dx), 1024.0));
    const int bdelta = cv_round_i32(__dmul_rn(__dmul_rn(M[3], dx), 1024.0));
    const unsigned X0 = (unsigned)cv_round_i32(__dmul_rn(__dadd_rn(__dmul_rn(M[1], dy), M[2]), 1024.0)) + 16u;
    const unsigned Y0 = (unsigned)cv_round_i32(__dmul_rn(__dadd_rn(__dmul_rn(M[4], dy), M[5]), 1024.0)) + 16u;
    const int X = (int)(X0 + (unsigned)adelta) >> 5;      // AB_BITS - INTER_BITS
    const int Y = (int)(Y0 + (unsigned)bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;                          // saturate_cast<short>
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }          // initInterTab2D: short saturation of 2^15 + the sum fix-up
    const bool vx0 = (unsigned)sx < (unsigned)W, vx1 = (unsigned)(sx + 1) < (unsigned)W;
    const bool vy0 = (unsigned)sy < (unsigned)H, vy1 = (unsigned)(sy + 1) < (unsigned)H;
    for (int c = 0; c < C; ++c) {
      const int p00 = (vx0 && vy0) ? (int)img[((size_t)sy * W + sx) * C + c] : 0;
      const int p01 = (vx1 && vy0) ? (int)img[((size_t)sy * W + sx + 1) * C + c] : 0;
      const int p10 = (vx0 && vy1) ? (int)img[((size_t)(sy + 1) * W + sx) * C + c] : 0;
      const int p11 = (vx1 && vy1) ? (int)img[((size_t)(sy + 1) * W + sx + 1) * C + c] : 0;
      int q = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
      q = q < 0 ? 0 : (q > 255 ? 255 : q);
      if (out_u8) out_u8[((b * rh + y) * rw + x) * C + c] = (uint8_t)q;
      if (out) {
        float v = (float)q * pre_scale;
        if (mean) v -= mean[c];
        if (inv_std) v *= inv_std[c];
        out[((b * C + c) * rh + y) * rw + x] = v;
      }
    }
  }
}

}  // namespace ft

using namespace ft;

extern "C" int ft_crop_affine_cv2_fwd(const uint8_t* img, int H, int W, int C, const double* minv, int nb, int rh, int rw,
                                      const float* mean, const float* inv_std, float pre_scale, uint8_t* out_u8,
                                      float* out, ft_stream_t stream) {
  if (!img || !minv || (!out && !out_u8) || H <= 0 || W <= 0 || C <= 0 || C > 4 || nb <= 0 || rh <= 0 || rw <= 0) return FT_ERR_INVALID_ARG;
  const size_t total = (size_t)nb * rh * rw;
  size_t g = (total + 255) / 256;
  g = g > 16384 ? 16384 : g;
  hipLaunchKernelGGL(crop_affine_cv2_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), img, H, W, C, minv, rh, rw,
                     mean, inv_std, pre_scale, out_u8, out, total);
  FT_LAUNCH_CHECK("crop_affine_cv2_kernel");
  return FT_OK;
}

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

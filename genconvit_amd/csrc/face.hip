// Row N4 of SURVEY.md section 8(f): the crop + resize of the reference's face_rec (model/pred_func.py:79-85),
//     cv2.resize(frame[top:bottom, left:right], (224, 224), interpolation=cv2.INTER_AREA)
// for many faces in one launch, 8-bit RGB frames resident in HBM.  One thread = one output pixel (3 channels).
// The arithmetic follows OpenCV's imgproc/resize.cpp case by case — whole-factor shrink (integer block sums), general
// shrink (fp32 area tables, x then y, one multiply and one add per term), and the area-mode bilinear used as soon as a
// dimension grows (11-bit fixed point) — exactly as oracle/cv_area.py restates it; the kernel is bit-equal to that
// restatement (tests/test_parity_gpu.py), which itself is unpinned against a real OpenCV build (see its header).
// fp contraction is off for this TU: every multiply and add rounds on its own, as in the CPU code.
#pragma clang fp contract(off)
#include <cfloat>
#include <cstdint>

#include "common.h"

namespace gcv {

struct AreaSpan {          // computeResizeAreaTab for one destination index
  int s1, s2;              // full-weight source indices [s1, s2)
  bool left, right;        // partial cells s1 - 1 and s2
  float a_left, a_mid, a_right;
};

__device__ __forceinline__ AreaSpan area_span(int d, int ssize, double scale) {
  AreaSpan t;
  const double f1 = d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = min(s2, ssize - 1);
  s1 = min(s1, s2);
  t.s1 = s1; t.s2 = s2;
  t.left = (double)s1 - f1 > 1e-3;
  t.a_left = (float)(((double)s1 - f1) / cell);
  t.a_mid = (float)(1.0 / cell);
  t.right = f2 - (double)s2 > 1e-3;
  t.a_right = (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell);
  return t;
}

struct LinCoef { int s, a0, a1; bool pair; };   // area-mode bilinear: source index, 11-bit weights, right neighbour exists

__device__ __forceinline__ LinCoef lin_coef(int d, int ssize, double scale, double inv_scale) {
  LinCoef c;
  int s = (int)floor(d * scale);
  float f = (float)((double)(d + 1) - (double)(s + 1) * inv_scale);
  f = f <= 0.0f ? 0.0f : f - floorf(f);
  if (s < 0) { f = 0.0f; s = 0; }
  c.pair = s + 1 < ssize;
  if (s >= ssize - 1) { f = 0.0f; s = ssize - 1; }
  c.s = s;
  c.a0 = max(-32768, min(32767, __float2int_rn((1.0f - f) * 2048.0f)));
  c.a1 = max(-32768, min(32767, __float2int_rn(f * 2048.0f)));
  return c;
}

__device__ __forceinline__ unsigned char sat_u8(float v) { return (unsigned char)max(0, min(255, __float2int_rn(v))); }

__global__ void __launch_bounds__(256) face_crop_resize_kernel(const unsigned char* __restrict__ frames, int nframes,
                                                               int H, int W, const int* __restrict__ boxes, int n,
                                                               unsigned char* __restrict__ out, int S) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * S * S) return;
  const int box = (int)(idx / (S * S)), rem = (int)(idx - (int64_t)box * S * S);
  const int dy = rem / S, dx = rem - dy * S;
  const int f = boxes[5 * box], top = boxes[5 * box + 1], right = boxes[5 * box + 2], bottom = boxes[5 * box + 3],
            left = boxes[5 * box + 4];
  unsigned char* o = out + idx * 3;
  if (f < 0 || f >= nframes || top < 0 || left < 0 || bottom > H || right > W || top >= bottom || left >= right) {
    o[0] = o[1] = o[2] = 0;                  // a box outside its frame reads nothing (the host wrapper rejects it first)
    return;
  }
  const int sh = bottom - top, sw = right - left;
  const unsigned char* src = frames + (((int64_t)f * H + top) * W + left) * 3;
  const int64_t rs = (int64_t)W * 3;
  const double inv_sx = (double)S / sw, inv_sy = (double)S / sh;
  const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
  if (scale_x >= 1.0 && scale_y >= 1.0) {
    const int isx = (int)rint(scale_x), isy = (int)rint(scale_y);
    if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON) {      // ResizeAreaFast_
      int sum[3] = {0, 0, 0};
      for (int y = 0; y < isy; ++y) {
        const unsigned char* p = src + (int64_t)(dy * isy + y) * rs + (int64_t)dx * isx * 3;
        for (int x = 0; x < isx; ++x)
#pragma unroll
          for (int c = 0; c < 3; ++c) sum[c] += p[3 * x + c];
      }
      if (isx == 2 && isy == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (unsigned char)((sum[c] + 2) >> 2);
      } else {
        const float scale = 1.0f / (float)(isx * isy);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = sat_u8((float)sum[c] * scale);
      }
      return;
    }
    // ResizeArea_: rows in table order; within a row, columns in table order
    const AreaSpan tx = area_span(dx, sw, scale_x), ty = area_span(dy, sh, scale_y);
    float total[3] = {0.0f, 0.0f, 0.0f};
    auto row = [&](int sy, float beta) {
      const unsigned char* p = src + (int64_t)sy * rs;
      float buf[3] = {0.0f, 0.0f, 0.0f};
      if (tx.left)
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)p[3 * (tx.s1 - 1) + c] * tx.a_left;
      for (int sx = tx.s1; sx < tx.s2; ++sx)
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)p[3 * sx + c] * tx.a_mid;
      if (tx.right)
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)p[3 * tx.s2 + c] * tx.a_right;
#pragma unroll
      for (int c = 0; c < 3; ++c) total[c] = total[c] + beta * buf[c];
    };
    if (ty.left) row(ty.s1 - 1, ty.a_left);
    for (int sy = ty.s1; sy < ty.s2; ++sy) row(sy, ty.a_mid);
    if (ty.right) row(ty.s2, ty.a_right);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = sat_u8(total[c]);
    return;
  }
  // a dimension grows: bilinear in area mode, HResizeLinear then VResizeLinear in 11-bit fixed point
  const LinCoef cx = lin_coef(dx, sw, scale_x, inv_sx), cy = lin_coef(dy, sh, scale_y, inv_sy);
  const int y0 = min(max(cy.s, 0), sh - 1), y1 = min(max(cy.s + 1, 0), sh - 1);
  const unsigned char* p0 = src + (int64_t)y0 * rs + 3 * cx.s;
  const unsigned char* p1 = src + (int64_t)y1 * rs + 3 * cx.s;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = cx.pair ? p0[c] * cx.a0 + p0[3 + c] * cx.a1 : p0[c] * 2048;
    const int h1 = cx.pair ? p1[c] * cx.a0 + p1[3 + c] * cx.a1 : p1[c] * 2048;
    o[c] = (unsigned char)((((cy.a0 * (h0 >> 4)) >> 16) + ((cy.a1 * (h1 >> 4)) >> 16) + 2) >> 2);
  }
}

int launch_face_crop_resize(const unsigned char* frames, int nframes, int H, int W, const int* boxes, int n,
                            unsigned char* out, int S, hipStream_t s) {
  GCV_REQUIRE(nframes > 0 && H > 0 && W > 0 && S > 0 && S <= 4096, "face crop: bad geometry");
  if (n <= 0) return 0;
  const int64_t total = (int64_t)n * S * S;
  GCV_REQUIRE(total < ((int64_t)1 << 31) * 256, "face crop: too many output pixels for one launch");
  hipLaunchKernelGGL(face_crop_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frames, nframes, H, W,
                     boxes, n, out, S);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace gcv

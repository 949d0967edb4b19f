// Launch wrappers for kernels.h (included by kern_{f32,f16,bf16}.hip).
#pragma once
#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "kernels_dev.h"

namespace gcv {

template <typename T>
int launch_stem_ln(const T* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* wp, const float* bias,
                   const float* lnw, const float* lnb, T* out, int nimg, int Ho, int Wo, float eps, hipStream_t s) {
  const int64_t total = (int64_t)nimg * Ho * Wo;
  GCV_REQUIRE(total > 0, "stem: empty");
  GCV_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15u) == 0, "stem: packed weights must be 16-byte aligned");
  if constexpr (sizeof(T) == 2) {
    // matrix-pipe stem: the two frame layouts of the path with 8-byte aligned patch pieces (GCV_STEM_VALU=1: A/B switch)
    static const bool valu = exp_env("GCV_STEM_VALU") != nullptr;
    const bool al = (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && ((sb | sy) & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    const bool nchw4 = al && sx == 1 && (sc & 3) == 0, nhwc4 = al && sc == 1 && sx == 3;
    if (!valu && (nchw4 || nhwc4) && total < (int64_t)1 << 30) {
      const int ntiles = (int)cdiv64(total, 32);
      const int grid = std::min(cdiv(ntiles, 4), 256 * 8);
      hipLaunchKernelGGL((stem_ln_mfma_kernel<T>), dim3(grid), dim3(256), 0, s, x, sb, sc, sy, nhwc4 ? 1 : 0, wp, bias, lnw,
                         lnb, out, (int)total, Ho, Wo, eps);
      GCV_CHECK_HIP(hipGetLastError());
      return 0;
    }
  }
  hipLaunchKernelGGL((stem_ln_kernel<T>), dim3((unsigned)cdiv64(total, kStemTok)), dim3(256), 0, s, x, sb, sc, sy, sx, wp,
                     bias, lnw, lnb, out, nimg, Ho, Wo, eps);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int C>
static int launch_dw_c(const T* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb, T* y,
                       int nimg, int H, int W, float eps, hipStream_t s) {
  constexpr int TILES = (C == 96) ? 2 : 1;
  constexpr int TPB = C * TILES;
  constexpr size_t LDS = (size_t)TILES * 49 * C * 4 + (size_t)TILES * 49 * 2 * 4;
  if (LDS > 64 * 1024) GCV_ENSURE_LDS((dwconv7_ln_kernel<T, C>), LDS);
  const int tiles = nimg * cdiv(H, 7) * cdiv(W, 7);
  hipLaunchKernelGGL((dwconv7_ln_kernel<T, C>), dim3(cdiv(tiles, TILES)), dim3(TPB), LDS, s, x, wdw, bdw, lnw, lnb, y,
                     nimg, H, W, eps);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_dwconv7_ln(const T* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb, T* y,
                      int nimg, int H, int W, int C, float eps, hipStream_t s) {
  GCV_REQUIRE(nimg > 0 && H > 0 && W > 0, "dwconv: empty");
  // the rolling-strip kernel covers every ConvNeXt-T shape but the 3x3 map of the 112-px pass
  // (GCV_DWCONV_GENERIC=1: A/B switch, the generic tile kernel everywhere)
  static const bool generic = exp_env("GCV_DWCONV_GENERIC") != nullptr;
  if (!generic && dwconv_roll_applicable<T>(H, W, C) &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0)
    return launch_dwconv7_ln_roll<T>(x, wdw, bdw, lnw, lnb, y, nimg, H, W, C, eps, s);
  if (!generic && C == 768 && H == W && H <= 4) {          // stage-3 maps of small inputs (3 x 3 in the 112-pixel pass)
#define GCV_DW_TINY(S)                                                                                              \
    if (H == S) {                                                                                                   \
      hipLaunchKernelGGL((dwconv7_ln_tiny_kernel<T, 768, S>), dim3(nimg), dim3(768), 0, s, x, wdw, bdw, lnw, lnb, y, eps); \
      GCV_CHECK_HIP(hipGetLastError());                                                                             \
      return 0;                                                                                                     \
    }
    GCV_DW_TINY(1) GCV_DW_TINY(2) GCV_DW_TINY(3) GCV_DW_TINY(4)
#undef GCV_DW_TINY
  }
  switch (C) {
    case 96:  return launch_dw_c<T, 96>(x, wdw, bdw, lnw, lnb, y, nimg, H, W, eps, s);
    case 192: return launch_dw_c<T, 192>(x, wdw, bdw, lnw, lnb, y, nimg, H, W, eps, s);
    case 384: return launch_dw_c<T, 384>(x, wdw, bdw, lnw, lnb, y, nimg, H, W, eps, s);
    case 768: return launch_dw_c<T, 768>(x, wdw, bdw, lnw, lnb, y, nimg, H, W, eps, s);
  }
  set_error("dwconv7_ln: C must be one of 96/192/384/768");
  return -3;
}

template <typename T>
int launch_ln_patchify(const T* x, const float* w, const float* b, T* out, int nimg, int H, int W, int C, float eps,
                       hipStream_t s) {
  GCV_REQUIRE(C <= 768 && nimg > 0, "ln_patchify: C <= 768");
  const int64_t total = (int64_t)nimg * H * W;
  if constexpr (sizeof(T) == 2) {
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 3u) == 0;
#define GCV_LNP(CC)                                                                                              \
    if (C == CC && al) {                                                                                         \
      constexpr int PPB = 256 / (CC / 6);                                                                        \
      hipLaunchKernelGGL((ln_patchify_vec_kernel<T, CC>), dim3((unsigned)cdiv64(total, PPB)), dim3(256), 0, s, x, w, b, \
                         out, nimg, H, W, eps);                                                                  \
      GCV_CHECK_HIP(hipGetLastError());                                                                          \
      return 0;                                                                                                  \
    }
    GCV_LNP(96) GCV_LNP(192) GCV_LNP(384)
#undef GCV_LNP
  }
  hipLaunchKernelGGL((ln_patchify_kernel<T>), dim3((unsigned)cdiv64(total, 4)), dim3(256), 0, s, x, w, b, out, nimg, H,
                     W, C, eps);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_layernorm_rows(const T* x, const float* w, const float* b, T* out, int64_t rows, int C, float eps,
                          hipStream_t s) {
  GCV_REQUIRE(C <= 1536 && rows > 0, "layernorm_rows: C <= 1536");
  hipLaunchKernelGGL((layernorm_rows_kernel<T>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, x, w, b, out, rows,
                     C, eps);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_pool_ln(const T* x, const float* w, const float* b, T* out, int nimg, int HW, int C, float eps,
                   hipStream_t s, int seg_n, int row_stride, int row0) {
  GCV_REQUIRE(C == 768 && nimg > 0 && HW > 0, "pool_ln: C == 768");
  if (seg_n <= 0) { seg_n = nimg; row_stride = 1; row0 = 0; }            // plain order: image b -> row b
  hipLaunchKernelGGL((pool_ln_kernel<T>), dim3(nimg), dim3(256), 0, s, x, w, b, out, HW, C, eps, seg_n, row_stride, row0);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_conv3_first(const T* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* wp, const float* bias,
                       T* out, int nimg, int H, int W, bool pool, int act, hipStream_t s) {
  GCV_REQUIRE(H % 2 == 0 && W % 2 == 0 && nimg > 0, "conv3_first: even H, W");
  const int64_t total = (int64_t)nimg * (H / 2) * (W / 2);
  if constexpr (sizeof(T) == 2) {
    // matrix-pipe variant: NCHW frames up to 224 wide whose rows are 16-byte pieces (GCV_CONV3_VALU=1: A/B switch)
    static const bool valu = exp_env("GCV_CONV3_VALU") != nullptr;
    const bool ok = sx == 1 && W % 32 == 0 && W <= kConv3MaxW && H % 8 == 0 && ((sb | sc | sy) & 7) == 0 &&
                    (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 &&
                    (reinterpret_cast<uintptr_t>(bias) & 15u) == 0;
    if (ok && !valu) {
      const dim3 g2((unsigned)(nimg * (H / 8)));
      if (pool)
        hipLaunchKernelGGL((conv3_first_mfma_kernel<T, true>), g2, dim3(256), 0, s, x, sb, sc, sy, wp, bias, out, nimg, H, W, act);
      else
        hipLaunchKernelGGL((conv3_first_mfma_kernel<T, false>), g2, dim3(256), 0, s, x, sb, sc, sy, wp, bias, out, nimg, H, W, act);
      GCV_CHECK_HIP(hipGetLastError());
      return 0;
    }
  }
  const dim3 grid((unsigned)cdiv64(total, 256));
  if (pool)
    hipLaunchKernelGGL((conv3_first_kernel<T, true>), grid, dim3(256), 0, s, x, sb, sc, sy, sx, wp, bias, out, nimg, H,
                       W, act);
  else
    hipLaunchKernelGGL((conv3_first_kernel<T, false>), grid, dim3(256), 0, s, x, sb, sc, sy, sx, wp, bias, out, nimg,
                       H, W, act);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_convt2_small(const T* x, const float* wp, const float* bias, T* out, int nimg, int H, int W, int act,
                        hipStream_t s) {
  const int64_t total = (int64_t)nimg * H * W;
  GCV_REQUIRE(total > 0, "convt2_small: empty");
  hipLaunchKernelGGL((convt2_small_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, x, wp, bias, out,
                     nimg, H, W, act);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_reparam(const float* partial, int splitk, const float* bias, const float* eps, float* mu_out, T* z_nhwc,
                   int B, int N, hipStream_t s) {
  GCV_REQUIRE(N == 12544 && B > 0, "reparam: latent 12544 = 256*7*7");
  const int64_t total = (int64_t)B * N;
  hipLaunchKernelGGL((reparam_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, partial, splitk, bias,
                     eps, mu_out, z_nhwc, B, N);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_head_tail(const T* h, const float* w, const float* bias, float* logits, int B, int K, hipStream_t s) {
  GCV_REQUIRE(B > 0, "head_tail: empty");
  hipLaunchKernelGGL((head_tail_kernel<T>), dim3(cdiv(B, 4)), dim3(256), 0, s, h, w, bias, logits, B, K);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_head_tail_splitk(const float* partial, int S, const float* b1, int act, const float* w, const float* bias,
                            float* logits, int B, int K, hipStream_t s) {
  GCV_REQUIRE(B > 0 && S >= 1, "head_tail_splitk: empty");
#define GCV_HT(A)                                                                                                    \
  case A: hipLaunchKernelGGL((head_tail_splitk_kernel<T, A>), dim3(B), dim3(256), 0, s, partial, S, b1, w, bias, \
                             logits, B, K); break;
  switch (act) {
    GCV_HT(ACT_NONE) GCV_HT(ACT_RELU) GCV_HT(ACT_GELU) GCV_HT(ACT_LEAKY)
    default: set_error("bad activation code"); return -2;
  }
#undef GCV_HT
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_resize_mse(const T* xhat, const T* img, T* recon, float* msepart, float* mse, int B, hipStream_t s) {
  GCV_REQUIRE(B > 0, "resize: empty");
  const int nblk = 224 * 224 / 256;   // 196
  hipLaunchKernelGGL((resize_mse_kernel<T>), dim3(nblk, B), dim3(256), 0, s, xhat, img, recon, mse ? msepart : nullptr);
  GCV_CHECK_HIP(hipGetLastError());
  if (mse) {
    hipLaunchKernelGGL(mse_finish_kernel, dim3(B), dim3(256), 0, s, msepart, mse, nblk, 1.0f / (3.0f * 224.0f * 224.0f));
    GCV_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

template <typename T>
int launch_swin_window_attn(const T* qkv, const float* rpb, T* out, int nimg, int H, int W, int C, int nH, int shift,
                            hipStream_t s) {
  GCV_REQUIRE(H % 7 == 0 && W % 7 == 0 && C == nH * 32 && nimg > 0, "swin attention: 7x7 windows, head_dim 32");
  GCV_REQUIRE(shift == 0 || (shift == 3 && H > 7), "swin attention: shift is 0 or 3");
  const float scale = 0.17677669529663689f;   // 32^-0.5
  if constexpr (sizeof(T) == 2) {
    static const bool valu_only = exp_env("GCV_SWIN_ATTN_VALU") != nullptr;   // A/B switch
    if (!valu_only && (reinterpret_cast<uintptr_t>(qkv) & 15u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && C % 8 == 0) {
      hipLaunchKernelGGL((swin_window_attn_mfma_kernel<T>), dim3(nimg * (H / 7) * (W / 7), nH), dim3(64), 0, s, qkv, rpb,
                         out, H, W, C, nH, shift, scale);
      GCV_CHECK_HIP(hipGetLastError());
      return 0;
    }
  }
  hipLaunchKernelGGL((swin_window_attn_kernel<T>), dim3(nimg * (H / 7) * (W / 7), nH), dim3(64), 0, s, qkv, rpb, out,
                     H, W, C, nH, shift, scale);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_patch_merge_ln(const T* x, const float* w, const float* b, T* out, int nimg, int H, int W, int C, float eps,
                          hipStream_t s) {
  GCV_REQUIRE(H % 2 == 0 && W % 2 == 0 && 4 * C <= 1536 && nimg > 0, "patch merging: even H, W; 4C <= 1536");
  const int64_t rows = (int64_t)nimg * (H / 2) * (W / 2);
  hipLaunchKernelGGL((patch_merge_ln_kernel<T>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, x, w, b, out, nimg,
                     H, W, C, eps);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T> int launch_mean_tokens(const T* x, T* out, int nimg, int L, int C, hipStream_t s) {
  GCV_REQUIRE(nimg > 0 && L > 0 && C > 0, "mean_tokens: empty");
  hipLaunchKernelGGL((mean_tokens_kernel<T>), dim3(nimg), dim3(256), 0, s, x, out, L, C);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T> int launch_preprocess(const unsigned char* u8, T* out, int n, int H, int W, hipStream_t s) {
  GCV_REQUIRE(n > 0 && H > 0 && W > 0 && u8 && out, "preprocess: empty");
  const int64_t total = (int64_t)n * H * W;
  hipLaunchKernelGGL((preprocess_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, u8, out, total, H * W);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

#define GCV_INSTANTIATE_KERNELS(T)                                                                                    \
  template int launch_stem_ln<T>(const T*, int64_t, int64_t, int64_t, int64_t, const float*, const float*,            \
                                 const float*, const float*, T*, int, int, int, float, hipStream_t);                  \
  template int launch_dwconv7_ln<T>(const T*, const float*, const float*, const float*, const float*, T*, int, int,   \
                                    int, int, float, hipStream_t);                                                    \
  template int launch_ln_patchify<T>(const T*, const float*, const float*, T*, int, int, int, int, float, hipStream_t); \
  template int launch_layernorm_rows<T>(const T*, const float*, const float*, T*, int64_t, int, float, hipStream_t);  \
  template int launch_pool_ln<T>(const T*, const float*, const float*, T*, int, int, int, float, hipStream_t, int, int, int); \
  template int launch_conv3_first<T>(const T*, int64_t, int64_t, int64_t, int64_t, const float*, const float*, T*,    \
                                     int, int, int, bool, int, hipStream_t);                                          \
  template int launch_convt2_small<T>(const T*, const float*, const float*, T*, int, int, int, int, hipStream_t);     \
  template int launch_reparam<T>(const float*, int, const float*, const float*, float*, T*, int, int, hipStream_t);   \
  template int launch_head_tail<T>(const T*, const float*, const float*, float*, int, int, hipStream_t);              \
  template int launch_head_tail_splitk<T>(const float*, int, const float*, int, const float*, const float*, float*, int, int, hipStream_t); \
  template int launch_resize_mse<T>(const T*, const T*, T*, float*, float*, int, hipStream_t);                        \
  template int launch_swin_window_attn<T>(const T*, const float*, T*, int, int, int, int, int, int, hipStream_t);     \
  template int launch_patch_merge_ln<T>(const T*, const float*, const float*, T*, int, int, int, int, float, hipStream_t); \
  template int launch_mean_tokens<T>(const T*, T*, int, int, int, hipStream_t);                                   \
  template int launch_preprocess<T>(const unsigned char*, T*, int, int, int, hipStream_t);

}  // namespace gcv

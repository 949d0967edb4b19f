// Launcher of the rolling-strip dw7x7 + LayerNorm kernel (dwconv_roll.h); included by dw_{f32,f16,bf16}.hip, which
// are compiled with -fno-slp-vectorize: left alone, hipcc packs the tap FMAs into v_pk_fma_f32 (no faster than two
// v_fma_f32 on gfx950) whose even-aligned register pairs cost ~290 spilled registers in this kernel.
#pragma once
#include <cstdlib>

#include "dwconv_roll.h"
#include "dwconv_mfma.h"
#include "kernels.h"

namespace gcv {

static inline int dw_env_int(const char* name, int dflt) {
  const char* e = exp_env(name);
  return e ? std::atoi(e) : dflt;
}

template <typename T, int C, int NS>
static int launch_dw_roll_cfg(const T* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb, T* y,
                              int nimg, int H, float eps, hipStream_t s) {
  constexpr int NT = DwRollLds<T, C, NS>::NT;
  constexpr int LDS = DwRollLds<T, C, NS>::bytes;
  if (LDS > 64 * 1024) GCV_ENSURE_LDS((dwconv7_ln_roll_kernel<T, C, NS>), LDS);
  // bands: enough workgroups to fill 256 CUs, never fewer than 7 output rows per band unless the image itself is
  // smaller (each band re-reads a 6-row input apron)
  static const int force_bands = dw_env_int("GCV_DW_BANDS", 0);
  const int per_cu = std::max(1, std::min(160 * 1024 / LDS, NT > 512 ? 1 : 2));   // 116 VGPRs: 16 waves per CU
  const int slots = 256 * per_cu;
  int nb = force_bands > 0 ? force_bands : (slots + nimg - 1) / nimg;
  // ... except for launches that 7-row bands would spread over at most half the CUs (batches of 32, the 112-pixel pass):
  // there a band's walk of rows + 6 steps at ~2 us each IS the launch time, so bands shrink to as little as two rows
  // (8 steps instead of 13; the re-read aprons come from L2)
  const int nb7 = std::max(1, H / 7);
  const bool small_launch = nimg * nb7 <= 128;
  nb = std::max(1, std::min(nb, small_launch ? std::max(1, H / 2) : nb7));
  const int band_rows = (H + nb - 1) / nb;
  const int nbands = (H + band_rows - 1) / band_rows;
  GCV_REQUIRE((int64_t)H * 7 * NS * C * (int64_t)sizeof(T) < (int64_t)1 << 31, "dwconv: one image must stay below 2 GiB");
  hipLaunchKernelGGL((dwconv7_ln_roll_kernel<T, C, NS>), dim3(nimg * nbands), dim3(NT), LDS, s, x, wdw, bdw, lnw, lnb, y,
                     H, band_rows, nbands, eps);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

// the matrix-pipe variant (dwconv_mfma.h): same grid and band rule, 12 MFMA waves + 4 staging / LayerNorm waves
template <typename T, int C, int NS>
static int launch_dw_mfma_cfg(const T* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb, T* y,
                              int nimg, int H, float eps, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    typedef DwMfmaLds<T, C, NS> LY;
    constexpr int LDS = LY::bytes;
    if (LDS > 64 * 1024) GCV_ENSURE_LDS((dwconv7_ln_mfma_kernel<T, C, NS>), LDS);
    const int slots = 256 * std::max(1, std::min(160 * 1024 / LDS, 1));
    int nb = (slots + nimg - 1) / nimg;
    const int nb7 = std::max(1, H / 7);
    const bool small_launch = nimg * nb7 <= 128;        // see launch_dw_roll_cfg
    nb = std::max(1, std::min(nb, small_launch ? std::max(1, H / 2) : nb7));
    const int band_rows = (H + nb - 1) / nb;
    const int nbands = (H + band_rows - 1) / band_rows;
    GCV_REQUIRE((int64_t)H * 7 * NS * C * (int64_t)sizeof(T) < (int64_t)1 << 31, "dwconv: one image must stay below 2 GiB");
    hipLaunchKernelGGL((dwconv7_ln_mfma_kernel<T, C, NS>), dim3(nimg * nbands), dim3(LY::NT), LDS, s, x, wdw, bdw, lnw, lnb,
                       y, H, band_rows, nbands, eps);
    GCV_CHECK_HIP(hipGetLastError());
    return 0;
  } else {
    set_error("dwconv7_ln_mfma: 16-bit storage only");
    return -3;
  }
}

// shapes covered: the whole image width (W = 7 * NS) in one workgroup of NS * C <= 768 threads
template <typename T> bool dwconv_roll_applicable(int H, int W, int C) {
  if (W % 7 != 0 || H < 1) return false;
  const int ns = W / 7;
  switch (C) {
    case 96:  return ns == 4 || ns == 8;
    case 192: return ns == 2 || ns == 4;
    case 384: return ns == 1 || ns == 2;
    case 768: return ns == 1;
  }
  return false;
}

template <typename T>
int launch_dwconv7_ln_roll(const T* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb, T* y,
                           int nimg, int H, int W, int C, float eps, hipStream_t s) {
  GCV_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0, "dwconv: x / y 16-byte aligned");
  const int ns = W / 7;
#ifndef GCV_DW_NO_MFMA
  // 16-bit storage, 56-pixel maps at C = 96 (stage 0 of the 224-pixel passes, the largest dw launch): taps on the matrix
  // pipe.  The same kernel is correct at C = 96 / 28 pixels and C = 192 / 28, 14 pixels (GCV_DWM_ALL builds; parity cases
  // green) and wins there in the single-kernel microbenchmark (66 -> 56 us at 256 images), but inside the step rocprofv3
  // shows those three 9 - 33 % SLOWER than the VALU kernel, and the whole step is the same with or without them
  // (17.24k against 17.27k fps in paired runs), so they are not dispatched.
  // Only launches whose bands are at least 14 rows long (batches of 64 images and more at this size): a workgroup of the
  // matrix-pipe kernel first zeroes its ring and builds 42 tap-operand registers per lane from 84 global loads, which a
  // 7-row band does not pay back (vae B = 32 bf16, paired runs: 19.3k fps with it, 19.9k with the VALU kernel there).
  const bool long_bands = (int64_t)nimg * H >= 14 * 256;
#define GCV_DWM(CC, NSS) if (sizeof(T) == 2 && C == CC && ns == NSS && long_bands) return launch_dw_mfma_cfg<T, CC, NSS>(x, wdw, bdw, lnw, lnb, y, nimg, H, eps, s)
  GCV_DWM(96, 8);
#ifdef GCV_DWM_ALL
  GCV_DWM(96, 4);
  GCV_DWM(192, 4); GCV_DWM(192, 2);
#endif
#undef GCV_DWM
#endif
#define GCV_ROLL(CC, NSS) if (C == CC && ns == NSS) return launch_dw_roll_cfg<T, CC, NSS>(x, wdw, bdw, lnw, lnb, y, nimg, H, eps, s)
  GCV_ROLL(96, 8); GCV_ROLL(96, 4);
  GCV_ROLL(192, 4); GCV_ROLL(192, 2);
  GCV_ROLL(384, 2); GCV_ROLL(384, 1);
  GCV_ROLL(768, 1);
#undef GCV_ROLL
  set_error("dwconv7_ln_roll: unsupported shape");
  return -3;
}

#define GCV_INSTANTIATE_DW_ROLL(T)                                                                                   \
  template bool dwconv_roll_applicable<T>(int, int, int);                                                            \
  template int launch_dwconv7_ln_roll<T>(const T*, const float*, const float*, const float*, const float*, T*, int,   \
                                         int, int, int, float, hipStream_t);

}  // namespace gcv

// launcher for gemm_glds_kernel (included by gemm_{f16,bf16}.hip)
#pragma once
#include <cstdlib>

#include "gemm_glds.h"

namespace gcv {

template <typename T, int EPI, int ACT, int BKB> static int launch_glds_bkb(const GemmArgs& g, hipStream_t s) {
  constexpr int SMEM = GldsSmem<T, BKB>::bytes;
  GCV_ENSURE_LDS((gemm_glds_kernel<T, EPI, ACT, BKB>), SMEM);
  const int ntm = cdiv(g.M, kGldsBM), ntn = g.N / kGldsBN;
  hipLaunchKernelGGL((gemm_glds_kernel<T, EPI, ACT, BKB>), dim3(ntm * ntn), dim3(256), SMEM, s, g);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int EPI, int ACT> static int launch_glds_cfg(const GemmArgs& g, hipStream_t s) {
  static const bool line64 = exp_env("GCV_GLDS_ROW64") != nullptr;    // A/B switch: force the 64-byte-row ring
  if (g.K % 64 == 0 && g.K >= 512 && !line64) return launch_glds_bkb<T, EPI, ACT, 128>(g, s);   // (K = 384: 6 stages, the 4-deep ring wins)
  return launch_glds_bkb<T, EPI, ACT, 64>(g, s);
}

template <typename T> bool gemm_glds_applicable(const GemmArgs& g, int a_mode, int epi) {
  static const bool disabled = exp_env("GCV_NO_GLDS") != nullptr;     // A/B switch for profiling
  if (disabled || sizeof(T) != 2 || a_mode != A_PLAIN) return false;
  if (!(epi == EPI_BIAS_ACT && (g.act == ACT_NONE || g.act == ACT_GELU)) && !(epi == EPI_RESID && g.act == ACT_NONE))
    return false;
  if (g.N % kGldsBN != 0 || g.K % 32 != 0 || g.M < 256 || g.lda % 8 != 0 || g.ldc % 8 != 0) return false;
  auto al = [](const void* p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; };
  if (!al(g.A, 16) || !al(g.Wt, 16) || !al(g.C, 16) || (g.bias && !al(g.bias, 16))) return false;
  if (epi == EPI_RESID && (!g.gamma || !g.resid || !al(g.gamma, 16) || !al(g.resid, 8))) return false;
  return true;
}

template <typename T> int launch_gemm_glds(const GemmArgs& g, int epi, hipStream_t s) {
  if (epi == EPI_RESID) return launch_glds_cfg<T, EPI_RESID, ACT_NONE>(g, s);
  if (g.act == ACT_GELU) return launch_glds_cfg<T, EPI_BIAS_ACT, ACT_GELU>(g, s);
  return launch_glds_cfg<T, EPI_BIAS_ACT, ACT_NONE>(g, s);
}

}  // namespace gcv

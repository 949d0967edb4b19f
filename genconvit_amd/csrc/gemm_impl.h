// Tile-config selection + launch for gemm_kernel (included by gemm_{f32,f16,bf16}.hip).
#pragma once
#include "gemm.h"

namespace gcv {

template <typename T, int BM, int BN, int WM, int WN, int AMODE, int EPI>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
  const int ntm = cdiv(g.M, BM), ntn = cdiv(g.N, BN);
  dim3 grid(ntm * ntn, EPI == EPI_SPLITK ? g.splitk : 1, 1);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, AMODE, EPI>), grid, dim3(256), 0, s, g);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T> int launch_gemm(const GemmArgs& g, int a_mode, int epi, hipStream_t s) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int BK = (sizeof(T) == 4) ? 16 : 64;
  GCV_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "empty GEMM");
  GCV_REQUIRE(g.K % EPC == 0, "K must be a multiple of the 16-byte chunk");
  GCV_REQUIRE(aligned16(g.A) && aligned16(g.Wt), "A/Wt must be 16-byte aligned");
  if (a_mode == A_PLAIN) {
    GCV_REQUIRE(g.lda % EPC == 0 && g.lda >= g.K, "lda must be a chunk multiple >= K");
  } else {
    GCV_REQUIRE((1 << g.cin_log2) % EPC == 0 && g.K == 9 * (1 << g.cin_log2), "im2col: K = 9*Cin, Cin chunk-aligned");
    GCV_REQUIRE((g.H % 2) == 0 && (g.W % 2) == 0, "im2col modes need even H, W");
  }
  if (epi == EPI_SPLITK) {
    GCV_REQUIRE(g.splitk >= 1 && g.k_per_split % BK == 0 && (int64_t)g.k_per_split * g.splitk >= g.K && g.partial, "bad split-K plan");
  }
  if (epi == EPI_RESID) GCV_REQUIRE(g.gamma && g.resid, "EPI_RESID needs gamma and resid");
  if (epi == EPI_POOL4) GCV_REQUIRE(g.M % 4 == 0, "EPI_POOL4 needs M % 4 == 0");
  if (epi == EPI_CONVT) GCV_REQUIRE(g.N == 4 << g.cout_log2 && g.M % (g.H * g.W) == 0, "EPI_CONVT shape");

  if (a_mode == A_PLAIN && epi == EPI_BIAS_ACT) {
    if (g.M <= 32) return launch_cfg<T, 32, 128, 32, 32, A_PLAIN, EPI_BIAS_ACT>(g, s);
    if (g.M <= 64) return launch_cfg<T, 64, 128, 32, 64, A_PLAIN, EPI_BIAS_ACT>(g, s);
    if (g.N % 96 == 0) return launch_cfg<T, 128, 96, 32, 96, A_PLAIN, EPI_BIAS_ACT>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, A_PLAIN, EPI_BIAS_ACT>(g, s);
  }
  if (a_mode == A_PLAIN && epi == EPI_RESID) {
    if (g.N % 96 == 0) return launch_cfg<T, 128, 96, 32, 96, A_PLAIN, EPI_RESID>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, A_PLAIN, EPI_RESID>(g, s);
  }
  if (a_mode == A_PLAIN && epi == EPI_CONVT) {
    if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, A_PLAIN, EPI_CONVT>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, A_PLAIN, EPI_CONVT>(g, s);
  }
  if (a_mode == A_PLAIN && epi == EPI_SPLITK) {
    if (g.M <= 32) return launch_cfg<T, 32, 128, 32, 32, A_PLAIN, EPI_SPLITK>(g, s);
    if (g.M <= 64) return launch_cfg<T, 64, 128, 32, 64, A_PLAIN, EPI_SPLITK>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, A_PLAIN, EPI_SPLITK>(g, s);
  }
  if (a_mode == A_IM2COL3_POOL && epi == EPI_POOL4) {
    if (g.N <= 32) return launch_cfg<T, 128, 32, 32, 32, A_IM2COL3_POOL, EPI_POOL4>(g, s);
    if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, A_IM2COL3_POOL, EPI_POOL4>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, A_IM2COL3_POOL, EPI_POOL4>(g, s);
  }
  if (a_mode == A_IM2COL3_S2 && epi == EPI_BIAS_ACT) {
    if (g.N <= 32) return launch_cfg<T, 128, 32, 32, 32, A_IM2COL3_S2, EPI_BIAS_ACT>(g, s);
    if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, A_IM2COL3_S2, EPI_BIAS_ACT>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, A_IM2COL3_S2, EPI_BIAS_ACT>(g, s);
  }
  set_error("launch_gemm: unsupported (a_mode, epilogue) combination");
  return -3;
}

}  // namespace gcv

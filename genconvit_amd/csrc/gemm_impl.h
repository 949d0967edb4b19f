// Tile-config selection + launch for gemm_kernel (included by gemm_{f32,f16,bf16}.hip).
#pragma once
#include "gemm.h"
#include "gemm_glds_impl.h"

namespace gcv {

template <typename T, int BM, int BN, int WM, int WN, int BKB, int AMODE, int EPI, int ACT>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
  constexpr int SMEM = GemmSmem<T, BM, BN, BKB, EPI>::bytes;
  if (SMEM > 64 * 1024) GCV_ENSURE_LDS((gemm_kernel<T, BM, BN, WM, WN, BKB, AMODE, EPI, ACT>), SMEM);
  const int ntm = cdiv(g.M, BM), ntn = cdiv(g.N, BN);
  dim3 grid(ntm * ntn, EPI == EPI_SPLITK ? g.splitk : 1, 1);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, BKB, AMODE, EPI, ACT>), grid, dim3(256), SMEM, s, g);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// default LDS row: 64 B for fp32 (16 k), 128 B for 16-bit (64 k)
template <typename T> struct DefBKB { static constexpr int v = sizeof(T) == 4 ? 64 : 128; };

#define GCV_ACT_SWITCH(CALL)                                     \
  switch (g.act) {                                               \
    case ACT_NONE:  return CALL(ACT_NONE);                       \
    case ACT_RELU:  return CALL(ACT_RELU);                       \
    case ACT_GELU:  return CALL(ACT_GELU);                       \
    case ACT_LEAKY: return CALL(ACT_LEAKY);                      \
    default: set_error("bad activation code"); return -2;        \
  }

#ifndef GCV_IM2COL_SHORTK
#define GCV_IM2COL_SHORTK 1     // 0: A/B builds (64-deep K tiles everywhere, the round-3 dispatch)
#endif
template <typename T> int launch_gemm(const GemmArgs& g, int a_mode, int epi, hipStream_t s) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int B = DefBKB<T>::v;
  constexpr int BK = B / 16 * EPC;
  GCV_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "empty GEMM");
  GCV_REQUIRE(g.K % EPC == 0, "K must be a multiple of the 16-byte chunk");
  GCV_REQUIRE(g.N % 4 == 0, "N must be a multiple of 4");
  GCV_REQUIRE(aligned16(g.A) && aligned16(g.Wt), "A/Wt must be 16-byte aligned");
  if (epi != EPI_SPLITK) {
    GCV_REQUIRE(g.C && (reinterpret_cast<uintptr_t>(g.C) % (4 * sizeof(T))) == 0, "C must be aligned to 4 elements");
    GCV_REQUIRE(epi == EPI_CONVT || (g.ldc % 4 == 0 && g.ldc >= g.N), "ldc must be a multiple of 4 and >= N");
    GCV_REQUIRE(!g.bias || aligned16(g.bias), "bias must be 16-byte aligned");
  }
  if (a_mode == A_PLAIN) {
    GCV_REQUIRE(g.lda % EPC == 0 && g.lda >= g.K, "lda must be a chunk multiple >= K");
  } else {
    GCV_REQUIRE((1 << g.cin_log2) % EPC == 0 && g.K == 9 * (1 << g.cin_log2), "im2col: K = 9*Cin, Cin chunk-aligned");
    GCV_REQUIRE((g.H % 2) == 0 && (g.W % 2) == 0, "im2col modes need even H, W");
  }
  if (epi == EPI_SPLITK) {
    GCV_REQUIRE(g.splitk >= 1 && g.k_per_split % BK == 0 && (int64_t)g.k_per_split * g.splitk >= g.K && g.partial &&
                aligned16(g.partial), "bad split-K plan");
  }
  if (epi == EPI_RESID) GCV_REQUIRE(g.gamma && g.resid && aligned16(g.gamma), "EPI_RESID needs gamma and resid");
  if (epi == EPI_POOL4) GCV_REQUIRE(g.M % 4 == 0, "EPI_POOL4 needs M % 4 == 0");
  if (epi == EPI_CONVT) GCV_REQUIRE(g.N == 4 << g.cout_log2 && g.cout_log2 >= 2 && g.M % (g.H * g.W) == 0, "EPI_CONVT shape");
  if constexpr (sizeof(T) == 2) {
    if (gemm_glds_applicable<T>(g, a_mode, epi)) return launch_gemm_glds<T>(g, epi, s);
  }
  // 16-bit GEMMs whose K leaves a tail of <= 32 use 64-byte LDS rows (K tile 32) instead of padding a 64-k tile
  const bool short_k = sizeof(T) == 2 && (g.K % 64) != 0 && (g.K % 64) <= 32;

  if (a_mode == A_PLAIN && epi == EPI_BIAS_ACT) {
#define C1(A) launch_cfg<T, 32, 128, 32, 32, B, A_PLAIN, EPI_BIAS_ACT, A>(g, s)
#define C2(A) launch_cfg<T, 64, 128, 32, 64, B, A_PLAIN, EPI_BIAS_ACT, A>(g, s)
#define C3(A) launch_cfg<T, 128, 96, 32, 96, B, A_PLAIN, EPI_BIAS_ACT, A>(g, s)
#define C3S(A) launch_cfg<T, 128, 96, 32, 96, 64, A_PLAIN, EPI_BIAS_ACT, A>(g, s)
#define C4(A) launch_cfg<T, 128, 128, 64, 64, B, A_PLAIN, EPI_BIAS_ACT, A>(g, s)
    // small problems (the classifier heads: M = frames <= 256, N = 500 / 1000): with 128-row tiles they are 4-16
    // workgroups on a 256-CU chip and take 20-50 us of pure latency; 32- / 64-row tiles spread the same work over 4x / 2x
    // as many CUs
    const int t128 = cdiv(g.M, 128) * cdiv(g.N, 128);
    if (g.M <= 32 || t128 < 48) { GCV_ACT_SWITCH(C1) }
    if (g.M <= 64 || t128 < 96) { GCV_ACT_SWITCH(C2) }
    if (g.N % 96 == 0) {
      if (short_k) { GCV_ACT_SWITCH(C3S) }
      GCV_ACT_SWITCH(C3)
    }
    GCV_ACT_SWITCH(C4)
#undef C1
#undef C2
#undef C3
#undef C3S
#undef C4
  }
  if (a_mode == A_PLAIN && epi == EPI_RESID) {
    GCV_REQUIRE(g.act == ACT_NONE, "EPI_RESID has no activation");
    if (g.N % 96 == 0) return launch_cfg<T, 128, 96, 32, 96, B, A_PLAIN, EPI_RESID, ACT_NONE>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, B, A_PLAIN, EPI_RESID, ACT_NONE>(g, s);
  }
  if (a_mode == A_PLAIN && epi == EPI_CONVT) {
    GCV_REQUIRE(g.act == ACT_RELU || g.act == ACT_LEAKY, "EPI_CONVT is built for ReLU / LeakyReLU");
    if (g.N <= 64) {
      if (short_k) {
        if (g.act == ACT_RELU) return launch_cfg<T, 128, 64, 32, 64, 64, A_PLAIN, EPI_CONVT, ACT_RELU>(g, s);
        return launch_cfg<T, 128, 64, 32, 64, 64, A_PLAIN, EPI_CONVT, ACT_LEAKY>(g, s);
      }
      if (g.act == ACT_RELU) return launch_cfg<T, 128, 64, 32, 64, B, A_PLAIN, EPI_CONVT, ACT_RELU>(g, s);
      return launch_cfg<T, 128, 64, 32, 64, B, A_PLAIN, EPI_CONVT, ACT_LEAKY>(g, s);
    }
    if (g.act == ACT_RELU) return launch_cfg<T, 128, 128, 64, 64, B, A_PLAIN, EPI_CONVT, ACT_RELU>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, B, A_PLAIN, EPI_CONVT, ACT_LEAKY>(g, s);
  }
  if (a_mode == A_PLAIN && epi == EPI_SPLITK) {
    if (g.M <= 32) return launch_cfg<T, 32, 128, 32, 32, B, A_PLAIN, EPI_SPLITK, ACT_NONE>(g, s);
    if (g.M <= 64) return launch_cfg<T, 64, 128, 32, 64, B, A_PLAIN, EPI_SPLITK, ACT_NONE>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, B, A_PLAIN, EPI_SPLITK, ACT_NONE>(g, s);
  }
  if (a_mode == A_IM2COL3_POOL && epi == EPI_POOL4) {
    GCV_REQUIRE(g.act == ACT_RELU, "conv3x3+pool is built for ReLU");
    // K = 144 / 288 (16 / 32 input channels): 32-deep K tiles instead of a 64-deep one that is three quarters / half padding
    // (round 4, paired runs on one box: ed.enc 0.200 -> 0.171 ms, vae.enc 0.077 -> 0.068 ms per step)
    if (short_k && GCV_IM2COL_SHORTK) {
      if (g.N <= 32) return launch_cfg<T, 128, 32, 32, 32, 64, A_IM2COL3_POOL, EPI_POOL4, ACT_RELU>(g, s);
      if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, 64, A_IM2COL3_POOL, EPI_POOL4, ACT_RELU>(g, s);
    }
    if (g.N <= 32) return launch_cfg<T, 128, 32, 32, 32, B, A_IM2COL3_POOL, EPI_POOL4, ACT_RELU>(g, s);
    if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, B, A_IM2COL3_POOL, EPI_POOL4, ACT_RELU>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, B, A_IM2COL3_POOL, EPI_POOL4, ACT_RELU>(g, s);
  }
  if (a_mode == A_IM2COL3_S2 && epi == EPI_BIAS_ACT) {
    GCV_REQUIRE(g.act == ACT_LEAKY, "conv3x3 stride 2 is built for LeakyReLU");
    if (short_k && GCV_IM2COL_SHORTK) {
      if (g.N <= 32) return launch_cfg<T, 128, 32, 32, 32, 64, A_IM2COL3_S2, EPI_BIAS_ACT, ACT_LEAKY>(g, s);
      if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, 64, A_IM2COL3_S2, EPI_BIAS_ACT, ACT_LEAKY>(g, s);
    }
    if (g.N <= 32) return launch_cfg<T, 128, 32, 32, 32, B, A_IM2COL3_S2, EPI_BIAS_ACT, ACT_LEAKY>(g, s);
    if (g.N <= 64) return launch_cfg<T, 128, 64, 32, 64, B, A_IM2COL3_S2, EPI_BIAS_ACT, ACT_LEAKY>(g, s);
    return launch_cfg<T, 128, 128, 64, 64, B, A_IM2COL3_S2, EPI_BIAS_ACT, ACT_LEAKY>(g, s);
  }
  set_error("launch_gemm: unsupported (a_mode, epilogue) combination");
  return -3;
}

}  // namespace gcv

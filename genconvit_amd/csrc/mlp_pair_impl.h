// launchers + weight packers of the C = 384 MLP kernel pair (included by mlp_pair_{f16,bf16}.hip)
#pragma once
#include "mlp_pair.h"

namespace gcv {

template <typename T, typename S> int launch_pack_w1_frag(const S* w1, T* out, int C, hipStream_t s) {
  GCV_REQUIRE(C % 32 == 0, "pack_w1_frag: C must be a multiple of 32");
  const int64_t total = (int64_t)4 * C * C;
  hipLaunchKernelGGL((pack_w1_frag_kernel<T, S>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w1, out, C);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}
template <typename T, typename S> int launch_pack_w2_frag(const S* w2, const float* gamma, T* out, int C, hipStream_t s) {
  GCV_REQUIRE(C % 32 == 0 && gamma, "pack_w2_frag: C must be a multiple of 32, gamma is folded into the packed weight");
  const int64_t total = (int64_t)4 * C * C;
  hipLaunchKernelGGL((pack_w2_frag_kernel<T, S>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w2, gamma, out, C);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

// Work split of pw1: a workgroup costs its x tile (~3 chunk times) plus its hidden chunks; workgroups beyond the CU
// count queue behind the first round (one workgroup per CU: the ring takes the whole LDS)
static inline int xs_pw1_pick_split(int ntile, int nkc) {
  int best = 1;
  long best_cost = -1;
  for (int ns = 1; ns <= 24; ++ns) {
    if (nkc % ns || (nkc / ns) % 2) continue;             // whole pairs of chunks per workgroup (two per barrier)
    const long rounds = ((long)ntile * ns + 255) / 256;
    const long cost = rounds * (nkc / ns + 3);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
  }
  return best;
}

static inline int mlp_pair_check(const MlpPairArgs& a, int C) {
  GCV_REQUIRE(a.M > 0 && a.X && a.W1f && a.W2f && a.b1 && a.b2 && a.gamma && a.resid && a.out && a.hidden,
              "MLP pair: null argument");
  GCV_REQUIRE(mlp_pair_supported(C), "MLP pair is built for C = 384");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  GCV_REQUIRE(al(a.X) && al(a.W1f) && al(a.W2f) && al(a.b1) && al(a.resid) && al(a.out) && al(a.hidden),
              "MLP pair: operands must be 16-byte aligned");
  GCV_REQUIRE((int64_t)cdiv(a.M, 32) * (4 * C / 32) * 2048 < ((int64_t)1 << 31),
              "MLP pair: hidden tensor exceeds 32-bit buffer offsets");
  return 0;
}

template <typename T, int C> static int launch_xs_pw1_c(const MlpPairArgs& a, hipStream_t s) {
  constexpr int NKC = 4 * C / 32;
  constexpr int NW = 8, D = 6;
  constexpr int SMEM = XsPw1Smem<C, NW, D>::bytes;
  GCV_ENSURE_LDS((xs_pw1_kernel<T, C, NW, D>), SMEM);
  const int ntile = cdiv(a.M, 32 * NW);
  const int ns = xs_pw1_pick_split(ntile, NKC);
  GCV_REQUIRE((NKC / ns) % 2 == 0, "xs_pw1: an even number of hidden chunks per workgroup (two per barrier)");
  hipLaunchKernelGGL((xs_pw1_kernel<T, C, NW, D>), dim3(ntile, ns), dim3(NW * 64), SMEM, s, a, NKC / ns);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int C, int BN, int D, int TB> static int launch_pw2f_cfg(const MlpPairArgs& a, hipStream_t s) {
  constexpr int SMEM = Pw2fSmem<C, BN, D, TB>::bytes;
  GCV_ENSURE_LDS((pw2f_kernel<T, C, BN, D, TB>), SMEM);
  hipLaunchKernelGGL((pw2f_kernel<T, C, BN, D, TB>), dim3(cdiv(a.M, 32 * TB) * (C / BN)), dim3(512), SMEM, s, a);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

// Ring depths: stages of one 32-deep K chunk + (D - 1) KB per wave of residual staging in 160 KB of LDS — 256 x 384 tiles
// 3 x 40 KB, 256 x 192 tiles 4 x 28 KB, 128 x 192 tiles 5 x 20 KB
template <typename T, int C> static int launch_pw2f_c(const MlpPairArgs& a, hipStream_t s) {
  const int ntm = cdiv(a.M, 256);
  // full-width tiles halve the operand bytes per FLOP but need >= ~0.7 x 256 of them to keep the chip busy
  if (ntm >= 180) return launch_pw2f_cfg<T, C, C, 3, 8>(a, s);
  if (ntm * (C / 192) >= 128) return launch_pw2f_cfg<T, C, 192, 4, 8>(a, s);
  // a few thousand tokens (batches of 32, the 112-pixel pass): 128-token tiles, twice the workgroups, each with half the
  // MFMAs per K chunk
  return launch_pw2f_cfg<T, C, 192, 5, 4>(a, s);
}

template <typename T> int launch_xs_pw1(const MlpPairArgs& a, int C, hipStream_t s) {
  GCV_TRY(mlp_pair_check(a, C));
  return launch_xs_pw1_c<T, 384>(a, s);
}
template <typename T> int launch_pw2f(const MlpPairArgs& a, int C, hipStream_t s) {
  GCV_TRY(mlp_pair_check(a, C));
  return launch_pw2f_c<T, 384>(a, s);
}
template <typename T> int launch_mlp_pair(const MlpPairArgs& a, int C, hipStream_t s) {
  GCV_TRY(launch_xs_pw1<T>(a, C, s));
  return launch_pw2f<T>(a, C, s);
}

}  // namespace gcv

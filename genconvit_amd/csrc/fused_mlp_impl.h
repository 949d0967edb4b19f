// launcher + weight packer for fused_mlp_kernel (included by fused_mlp_{f16,bf16}.hip)
#pragma once
#include <cstdlib>

#include "fused_mlp.h"
#include "fused_mlp_res.h"
#ifdef GCV_EXPERIMENTS
#include "diag/fused_mlp_ring.h"      // opt-in LDS-DMA ring MLP (C = 192 / 384): measured slower, kept for A/B runs only
#endif

namespace gcv {

// W2 (C, 4C) fp32 row-major -> [4C/HC][C][HC] in T; inside every 16 hidden indices bits 2 and 3 are
// swapped so that a lane's GEMM2 A-operand fragment (k = 16kk + 8(j>>2) + 4h + (j&3)) is one 16-byte chunk.
template <typename T>
__global__ void __launch_bounds__(256) pack_w2_chunks_kernel(const float* __restrict__ w2, T* __restrict__ out, int C,
                                                             int HC) {
  const int64_t total = (int64_t)C * 4 * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int p = (int)(i % HC);
  const int64_t t = i / HC;
  const int o = (int)(t % C);
  const int ch = (int)(t / C);
  const int kk = p >> 4, h = (p >> 3) & 1, jj = p & 7;
  const int hid = 16 * kk + 8 * (jj >> 2) + 4 * h + (jj & 3);
  out[i] = from_f<T>(w2[(int64_t)o * 4 * C + ch * HC + hid]);
}

// The LDS-DMA ring kernel (fused_mlp_ring.h, GCV_EXPERIMENTS builds only) wants W2 packed in 32-wide hidden groups, the
// streaming kernel in 96-wide chunks; it is opt-in there (GCV_FUSED_MLP384=1 at C = 384, GCV_MLP_RING192=1 at C = 192).
static inline bool mlp_use_ring(int C) {
#ifdef GCV_EXPERIMENTS
  static const bool ring192 = [] { const char* e = exp_env("GCV_MLP_RING192"); return e ? std::atoi(e) != 0 : false; }();
  return C == 384 || (C == 192 && ring192);
#else
  (void)C;
  return false;
#endif
}
static inline int mlp_chunk_width(int C) { return mlp_use_ring(C) ? 32 : kMlpHC; }

template <typename T> int launch_pack_w2_chunks(const float* w2_dev, T* out, int C, hipStream_t s) {
  const int HC = mlp_chunk_width(C);
  GCV_REQUIRE((4 * C) % HC == 0 && HC % 16 == 0, "hidden width must be a multiple of the chunk");
  const int64_t total = (int64_t)C * 4 * C;
  hipLaunchKernelGGL((pack_w2_chunks_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w2_dev, out, C, HC);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int C, int NW> static int launch_mlp_c(const MlpArgs& a, hipStream_t s) {
  constexpr int SMEM = MlpSmem<T, C, NW>::bytes;
  GCV_ENSURE_LDS((fused_mlp_kernel<T, C, NW>), SMEM);
  hipLaunchKernelGGL((fused_mlp_kernel<T, C, NW>), dim3(cdiv(a.M, NW * 32)), dim3(NW * 64), SMEM, s, a);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T> int launch_fused_mlp_res(const MlpArgs& a, hipStream_t s) {
  constexpr int SMEM = MlpResSmem::bytes;
  GCV_ENSURE_LDS((fused_mlp_res_kernel<T>), SMEM);
  static const int nw = [] { const char* e = exp_env("GCV_MLP_RES_WAVES"); return e ? std::atoi(e) : 8; }();
  const int wave_tiles = cdiv(a.M, 32);
  const int nwg = cdiv(wave_tiles, nw) < 256 ? cdiv(wave_tiles, nw) : 256;    // one persistent workgroup per CU
  if (a.lnp_nseg > 0) {                                    // last block of the stage: LayerNorm2d + space-to-depth epilogue
    GCV_REQUIRE(a.lnp_w && a.lnp_b && a.lnp_nseg <= 4, "fused MLP: LN-patchify epilogue arguments");
    GCV_ENSURE_LDS((fused_mlp_res_kernel<T, true>), SMEM);
    hipLaunchKernelGGL((fused_mlp_res_kernel<T, true>), dim3(nwg), dim3(64 * nw), SMEM, s, a);
  } else {
    hipLaunchKernelGGL((fused_mlp_res_kernel<T>), dim3(nwg), dim3(64 * nw), SMEM, s, a);
  }
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

#ifdef GCV_EXPERIMENTS
template <typename T, int C> static int launch_mlp_ring_c(const MlpArgs& a, hipStream_t s) {
  constexpr int SMEM = MlpRingSmem<C>::bytes;
  GCV_ENSURE_LDS((fused_mlp_ring_kernel<T, C>), SMEM);
  const int ntiles = cdiv(a.M, 128);
  const int slots = 256 * (C == 192 ? 2 : 1);              // persistent workgroups: two per CU at C = 192
  hipLaunchKernelGGL((fused_mlp_ring_kernel<T, C>), dim3(ntiles < slots ? ntiles : slots), dim3(256), SMEM, s, a, ntiles);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T> int launch_fused_mlp_ring(const MlpArgs& a, int C, hipStream_t s) {
  if (C == 192) return launch_mlp_ring_c<T, 192>(a, s);
  if (C == 384) return launch_mlp_ring_c<T, 384>(a, s);
  set_error("ring MLP kernel: C = 192 / 384");
  return -3;
}

#endif

template <typename T> int launch_fused_mlp(const MlpArgs& a, int C, hipStream_t s) {
  GCV_REQUIRE(a.M > 0 && a.X && a.W1 && a.W2c && a.b1 && a.b2 && a.gamma && a.resid && a.out, "fused MLP: null argument");
  // C=96: 4-wave workgroups (81 KB LDS -> two independent workgroups per CU overlap each other's
  // prologue / epilogue); C=192: the double-buffered chunks fill the LDS, one 8-wave workgroup per CU
  // C=96 with enough tokens to give every wave of the chip several tiles: weights resident in LDS, no barriers
  static const int res_mode = [] { const char* e = exp_env("GCV_MLP_RESIDENT"); return e ? std::atoi(e) : 1; }();
  if (res_mode && fused_mlp_res_applies(C, a.M)) return launch_fused_mlp_res<T>(a, s);
  GCV_REQUIRE(a.lnp_nseg == 0, "fused MLP: the LN-patchify epilogue exists in the LDS-resident kernel only");
  if (C == 96) return launch_mlp_c<T, 96, 4>(a, s);
#ifdef GCV_EXPERIMENTS
  if (mlp_use_ring(C)) return launch_fused_mlp_ring<T>(a, C, s);
#endif
#ifdef GCV_EXPERIMENTS
  if (C == 192) return launch_mlp_c<T, 192, 8>(a, s);   // round-2 streaming kernel: the product runs xs_mlp_kernel at C = 192
#endif
  set_error("fused MLP kernels of this file: C = 96, 192");
  return -3;
}

}  // namespace gcv

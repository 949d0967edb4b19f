// launcher + weight packer of the x-stationary fused MLP (included by xs_mlp_{f16,bf16}.hip)
#pragma once
#include "xs_mlp.h"

namespace gcv {

template <typename T, typename S> int launch_pack_xs_mlp(const T* w1, const S* w2, T* out, int C, hipStream_t s) {
  GCV_REQUIRE(xs_mlp_supported(C), "pack_xs_mlp: unsupported C");
  const int64_t total = (int64_t)xs_mlp_packed_elems(C);
  hipLaunchKernelGGL((pack_xs_mlp_kernel<T, S>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w1, w2, out, C);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int C> static int launch_xs_mlp_c(const XsMlpArgs& a, hipStream_t s) {
  constexpr int SMEM = XsMlpCfg<C>::bytes;
  GCV_ENSURE_LDS((xs_mlp_kernel<T, C>), SMEM);
  const int npass = cdiv(a.M, 256);
  const int nwg = npass < 256 ? npass : 256;               // one persistent workgroup per CU
  if (a.lnp_nseg > 0) {                                    // last block of the stage: LayerNorm2d + space-to-depth epilogue
    GCV_REQUIRE(a.lnp_w && a.lnp_b && a.lnp_nseg <= 4, "xs MLP: LN-patchify epilogue arguments");
    GCV_ENSURE_LDS((xs_mlp_kernel<T, C, true>), SMEM);
    hipLaunchKernelGGL((xs_mlp_kernel<T, C, true>), dim3(nwg), dim3(512), SMEM, s, a, npass);
    GCV_CHECK_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((xs_mlp_kernel<T, C>), dim3(nwg), dim3(512), SMEM, s, a, npass);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T> int launch_xs_mlp(const XsMlpArgs& a, int C, hipStream_t s) {
  GCV_REQUIRE(a.M > 0 && a.X && a.Wp && a.b1 && a.b2 && a.gamma && a.resid && a.out, "xs MLP: null argument");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  GCV_REQUIRE(al(a.X) && al(a.Wp) && al(a.resid) && al(a.out), "xs MLP: operands must be 16-byte aligned");
#ifdef GCV_EXPERIMENTS
  if (C == 96) return launch_xs_mlp_c<T, 96>(a, s);
#endif
  if (C == 192) return launch_xs_mlp_c<T, 192>(a, s);
  set_error("xs MLP is built for C = 192");
  return -3;
}

}  // namespace gcv

// explicit instantiation of the non-GEMM kernels for storage dtype float
#include "kernels_impl.h"
namespace gcv { GCV_INSTANTIATE_KERNELS(float) }
namespace gcv {
int launch_kl(const float* partial, int splitk, const float* bias, const float* mu, float* rowsum, float* kl, int B,
              int N, hipStream_t s) {
  hipLaunchKernelGGL(kl_rows_kernel, dim3(B), dim3(256), 0, s, partial, splitk, bias, mu, rowsum, B, N);
  GCV_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(64), 0, s, rowsum, kl, B);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_vote(const float* logits, int rows, float* mean2, hipStream_t s) {
  hipLaunchKernelGGL(vote_kernel, dim3(1), dim3(256), 0, s, logits, rows, mean2);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_vote_segments(const float* logits, int B, int nets, const int* off, int nvid, float* mean2, hipStream_t s) {
  GCV_REQUIRE(logits && off && mean2 && B > 0 && nvid > 0 && (nets == 1 || nets == 2), "vote_segments: bad arguments");
  hipLaunchKernelGGL(vote_segments_kernel, dim3(nvid), dim3(64), 0, s, logits, B, nets, off, mean2);
  GCV_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace gcv

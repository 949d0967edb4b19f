// explicit instantiation of the MFMA GEMM family for storage dtype half_t
#include "gemm_impl.h"
namespace gcv { template int launch_gemm<half_t>(const GemmArgs&, int, int, hipStream_t); }

#if GCV_GLDS_STAMPS
extern "C" __attribute__((visibility("default"))) int gcv_debug_read_glds_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gcv::gcv_glds_stamps), sizeof(unsigned long long) * n);
}
#endif

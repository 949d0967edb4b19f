// fused ConvNeXt MLP kernel for storage dtype half_t
#include "fused_mlp_impl.h"
namespace gcv {
template int launch_fused_mlp<half_t>(const MlpArgs&, int, hipStream_t);
template int launch_pack_w2_chunks<half_t>(const float*, half_t*, int, hipStream_t);
}

#if GCV_MLP_STAMPS
extern "C" __attribute__((visibility("default"))) int gcv_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gcv::gcv_mlp_stamps), sizeof(unsigned long long) * n);
}
#endif

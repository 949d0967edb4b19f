// C = 384 MLP kernel pair (x-stationary pw1 + fragment-major pw2) for storage dtype half_t
#include "mlp_pair_impl.h"
namespace gcv {
template int launch_xs_pw1<half_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_pw2f<half_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_mlp_pair<half_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_pack_w1_frag<half_t, half_t>(const half_t*, half_t*, int, hipStream_t);
template int launch_pack_w2_frag<half_t, half_t>(const half_t*, half_t*, int, hipStream_t);
template int launch_pack_w2_frag<half_t, float>(const float*, half_t*, int, hipStream_t);
}

#if GCV_XS_STAMPS
extern "C" __attribute__((visibility("default"))) int gcv_debug_read_xs_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gcv::gcv_xs_stamps), sizeof(unsigned long long) * n);
}
#endif

#if GCV_P2_STAMPS
extern "C" __attribute__((visibility("default"))) int gcv_debug_read_p2_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gcv::gcv_p2_stamps), sizeof(unsigned long long) * n);
}
#endif

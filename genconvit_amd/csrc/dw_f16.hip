// rolling-strip dw7x7 + LayerNorm for storage dtype half_t (own TU: built with -fno-slp-vectorize)
#include "dwconv_roll_impl.h"
namespace gcv { GCV_INSTANTIATE_DW_ROLL(half_t) }

#if GCV_DW_STAMPS
extern "C" __attribute__((visibility("default"))) int gcv_debug_read_dw_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gcv::gcv_dw_stamps), sizeof(unsigned long long) * n);
}
#endif

// x-stationary fused ConvNeXt MLP (C = 96 / 192) for storage dtype half_t
#include "xs_mlp_impl.h"
namespace gcv {
template int launch_xs_mlp<half_t>(const XsMlpArgs&, int, hipStream_t);
template int launch_pack_xs_mlp<half_t, half_t>(const half_t*, const half_t*, half_t*, int, hipStream_t);
template int launch_pack_xs_mlp<half_t, float>(const half_t*, const float*, half_t*, int, hipStream_t);
}

GCV_XM_STAMP_READER      // (diag/diag.h: nothing unless the build defines GCV_XM_STAMPS)

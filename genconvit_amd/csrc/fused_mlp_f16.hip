// fused ConvNeXt MLP kernel for storage dtype half_t
#include "fused_mlp_impl.h"
namespace gcv {
template int launch_fused_mlp<half_t>(const MlpArgs&, int, hipStream_t);
template int launch_pack_w2_chunks<half_t>(const float*, half_t*, int, hipStream_t);
}

GCV_MLP_STAMP_READER      // (diag/diag.h: nothing unless the build defines GCV_MLP_STAMPS)

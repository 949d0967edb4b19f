// C = 384 MLP kernel pair (x-stationary pw1 + fragment-major pw2) for storage dtype half_t
#include "mlp_pair_impl.h"
namespace gcv {
template int launch_xs_pw1<half_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_pw2f<half_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_mlp_pair<half_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_pack_w1_frag<half_t, half_t>(const half_t*, half_t*, int, hipStream_t);
template int launch_pack_w2_frag<half_t, float>(const float*, const float*, half_t*, int, hipStream_t);
}

GCV_XS_STAMP_READER      // (diag/diag.h: nothing unless the build defines GCV_XS_STAMPS / GCV_P2_STAMPS)
GCV_P2_STAMP_READER

// C = 384 MLP kernel pair (x-stationary pw1 + fragment-major pw2) for storage dtype bf16_t
#include "mlp_pair_impl.h"
namespace gcv {
template int launch_xs_pw1<bf16_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_pw2f<bf16_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_mlp_pair<bf16_t>(const MlpPairArgs&, int, hipStream_t);
template int launch_pack_w1_frag<bf16_t, bf16_t>(const bf16_t*, bf16_t*, int, hipStream_t);
template int launch_pack_w2_frag<bf16_t, float>(const float*, const float*, bf16_t*, int, hipStream_t);
}

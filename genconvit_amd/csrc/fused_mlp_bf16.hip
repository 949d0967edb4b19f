// fused ConvNeXt MLP kernel for storage dtype bf16_t
#include "fused_mlp_impl.h"
namespace gcv {
template int launch_fused_mlp<bf16_t>(const MlpArgs&, int, hipStream_t);
template int launch_pack_w2_chunks<bf16_t>(const float*, bf16_t*, int, hipStream_t);
}

// x-stationary fused ConvNeXt MLP (C = 96 / 192) for storage dtype bf16_t
#include "xs_mlp_impl.h"
namespace gcv {
template int launch_xs_mlp<bf16_t>(const XsMlpArgs&, int, hipStream_t);
template int launch_pack_xs_mlp<bf16_t, bf16_t>(const bf16_t*, const bf16_t*, bf16_t*, int, hipStream_t);
template int launch_pack_xs_mlp<bf16_t, float>(const bf16_t*, const float*, bf16_t*, int, hipStream_t);
}

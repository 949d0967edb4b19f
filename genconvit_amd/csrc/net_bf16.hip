// NetImpl instantiation for storage dtype bf16_t
#include "net_impl.h"
namespace gcv { NetBase* make_net_bf16() { return new NetImpl<bf16_t>(); } }

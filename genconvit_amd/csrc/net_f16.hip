// NetImpl instantiation for storage dtype half_t
#include "net_impl.h"
namespace gcv { NetBase* make_net_f16() { return new NetImpl<half_t>(); } }

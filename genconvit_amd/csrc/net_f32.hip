// NetImpl instantiation for storage dtype float
#include "net_impl.h"
namespace gcv { NetBase* make_net_f32() { return new NetImpl<float>(); } }

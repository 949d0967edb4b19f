// explicit instantiation of the non-GEMM kernels for storage dtype half_t
#include "kernels_impl.h"
namespace gcv { GCV_INSTANTIATE_KERNELS(half_t) }

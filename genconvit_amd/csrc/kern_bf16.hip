// explicit instantiation of the non-GEMM kernels for storage dtype bf16_t
#include "kernels_impl.h"
namespace gcv { GCV_INSTANTIATE_KERNELS(bf16_t) }

// explicit instantiation of the MFMA GEMM family for storage dtype float
#include "gemm_impl.h"
namespace gcv { template int launch_gemm<float>(const GemmArgs&, int, int, hipStream_t); }

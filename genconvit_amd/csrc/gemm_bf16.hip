// explicit instantiation of the MFMA GEMM family for storage dtype bf16_t
#include "gemm_impl.h"
namespace gcv { template int launch_gemm<bf16_t>(const GemmArgs&, int, int, hipStream_t); }

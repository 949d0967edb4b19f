// explicit instantiation of the MFMA GEMM family for storage dtype half_t
#include "gemm_impl.h"
namespace gcv { template int launch_gemm<half_t>(const GemmArgs&, int, int, hipStream_t); }

GCV_GLDS_STAMP_READER      // (diag/diag.h: nothing unless the build defines GCV_GLDS_STAMPS)

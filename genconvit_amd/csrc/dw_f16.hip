// rolling-strip dw7x7 + LayerNorm for storage dtype half_t (own TU: built with -fno-slp-vectorize)
#include "dwconv_roll_impl.h"
namespace gcv { GCV_INSTANTIATE_DW_ROLL(half_t) }

GCV_DW_STAMP_READER      // (diag/diag.h: nothing unless the build defines GCV_DW_STAMPS)

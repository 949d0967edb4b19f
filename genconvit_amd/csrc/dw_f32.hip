// rolling-strip dw7x7 + LayerNorm for storage dtype float (own TU: built with -fno-slp-vectorize)
#include "dwconv_roll_impl.h"
namespace gcv { GCV_INSTANTIATE_DW_ROLL(float) }

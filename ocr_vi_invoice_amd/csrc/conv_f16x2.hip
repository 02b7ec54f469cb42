// conv_gemm instantiation for f16x2_t (one TU per dtype: parallel compile).
#include "conv_launch.h"
namespace ocrvi {
OCRVI_RANGE_FLAG_TU()   // binds this unit's f16x2 range-flag pointer (common.h)

template int launch_conv<f16x2_t>(const ConvParams&, int, hipStream_t);
}

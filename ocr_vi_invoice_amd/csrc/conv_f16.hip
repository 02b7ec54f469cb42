// conv_gemm instantiation for f16_t (one TU per dtype: parallel compile).
#include "conv_launch.h"
namespace ocrvi {
template int launch_conv<f16_t>(const ConvParams&, int, hipStream_t);
}

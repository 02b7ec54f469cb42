// conv_gemm instantiation for bf16_t (one TU per dtype: parallel compile).
#include "conv_launch.h"
namespace ocrvi {
template int launch_conv<bf16_t>(const ConvParams&, int, hipStream_t);
}

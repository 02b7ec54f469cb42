// conv_gemm instantiation for float (one TU per dtype: parallel compile).
#include "conv_launch.h"
namespace ocrvi {
template int launch_conv<float>(const ConvParams&, int, hipStream_t);
}

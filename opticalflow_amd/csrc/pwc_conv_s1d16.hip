// conv3x3 MFMA kernels, stride 1 dilation 16 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d16(const ConvArgs &a) { return dispatch<1, 16, 2, 3>(a); }
}  // namespace pwc_conv

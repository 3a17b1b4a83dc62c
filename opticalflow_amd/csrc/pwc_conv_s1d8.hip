// conv3x3 MFMA kernels, stride 1 dilation 8 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d8(const ConvArgs &a) { return dispatch<1, 8, 2, 4>(a); }
}  // namespace pwc_conv

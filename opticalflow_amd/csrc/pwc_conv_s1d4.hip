// conv3x3 MFMA kernels, stride 1 dilation 4 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d4(const ConvArgs &a) { return dispatch<1, 4, 4, 4>(a); }
}  // namespace pwc_conv

// conv3x3 MFMA kernels, stride 2 dilation 1 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s2d1(const ConvArgs &a) { return dispatch<2, 1, 2, 4>(a); }
}  // namespace pwc_conv

// conv3x3 MFMA kernels, stride 1 dilation 2 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d2(const ConvArgs &a) { return dispatch<1, 2, 4, 4>(a); }
}  // namespace pwc_conv

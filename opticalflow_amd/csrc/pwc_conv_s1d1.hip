// conv3x3 MFMA kernels, stride 1 dilation 1 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d1(const ConvArgs &a) { return a.ksplit > 1 ? launch_split<1, 1>(a) : dispatch<1, 1, 4, 4>(a); }
}  // namespace pwc_conv

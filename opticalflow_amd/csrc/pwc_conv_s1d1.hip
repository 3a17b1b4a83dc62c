// conv3x3 MFMA kernels, stride 1 dilation 1 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d1(const ConvArgs &a) {
    if (a.ksplit > 1) return launch_split<1, 1>(a);
    static const bool use16 = [] { const char *e = getenv("PWC_CONV16"); return !(e && *e == '0'); }();
    if (a.Cout <= 16 && use16) return dispatch16(a);
    return dispatch<1, 1, 4, 4>(a);
}
}  // namespace pwc_conv

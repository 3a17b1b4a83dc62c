// conv3x3 MFMA kernels, stride 2 dilation 1 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s2d1(const ConvArgs &a) {
    if (fold_tile(a.Ho, a.Wo)) return launch<1, 1, 2, 1, 1, 1>(a);     // outputs of at most 16 columns (conv6aa): folded 8 x 16 tile
    return dispatch<2, 1, 2, 4>(a);
}
}  // namespace pwc_conv

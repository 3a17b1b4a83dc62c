// conv3x3 MFMA kernels, stride 1 dilation 1 (see pwc_conv_mfma.h)
#include "pwc_conv_mfma.h"
namespace pwc_conv {
int run_s1d1(const ConvArgs &a) {
    const bool fold = fold_tile(a.Ho, a.Wo);        // maps of at most 16 columns: 8 x 16 pixel tiles (pyramid / decoder level 6)
    if (a.ksplit > 1) return fold ? launch_split<1, 1, 1>(a) : launch_split<1, 1>(a);
    static const bool use16 = [] { const char *e = getenv("PWC_CONV16"); return !(e && *e == '0'); }();
    if (a.Cout <= 16 && use16) return dispatch16(a);
    if (fold) return launch<1, 1, 1, 1, 1, 1>(a);   // two workgroups per CU (4-channel chunks), as the tile model picks for these layers
    return dispatch<1, 1, 4, 4>(a);
}
}  // namespace pwc_conv

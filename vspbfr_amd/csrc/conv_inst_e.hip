// Instantiation unit E: (1) dilation-group fused kernels (four dilated branches over one shared input patch),
// (2) variants whose hoisted patch covers 16 words per lane and channel (1024-word planes: dilation 8 halos), so that the
// big-halo groups of a SMART / LargeConv launch do not fall onto the slow direct tail path.
#include "conv_kernel.h"
namespace vspconv {
extern const Cfg kCfgsE[] = {
    VSP_CFGD(4, 4, 4, 16, 0, 2),   // 4 groups x 16 co x 256 pixels
    VSP_CFGD(4, 4, 8, 16, 0, 2),
    VSP_CFGD(4, 4, 4, 16, 2, 2),   // LDS-DMA patch staging: the shared 32x32 patch arrives asynchronously
    VSP_CFGD(4, 4, 4, 16, 2, 3),
    VSP_CFGD(4, 4, 8, 16, 2, 2),
    VSP_CFGM(4, 4, 1, 4, 4, 1, 16, 0, 3),
    VSP_CFGM(4, 4, 1, 4, 4, 1, 16, 0, 4),
    VSP_CFGM(4, 4, 1, 4, 8, 1, 16, 0, 2),
    VSP_CFGM(4, 4, 1, 4, 8, 1, 16, 0, 3),
    VSP_CFGM(2, 4, 1, 4, 8, 1, 16, 0, 2),
    VSP_CFGM(2, 4, 1, 4, 8, 1, 16, 0, 3),
    VSP_CFGM(2, 4, 1, 4, 4, 1, 16, 0, 4),
    VSP_CFGM(1, 8, 1, 4, 4, 1, 16, 0, 4),
    VSP_CFGM(1, 8, 1, 4, 8, 1, 16, 0, 3),
    VSP_CFGM(4, 4, 2, 2, 8, 1, 16, 1, 1),
    VSP_CFGM(4, 2, 2, 2, 8, 1, 16, 1, 1),
};
extern const int kNumE = sizeof(kCfgsE) / sizeof(kCfgsE[0]);
}  // namespace vspconv

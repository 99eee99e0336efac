// Instantiation unit C of the conv tile configurations (split so that hipcc compiles them in parallel).
#include "conv_kernel.h"
namespace vspconv {
extern const Cfg kCfgsC[] = {
    VSP_CFG(4, 4, 1, 4, 8, 1, 6, 0, 4),
    VSP_CFG(4, 4, 1, 4, 4, 1, 6, 0, 4),
    VSP_CFG(4, 4, 2, 2, 8, 1, 6, 0, 4),
    VSP_CFG(4, 2, 2, 2, 8, 1, 12, 0, 2),
    VSP_CFG(2, 8, 1, 4, 8, 1, 6, 0, 4),
    VSP_CFG(1, 8, 1, 4, 8, 1, 12, 0, 2),
    VSP_CFG(1, 8, 1, 4, 4, 1, 12, 0, 2),
    VSP_CFG(4, 1, 1, 4, 8, 1, 12, 0, 2),
    VSP_CFG(2, 1, 4, 1, 8, 1, 12, 0, 2),
    VSP_CFG(2, 4, 1, 4, 8, 1, 12, 0, 2),
    VSP_CFG(1, 4, 1, 4, 8, 1, 12, 0, 2),
    VSP_CFG(1, 1, 4, 1, 8, 1, 12, 0, 2),
    // small maps (32x32 ... 4x4 with 256-512 channels): 64-pixel tiles, K split over wave slices
    VSP_CFG(4, 4, 1, 1, 32, 4, 2, 1, 1),   // 64 co x 64 pix, 4 k-slices
    VSP_CFG(4, 4, 1, 1, 32, 4, 2, 0, 2),
    VSP_CFG(4, 4, 1, 1, 16, 4, 2, 0, 2),
    VSP_CFG(4, 2, 1, 2, 16, 2, 2, 1, 1),   // 64 co x 64 pix, 2 pixel waves x 2 k-slices
    VSP_CFG(4, 2, 1, 2, 16, 2, 2, 0, 2),
    VSP_CFG(4, 1, 1, 4, 16, 2, 2, 0, 2),   // 8 waves: 4 pixel waves x 2 k-slices
    VSP_CFG(4, 1, 1, 4, 16, 2, 2, 1, 1),
    VSP_CFG(2, 4, 2, 1, 32, 4, 2, 0, 1),   // 8 waves: 64 co (2 wave rows) x 64 pix, 4 k-slices
    VSP_CFG(4, 1, 1, 4, 8, 2, 2, 0, 3),
    VSP_CFG(4, 1, 1, 4, 8, 2, 2, 0, 4),
    VSP_CFG(4, 1, 1, 4, 16, 4, 2, 0, 1),   // 16 waves: 4 pixel waves x 4 k-slices
    VSP_CFG(2, 1, 1, 4, 16, 2, 2, 0, 2),   // 32 co x 64 pix, 8 waves
    VSP_CFG(2, 1, 1, 4, 16, 2, 2, 0, 4),
    VSP_CFG(2, 2, 1, 4, 16, 2, 4, 0, 2),   // 32 co x 128 pix, 8 waves
    VSP_CFG(4, 2, 1, 4, 16, 2, 4, 0, 2),   // 64 co x 128 pix, 8 waves
};
extern const int kNumC = sizeof(kCfgsC) / sizeof(kCfgsC[0]);
}  // namespace vspconv

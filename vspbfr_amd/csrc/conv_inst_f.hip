// Instantiation unit F: 8-wave (512-thread) workgroups on full-size tiles -- twice the waves per staged byte, for the
// layers where 4-wave blocks leave the matrix pipe waiting on staging.
#include "conv_kernel.h"
namespace vspconv {
extern const Cfg kCfgsF[] = {
    VSP_CFG(4, 4, 1, 4, 16, 2, 6, 0, 2),   // 64 co x 256 pix, 2 k-slices
    VSP_CFG(4, 4, 1, 4, 8, 2, 6, 0, 2),
    VSP_CFG(4, 4, 1, 4, 8, 2, 6, 0, 3),
    VSP_CFG(4, 4, 2, 4, 8, 1, 6, 0, 2),    // 128 co x 256 pix
    VSP_CFG(4, 4, 2, 4, 4, 1, 6, 0, 2),
    VSP_CFG(2, 4, 2, 4, 8, 1, 6, 0, 2),    // 64 co x 256 pix, 32 co per wave row
    VSP_CFG(2, 4, 2, 4, 8, 1, 6, 0, 4),
    VSP_CFG(4, 2, 2, 4, 8, 1, 12, 0, 2),   // 128 co x 128 pix
    VSP_CFG(4, 4, 2, 2, 8, 2, 6, 0, 2),    // 128 co x 128 pix, 2 k-slices
};
extern const int kNumF = sizeof(kCfgsF) / sizeof(kCfgsF[0]);
}  // namespace vspconv

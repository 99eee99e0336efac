// Instantiation unit D: stride-2 transposed 3x3 convolution kernels (one pass over the four sub-pixel phases).
#include "conv_kernel.h"
namespace vspconv {
extern const Cfg kCfgsD[] = {
    VSP_CFGT(4, 1, 4, 8, 6, 0, 3),   // 64 co x 64 input positions (256 outputs)
    VSP_CFGT(4, 1, 4, 4, 6, 0, 4),
    VSP_CFGT(4, 1, 4, 4, 6, 0, 3),
    VSP_CFGT(4, 1, 4, 8, 6, 0, 2),
    VSP_CFGT(4, 2, 2, 8, 6, 0, 2),   // 128 co x 32 positions
    VSP_CFGT(2, 1, 4, 8, 6, 0, 4),   // 32 co x 64 positions
    VSP_CFGT(1, 1, 4, 8, 6, 0, 4),   // 16 co
    VSP_CFGT(4, 1, 1, 8, 6, 0, 2),   // 64 co x 16 positions (tiny maps)
    VSP_CFGT(2, 4, 1, 8, 6, 0, 3),   // 128 co x 16 positions
};
extern const int kNumD = sizeof(kCfgsD) / sizeof(kCfgsD[0]);
}  // namespace vspconv

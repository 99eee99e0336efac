// Instantiation unit D: stride-2 transposed 3x3 convolution kernels (one pass over the four sub-pixel phases).
// NB = 4 * NP: NP groups of 16 input positions per wave, each with its four phases.
#include "conv_kernel.h"
namespace vspconv {
extern const Cfg kCfgsD[] = {
    VSP_CFGT(4, 4, 1, 4, 8, 6, 0, 3),   // 64 co x 64 input positions (256 outputs)
    VSP_CFGT(4, 4, 1, 4, 4, 6, 0, 4),
    VSP_CFGT(4, 4, 1, 4, 4, 6, 0, 3),
    VSP_CFGT(4, 4, 1, 4, 8, 6, 0, 2),
    VSP_CFGT(4, 4, 2, 2, 8, 6, 0, 2),   // 128 co x 32 positions
    VSP_CFGT(2, 4, 1, 4, 8, 6, 0, 4),   // 32 co x 64 positions
    VSP_CFGT(1, 4, 1, 4, 8, 6, 0, 4),   // 16 co
    VSP_CFGT(4, 4, 1, 1, 8, 6, 0, 2),   // 64 co x 16 positions (tiny maps)
    VSP_CFGT(2, 4, 4, 1, 8, 6, 0, 3),   // 128 co x 16 positions
    VSP_CFGT(4, 8, 1, 4, 8, 6, 0, 2),   // 64 co x 128 positions, 128 accumulator registers
    VSP_CFGT(4, 8, 1, 4, 4, 6, 0, 2),
    VSP_CFGT(2, 8, 1, 4, 8, 6, 0, 3),   // 32 co x 128 positions
    VSP_CFGT(2, 8, 1, 4, 8, 6, 0, 4),
    VSP_CFGT(2, 8, 2, 4, 8, 6, 0, 2),   // 8 waves: 64 co x 128 positions
    VSP_CFGT(2, 8, 2, 4, 4, 6, 0, 2),
    VSP_CFGT(2, 16, 1, 4, 8, 6, 0, 2),  // 32 co x 256 positions
    VSP_CFGT(1, 16, 1, 4, 8, 6, 0, 3),  // 16 co x 256 positions
    VSP_CFGT(1, 16, 2, 4, 8, 6, 0, 2),  // 8 waves: 32 co x 256 positions
    VSP_CFGT(4, 4, 1, 4, 8, 3, 1, 2),   // register prefetch of the next chunk (PF = 1)
    VSP_CFGT(4, 4, 1, 4, 8, 3, 1, 1),
    VSP_CFGT(4, 8, 1, 4, 8, 3, 1, 1),
    VSP_CFGT(2, 8, 1, 4, 8, 3, 1, 2),
    VSP_CFGT(4, 4, 1, 4, 16, 6, 0, 2),  // 16-channel chunks: half the barriers per MFMA
    VSP_CFGT(4, 4, 1, 4, 16, 6, 0, 3),
    VSP_CFGT(2, 8, 1, 4, 16, 6, 0, 2),
};
extern const int kNumD = sizeof(kCfgsD) / sizeof(kCfgsD[0]);
}  // namespace vspconv

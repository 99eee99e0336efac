// Instantiation unit G: asynchronous staging (PF = 2): the next chunk's input patch goes global -> LDS by LDS-DMA
// (global_load_lds) into a second patch buffer while the current chunk is multiplied; weights prefetch through registers.
#include "conv_kernel.h"
namespace vspconv {
extern const Cfg kCfgsG[] = {
    VSP_CFG(4, 4, 1, 4, 8, 1, 6, 2, 2),
    VSP_CFG(4, 4, 1, 4, 8, 1, 6, 2, 3),
    VSP_CFG(4, 4, 1, 4, 4, 1, 6, 2, 3),
    VSP_CFG(4, 4, 1, 4, 4, 1, 6, 2, 4),
    VSP_CFG(4, 4, 2, 2, 8, 1, 6, 2, 2),
    VSP_CFG(4, 2, 2, 2, 8, 1, 6, 2, 2),
    VSP_CFG(2, 4, 1, 4, 8, 1, 6, 2, 3),
    VSP_CFG(2, 4, 1, 4, 4, 1, 6, 2, 4),
    VSP_CFG(1, 8, 1, 4, 4, 1, 6, 2, 4),
    VSP_CFG(1, 8, 1, 4, 8, 1, 6, 2, 3),
    VSP_CFG(4, 1, 1, 4, 8, 1, 6, 2, 3),
    VSP_CFG(4, 1, 1, 4, 16, 2, 2, 2, 2),
    VSP_CFGT(4, 4, 1, 4, 8, 3, 2, 2),
    VSP_CFGT(4, 4, 1, 4, 8, 3, 2, 3),
    VSP_CFGT(2, 8, 1, 4, 8, 3, 2, 2),
    VSP_CFGT(2, 8, 1, 4, 8, 3, 2, 3),
};
extern const int kNumG = sizeof(kCfgsG) / sizeof(kCfgsG[0]);
}  // namespace vspconv

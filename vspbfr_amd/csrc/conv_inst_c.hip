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
};
extern const int kNumC = sizeof(kCfgsC) / sizeof(kCfgsC[0]);
}  // namespace vspconv

// Error state + version / device probes of the C ABI.
#include "vsp_common.h"
#include <cstring>

namespace vsp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace vsp

extern "C" {

int vsp_abi_version(void) { return VSP_ABI_VERSION; }

const char* vsp_last_error(void) { return vsp::g_err; }

int vsp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

/* sizeof() of the ABI structs, so that a foreign-language binding can verify its own layout at load time. */
int vsp_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(vsp_fir_epilogue);
    case 1: return (int)sizeof(vsp_conv_params);
    case 2: return (int)sizeof(vsp_gemm_params);
    case 3: return (int)sizeof(vsp_tacc_block);
    case 4: return (int)sizeof(vsp_tacc_chain_params);
    case 5: return (int)sizeof(vsp_conv_wgrad_params);
    default: return -1;
  }
}

}  // extern "C"

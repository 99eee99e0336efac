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

/* Stream-ordered flags (round 6): one 64-bit signal word that one stream writes and another waits on -- the pipeline's way of letting the
 * sampler chain of the NEXT batch (side stream) run only underneath the phases of the main stream that leave the chip idle.  A wait may be
 * enqueued BEFORE the write that satisfies it is enqueued (an event cannot do that). */
int vsp_signal_alloc(void** sig) {
  VSP_REQUIRE(sig != nullptr, "signal_alloc: null pointer");
  int dev = 0, can = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess || !can)
    return vsp::fail(VSP_ENOTSUP, "signal_alloc: this device does not support hipStreamWaitValue32");
  hipError_t e = hipExtMallocWithFlags(sig, 8, hipMallocSignalMemory);
  if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "signal_alloc: %s", hipGetErrorString(e));
  e = hipMemset(*sig, 0, 8);
  if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "signal_alloc: memset: %s", hipGetErrorString(e));
  return VSP_OK;
}

int vsp_signal_free(void* sig) {
  if (sig && hipFree(sig) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "signal_free failed");
  return VSP_OK;
}

int vsp_stream_wait_geq32(void* sig, uint32_t value, vsp_stream_t stream) {
  VSP_REQUIRE(sig != nullptr, "stream_wait_geq32: null signal");
  hipError_t e = hipStreamWaitValue32(vsp::as_stream(stream), sig, value, hipStreamWaitValueGte, 0xffffffffu);
  if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "hipStreamWaitValue32: %s", hipGetErrorString(e));
  return VSP_OK;
}

int vsp_stream_write32(void* sig, uint32_t value, vsp_stream_t stream) {
  VSP_REQUIRE(sig != nullptr, "stream_write32: null signal");
  hipError_t e = hipStreamWriteValue32(vsp::as_stream(stream), sig, value, 0);
  if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "hipStreamWriteValue32: %s", hipGetErrorString(e));
  return VSP_OK;
}

}  // extern "C"

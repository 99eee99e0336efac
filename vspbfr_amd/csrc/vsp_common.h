// Shared host-side helpers for libvspbfr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include "../../include/vspbfr_hip.h"

namespace vsp {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(VSP_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return VSP_OK;
}

// Tuning / A-B switches (VSP_WINO_RS, VSP_BF16_PAIR, VSP_*_WGS, the work-order bits of VSP_CONV_DBG ...) are read ONLY when VSP_TUNE=1 is
// set as well: a stray variable in a production environment cannot change which kernel a launch takes (tools/ set VSP_TUNE=1 themselves).
inline const char* tune_env(const char* name) {
  static const bool on = [] { const char* t = getenv("VSP_TUNE"); return t && t[0] && !(t[0] == '0' && !t[1]); }();
  return on ? getenv(name) : nullptr;
}

inline hipStream_t as_stream(vsp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: one flag bit per device ordinal (a process that drives a
// second GPU, or two threads racing through a first call, must not launch a > 64 KB kernel without it; setting it twice is harmless).
struct LdsAttrOnce {
  std::atomic<uint64_t> done{0};
  int ensure(const void* fn, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return fail(VSP_ELAUNCH, "%s: hipGetDevice failed", what);
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return VSP_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(VSP_ELAUNCH, "%s: cannot reserve %d bytes of LDS: %s", what, bytes, hipGetErrorString(e));
    if (dev < 64) done.fetch_or(bit, std::memory_order_release);
    return VSP_OK;
  }
};

// MI355X: 256 CUs; memory-bound grids are capped at 8 resident 256-thread blocks per CU and grid-strided.
constexpr int kNumCU = 256;
constexpr int kMaxStreamBlocks = kNumCU * 8;

}  // namespace vsp

#define VSP_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return vsp::fail(VSP_EINVAL, __VA_ARGS__);     \
  } while (0)

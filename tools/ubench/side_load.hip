// Synthetic side-stream loads for tools/bench_side_load.py (round 6): what does a stream of small kernels cost the big convolutions it runs
// under -- the kernel BOUNDARIES (command processor, cache invalidate / write-back at every dispatch), the workgroup SLOTS, or the L2 traffic?
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void empty_kernel(int* sink) {
  if (sink && threadIdx.x == 1023 && blockIdx.x == 0x7fffffff) *sink = 1;
}

// every thread streams `n16` 16-byte words of `src` (read-only, L2 traffic) and spins `spin` iterations
__global__ void stream_kernel(const uint4* __restrict__ src, int n16_per_wg, int spin, int* sink) {
  uint4 acc = {0, 0, 0, 0};
  const uint4* p = src + (int64_t)blockIdx.x * n16_per_wg;
  for (int i = threadIdx.x; i < n16_per_wg; i += blockDim.x) {
    const uint4 v = p[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  float f = 1.f;
  for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.5f;
  if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u && f == 3.f) *sink = 1;
}

extern "C" int side_load(int launches, int wgs, int threads, const void* src, int n16_per_wg, int spin, int* sink, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < launches; ++i) {
    if (src) stream_kernel<<<wgs, threads, 0, s>>>(static_cast<const uint4*>(src), n16_per_wg, spin, sink);
    else if (spin) stream_kernel<<<wgs, threads, 0, s>>>(nullptr, 0, spin, sink);
    else empty_kernel<<<wgs, threads, 0, s>>>(sink);
  }
  return (int)hipGetLastError();
}

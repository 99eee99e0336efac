// fp32 MFMA issue-rate micro-benchmark (GPU box): v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flops) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  launch(); hipDeviceSynchronize();
  hipEventRecord(s);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  return flops * 5 / (ms * 1e-3) / 1e12;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  const int iters = 4000;
  for (int bpc = 1; bpc <= 4; ++bpc) {   // blocks (4 waves) per CU = waves per SIMD
    int grid = 256 * bpc;
    double f16 = 2.0 * 16 * 16 * 4 * 16 /*acc*/ * iters * 4 /*waves*/ * (double)grid;
    double t16 = run([&] { k16<16><<<grid, 256>>>(out, iters, 1.f, 2.f); }, f16);
    double f16b = 2.0 * 16 * 16 * 4 * 4 * iters * 4 * (double)grid;
    double t16b = run([&] { k16<4><<<grid, 256>>>(out, iters, 1.f, 2.f); }, f16b);
    double f32 = 2.0 * 32 * 32 * 2 * 4 * iters * 4 * (double)grid;
    double t32 = run([&] { k32<4><<<grid, 256>>>(out, iters, 1.f, 2.f); }, f32);
    printf("waves/SIMD %d: 16x16x4 (16 acc) %.1f TF | 16x16x4 (4 acc) %.1f TF | 32x32x2 (4 acc) %.1f TF\n", bpc, t16, t16b, t32);
  }
  return 0;
}

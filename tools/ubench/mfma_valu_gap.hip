// Round 5 microbenchmark: what one wave per SIMD pays for VALU fillers between fp32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles each).
// 256 threads per block (one wave per SIMD), 16 independent accumulators in AccVGPRs, K fillers after every MFMA, several filler kinds.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_gap.hip -o tools/ubench/mfma_valu_gap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

// clustered form: G MFMAs back to back, then G * K fillers in one run
template <int K, int KIND, int G>
__global__ __launch_bounds__(256, 1) void cluster_kernel(unsigned long long* out, float* sink, int iters, float c0) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  f32x2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b + i};
  f32x2 cs = {c0, c0};
  __shared__ f32x4 lbuf[256];
  lbuf[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
  const unsigned ldsaddr = threadIdx.x * 16;
  const f32x4* gptr = reinterpret_cast<const f32x4*>(sink) + 4 + threadIdx.x;
  f32x4 ld[4] = {};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += G) {
#pragma unroll
      for (int m = m0; m < m0 + G; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K * G; ++k) {
        f32x2& x = v[(m0 * K + k) % 8];
        f32x2& y = v[(m0 * K + k + 3) % 8];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[0]) : "v"(y[0]), "v"(cs[0]));
        if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(cs));
        if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(cs));
        if (KIND == 8) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[0]) : "v"(y[0]));
        if (KIND == 9) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(cs));
        if (KIND == 15) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0][0]) : "v"(cs[0]));                  // one dependent chain
        if (KIND == 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(cs));
        if (KIND == 17) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[k % 2]) : "v"(cs));                   // two interleaved chains
        if (KIND == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[k % 4]) : "v"(cs));                   // four
        if (KIND == 10) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[k % 4]) : "v"(ldsaddr));
        if (KIND == 11) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[k % 4]) : "v"(gptr));
        if (KIND == 12) asm volatile("ds_write_b128 %0, %1" :: "v"(ldsaddr), "v"(ld[0]));
        if (KIND == 13) asm volatile("ds_read_b32 %0, %1" : "=v"(ld[k % 4][0]) : "v"(ldsaddr));
        if (KIND == 14) asm volatile("global_load_dword %0, %1, off" : "=v"(ld[k % 4][0]) : "v"(gptr));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");   // (memory fillers are never waited for inside the loop: issue cost only)
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][3];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int K, int KIND, int G>
void runc(const char* name, unsigned long long* d_out, float* d_sink) {
  const int iters = 2000;
  cluster_kernel<K, KIND, G><<<256, 256>>>(d_out, d_sink, iters, 0.75f);
  cluster_kernel<K, KIND, G><<<256, 256>>>(d_out, d_sink, iters, 0.75f);
  (void)hipDeviceSynchronize();
  unsigned long long h[4];
  (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s %d fillers per MFMA, runs after every %d MFMAs: %.1f cycles per MFMA\n", name, K, G, (double)h[0] / (iters * 16.0));
}

template <int K, int KIND>
__global__ __launch_bounds__(256, 1) void gap_kernel(unsigned long long* out, float* sink, int iters, float c0) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  float cs = c0;    // filler constant in a VGPR / SGPR
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float& x = v[(m * K + k) % 8];
        float& y = v[(m * K + k + 3) % 8];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(cs));             // VOP3, registers only
        if (KIND == 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(cs));            // VOP2 (4 bytes)
        if (KIND == 2) asm volatile("v_fmamk_f32 %0, %1, 0x3f400000, %0" : "+v"(x) : "v"(y));            // literal (8 bytes)
        if (KIND == 3) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(y));                      // VOP2 add
        if (KIND == 4) asm volatile("s_nop 0");                                                          // SALU-class filler
        if (KIND == 5) asm volatile("v_fmamk_f32 %0, %1, 0x3f400000, %2" : "=v"(x) : "v"(y), "v"(cs));   // literal, fresh destination
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int K, int KIND>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int iters = 2000;
  gap_kernel<K, KIND><<<256, 256>>>(d_out, d_sink, iters, 0.75f);
  gap_kernel<K, KIND><<<256, 256>>>(d_out, d_sink, iters, 0.75f);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s K=%d: %.1f cycles per MFMA slot\n", name, K, (double)h[0] / (iters * 16.0));
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 64); hipMalloc(&d_sink, 1 << 16); hipMemset(d_sink, 0, 1 << 16);
#define ROW(KIND, NAME) run<0, KIND>(NAME, d_out, d_sink); run<1, KIND>(NAME, d_out, d_sink); run<2, KIND>(NAME, d_out, d_sink); run<3, KIND>(NAME, d_out, d_sink); run<4, KIND>(NAME, d_out, d_sink); run<6, KIND>(NAME, d_out, d_sink);
  ROW(0, "v_fma_f32 (VOP3)")
  ROW(1, "v_fmac_f32_e32 (VOP2)")
  ROW(2, "v_fmamk_f32 literal in-place")
  ROW(5, "v_fmamk_f32 literal new dst")
  ROW(3, "v_add_f32_e32")
  ROW(4, "s_nop 0")
#define ROWC(KIND, NAME) runc<1, KIND, 1>(NAME, d_out, d_sink); runc<2, KIND, 1>(NAME, d_out, d_sink); runc<2, KIND, 4>(NAME, d_out, d_sink); runc<2, KIND, 8>(NAME, d_out, d_sink); runc<2, KIND, 16>(NAME, d_out, d_sink); runc<1, KIND, 8>(NAME, d_out, d_sink);
  ROWC(0, "v_fma_f32")
  ROWC(6, "v_pk_fma_f32")
  ROWC(7, "v_pk_mul_f32")
  ROWC(9, "v_pk_add_f32")
  ROWC(8, "v_mov_b32_dpp")
  runc<2, 15, 8>("dependent v_fma_f32", d_out, d_sink); runc<2, 16, 8>("dependent v_pk_fma_f32", d_out, d_sink);
  runc<2, 17, 8>("v_pk_fma_f32, 2 chains", d_out, d_sink); runc<2, 18, 8>("v_pk_fma_f32, 4 chains", d_out, d_sink);
  runc<2, 6, 8>("v_pk_fma_f32 independent", d_out, d_sink);
#define ROWM(KIND, NAME) runc<1, KIND, 1>(NAME, d_out, d_sink); runc<1, KIND, 4>(NAME, d_out, d_sink); runc<1, KIND, 16>(NAME, d_out, d_sink);
  ROWM(10, "ds_read_b128")
  ROWM(13, "ds_read_b32")
  ROWM(12, "ds_write_b128")
  ROWM(11, "global_load_dwordx4")
  ROWM(14, "global_load_dword")
  return 0;
}

// Does a buffer -> LDS DMA load (raw_buffer_load_lds, 16 bytes per lane) write ZERO for an out-of-range lane, or leave the LDS word alone?
// hipcc --offload-arch=gfx950 -O2 -o lds_dma_oob lds_dma_oob.hip && ./lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float buf[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = -7.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  const int off = (lane & 1) ? 0x7ffffff0 : lane * 16;       // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)buf, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}
int main() {
  float *s, *o, h[256];
  (void)hipMalloc(&s, 4096); (void)hipMalloc(&o, 1024);
  float hs[1024]; for (int i = 0; i < 1024; ++i) hs[i] = (float)i;
  (void)hipMemcpy(s, hs, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(s, o, 4096);
  (void)hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  printf("lane 0: %g %g %g %g | lane 1 (out of range): %g %g %g %g | lane 2: %g %g %g %g | lane 3: %g %g %g %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
  return 0;
}

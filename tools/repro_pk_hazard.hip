// Round 5: reproducer attempt for the packed-fp32 miscompare of conv_bf16_rv.hip.
//
// The failing pattern in the SLP-vectorised build: an operand pair that has JUST come back from LDS (ds_read_b64 / ds_read_b128 of a
// small table, half-wave broadcast) feeds v_pk_fma_f32 with op_sel broadcasts right behind the s_waitcnt, while the CU's other waves keep
// the LDS busy (16-byte commits, fragment reads, LDS-DMA); wrong values only in lanes 48-63, only with two workgroups per CU.
// This program runs that instruction pair in isolation under the same kind of LDS traffic and counts bit mismatches against scalar v_fma_f32
// of the same operands, for three variants of the consumer: packed right behind the wait, packed behind s_nop 7, scalar right behind the wait.
//
//   hipcc --offload-arch=gfx950 -O2 tools/repro_pk_hazard.hip -o tools/repro_pk_hazard ; tools/repro_pk_hazard [iters] [wgs_per_cu 1|2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tab_a(int it, int e) { return 1.0f + (float)((it * 7 + e * 3) & 1023) * (1.0f / 1024.0f); }
__device__ __forceinline__ float tab_b(int it, int e) { return 3.0f + (float)((it * 5 + e * 11) & 1023) * (1.0f / 512.0f); }

template <int MODE>   // 0: v_pk_fma_f32 right behind the wait, 1: behind s_nop 7, 2: two v_fma_f32 right behind the wait, 3 / 4: writer of the dead high half in front of the packed op (0 / 2 wait states)
__global__ __launch_bounds__(256, 2) void repro(unsigned* bad, unsigned* first, int iters, const float* gsrc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* tab = reinterpret_cast<f32x4*>(smem);                 // 64 entries {a, b, a', b'}
  f32x4* scratch = reinterpret_cast<f32x4*>(smem + 1024);      // 32 KB of traffic
  float* dma = reinterpret_cast<float*>(smem + 1024 + 32768);  // 4 KB LDS-DMA target
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    if (tid < 64) tab[tid] = f32x4{tab_a(it, tid), tab_b(it, tid), tab_a(it, tid) + 0.5f, tab_b(it, tid) + 0.25f};
    __syncthreads();
    // LDS traffic of the neighbours: 16-byte writes, 4-byte reads, an LDS-DMA in flight
    __builtin_amdgcn_global_load_lds(gsrc + (size_t)((it & 63) * 256 + tid) * 4, reinterpret_cast<f32x4*>(dma) + wave * 64, 16, 0, 0);
    float sink = 0.f;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
      scratch[((k * 256 + tid) * 5 + wave) & 2047] = f32x4{(float)k, (float)tid, (float)it, 1.f};
      sink += reinterpret_cast<const float*>(scratch)[(tid * 33 + k * 257 + it) & 8191];
      // ---- the pair under test: entry k + 4 kh (half-wave broadcast), operands used right behind the wait
      const int e = (k & 3) + 4 * kh + 8 * (k >> 2);
      const float acc0 = 0.5f + (float)((lane + k + it) & 255) * (1.0f / 256.0f), acc1 = acc0 + 0.125f;
      const unsigned addr = (unsigned)(uintptr_t)(e * 16);   // (byte offset of the entry in LDS: the table sits at offset 0 of the dynamic segment)
      f32x2 res;
      f32x2 accp = {acc0, acc1};
      f32x2 t;
      if (MODE == 0) {
        asm volatile("ds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_fma_f32 %0, %2, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                     : "=&v"(res), "=&v"(t) : "v"(accp), "v"(addr) : "memory");
      } else if (MODE == 1) {
        asm volatile("ds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\tv_pk_fma_f32 %0, %2, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                     : "=&v"(res), "=&v"(t) : "v"(accp), "v"(addr) : "memory");
      } else if (MODE == 9) {
        // The failing slot of the kernel with ITS register numbers (round 6, bisection round 3: only the first `op_sel:[0,1,1]` slot of a commit
        // site fails): v[2:3] = {scale0, scale1}, v[8:9] = {shift0, shift1}, data in v[4:7] / v[14:15], in place, the conversions and the
        // lane reads of the kernel in between.
        asm volatile("ds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(res), "=&v"(t) : "v"(accp), "v"(addr) : "memory");
        const unsigned w0 = (__builtin_bit_cast(unsigned, acc0) >> 16) | (__builtin_bit_cast(unsigned, acc1) & 0xffff0000u);
        float o0, o1, o2, o3;
        float s1_ = t[0] + 0.5f * (float)wave, h1_ = t[1] + 0.25f * (float)wave;   // per-wave operands, as in the kernel
        asm volatile("v_sub_f32 v2, 0x40e00000, %5\n\tv_mov_b32 v3, %5\n\tv_sub_f32 v8, 0xc1200000, %6\n\tv_mov_b32 v9, %6\n\t"
                     "v_mov_b32 v12, 0\n\ts_nop 4\n\t"
                     "v_lshlrev_b32 v4, 16, %4\n\tv_and_b32 v5, 0xffff0000, %4\n\tv_lshlrev_b32 v6, 16, %4\n\tv_and_b32 v7, 0xffff0000, %4\n\t"
                     "v_pk_fma_f32 v[4:5], v[4:5], v[2:3], v[8:9] op_sel:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[6:7], v[6:7], v[2:3], v[8:9] op_sel:[0,1,1]\n\t"
                     "v_cvt_pk_bf16_f32 v10, v4, v5\n\t"
                     "v_readlane_b32 s34, v12, 45\n\t"
                     "v_cvt_pk_bf16_f32 v11, v6, v7\n\t"
                     "v_lshlrev_b32 v6, 16, %4\n\tv_and_b32 v7, 0xffff0000, %4\n\tv_lshlrev_b32 v14, 16, %4\n\tv_and_b32 v15, 0xffff0000, %4\n\t"
                     "v_cmp_eq_u32 vcc, 0, v12\n\t"
                     "v_readlane_b32 s35, v12, 46\n\t"
                     "v_pk_fma_f32 v[6:7], v[6:7], v[2:3], v[8:9] op_sel:[0,1,1]\n\t"
                     "v_pk_fma_f32 v[14:15], v[14:15], v[2:3], v[8:9] op_sel:[0,1,1]\n\t"
                     "s_nop 4\n\tv_mov_b32 %0, v4\n\tv_mov_b32 %1, v5\n\tv_mov_b32 %2, v14\n\tv_mov_b32 %3, v15"
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(w0), "v"(s1_), "v"(h1_)
                     : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v14", "v15", "s34", "s35", "vcc");
        const float q0 = __builtin_bit_cast(float, w0 << 16), q1 = __builtin_bit_cast(float, w0 & 0xffff0000u);
        const float y0 = fmaf(q0, s1_, h1_), y1 = fmaf(q1, s1_, h1_);
        const bool ok9 = __builtin_bit_cast(unsigned, o0) == __builtin_bit_cast(unsigned, y0) && __builtin_bit_cast(unsigned, o1) == __builtin_bit_cast(unsigned, y1) &&
                         __builtin_bit_cast(unsigned, o2) == __builtin_bit_cast(unsigned, y0) && __builtin_bit_cast(unsigned, o3) == __builtin_bit_cast(unsigned, y1);
        res = ok9 ? f32x2{fmaf(acc0, tab_a(it, e), tab_b(it, e)), fmaf(acc1, tab_a(it, e), tab_b(it, e))} : f32x2{o0, o1};
      } else if (MODE == 7 || MODE == 8) {
        // Round 6: the form the assembly-level bisection names (tools/rv_asm_variants.py: scalarising ONLY the `op_sel:[0,1,1]` multiply-adds of
        // the SLP build makes the kernel exact): scale and shift are the HIGH halves of two register pairs, selected for BOTH lanes.
        // MODE 7: the pairs straight from LDS; MODE 8: the kernel's sequence -- source 0 produced by a shift / and pair right in front, two
        // packed operations back to back on the same scale / shift pairs.
        asm volatile("ds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(res), "=&v"(t) : "v"(accp), "v"(addr) : "memory");
        // (scale / shift differ from wave to wave -- in the kernel every wave stages its own two channels: a value picked up from another wave's
        //  operand read would go unnoticed with wave-uniform operands)
        const float sW = t[0] + 0.5f * (float)wave, hW = t[1] + 0.25f * (float)wave;
        f32x2 sp = {-7.25f - (float)wave, sW}, hp = {11.5f + (float)wave, hW};      // {junk, scale}, {junk, shift}
        f32x2 r2, r3;
        if (MODE == 7) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=&v"(r2) : "v"(accp), "v"(sp), "v"(hp));
          const float z0 = fmaf(acc0, sW, hW), z1 = fmaf(acc1, sW, hW), u0 = r2[0], u1 = r2[1];
          const bool ok7 = __builtin_bit_cast(unsigned, u0) == __builtin_bit_cast(unsigned, z0) && __builtin_bit_cast(unsigned, u1) == __builtin_bit_cast(unsigned, z1);
          r2 = ok7 ? f32x2{fmaf(acc0, tab_a(it, e), tab_b(it, e)), fmaf(acc1, tab_a(it, e), tab_b(it, e))} : f32x2{u0 + 1.f, u1};
        } else {
          unsigned w0 = (__builtin_bit_cast(unsigned, acc0) >> 16) | (__builtin_bit_cast(unsigned, acc1) & 0xffff0000u);
          float o0, o1, o2, o3;
          asm volatile("v_lshlrev_b32 v60, 16, %4\n\tv_and_b32 v61, 0xffff0000, %4\n\tv_lshlrev_b32 v62, 16, %4\n\tv_and_b32 v63, 0xffff0000, %4\n\t"
                       "v_pk_fma_f32 v[60:61], v[60:61], %5, %6 op_sel:[0,1,1]\n\tv_pk_fma_f32 v[62:63], v[62:63], %5, %6 op_sel:[0,1,1]\n\t"
                       "s_nop 4\n\tv_mov_b32 %0, v60\n\tv_mov_b32 %1, v61\n\tv_mov_b32 %2, v62\n\tv_mov_b32 %3, v63"
                       : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(w0), "v"(sp), "v"(hp) : "v60", "v61", "v62", "v63");
          r2 = f32x2{o0, o1};
          r3 = f32x2{o2, o3};
          // expected for MODE 8: bf16-truncated operands
          const float q0 = __builtin_bit_cast(float, w0 << 16), q1 = __builtin_bit_cast(float, w0 & 0xffff0000u);
          const float y0 = fmaf(q0, sW, hW), y1 = fmaf(q1, sW, hW);
          const float g0_ = r2[0], g1_ = r2[1], g2_ = r3[0], g3_ = r3[1];
          const bool ok8 = __builtin_bit_cast(unsigned, g0_) == __builtin_bit_cast(unsigned, y0) && __builtin_bit_cast(unsigned, g1_) == __builtin_bit_cast(unsigned, y1) &&
                           __builtin_bit_cast(unsigned, g2_) == __builtin_bit_cast(unsigned, y0) && __builtin_bit_cast(unsigned, g3_) == __builtin_bit_cast(unsigned, y1);
          r2 = ok8 ? f32x2{fmaf(acc0, tab_a(it, e), tab_b(it, e)), fmaf(acc1, tab_a(it, e), tab_b(it, e))} : f32x2{g0_, g1_};
        }
        res = r2;
      } else if (MODE >= 3 && MODE <= 6) {
        // The sequence of the SLP build (commit_one, second slot): pk_fma -> v_cvt_pk_bf16_f32 writing the DEAD high register of the next
        // packed op's source pair (op_sel_hi never selects it) -> pk_fma reading that pair.  Fixed registers: v[60:61] = {scale, dead},
        // v[62:63] = {shift, dead}, v[64:65] first product, v[66:67] result.  MODE 4: two wait states between the writer and the packed op.
        asm volatile("ds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(res), "=&v"(t) : "v"(accp), "v"(addr) : "memory");
        float sc_ = t[0], sh_ = t[1];
        f32x2 r2;
        if (MODE >= 5)   // the commit of the kernel sits BETWEEN the MFMAs of a fragment group: four of them in flight in front of the packed sequence
          asm volatile("v_mfma_f32_32x32x16_bf16 v[80:95], v[112:115], v[116:119], v[80:95]\n\t"
                       "v_mfma_f32_32x32x16_bf16 v[96:111], v[112:115], v[116:119], v[96:111]\n\t"
                       "v_mfma_f32_32x32x16_bf16 v[80:95], v[116:119], v[112:115], v[80:95]\n\t"
                       "v_mfma_f32_32x32x16_bf16 v[96:111], v[116:119], v[112:115], v[96:111]"
                       : : : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",
                             "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
                             "v115", "v116", "v117", "v118", "v119");
        if (MODE == 3 || MODE == 5)
          asm volatile("v_mov_b32 v60, %2\n\tv_mov_b32 v62, %3\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v63, 0\n\ts_nop 4\n\t"
                       "v_pk_fma_f32 v[64:65], %1, v[60:61], v[62:63] op_sel_hi:[1,0,0]\n\t"
                       "v_cvt_pk_bf16_f32 v61, v64, v65\n\t"
                       "v_pk_fma_f32 v[66:67], %1, v[60:61], v[62:63] op_sel_hi:[1,0,0]\n\t"
                       "s_nop 4\n\tv_mov_b32 %0, v66\n\tv_mov_b32 %4, v67"
                       : "=&v"(r2[0]), "+v"(accp), "+v"(sc_), "+v"(sh_), "=&v"(r2[1]) : : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        else
          asm volatile("v_mov_b32 v60, %2\n\tv_mov_b32 v62, %3\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v63, 0\n\ts_nop 4\n\t"
                       "v_pk_fma_f32 v[64:65], %1, v[60:61], v[62:63] op_sel_hi:[1,0,0]\n\t"
                       "v_cvt_pk_bf16_f32 v61, v64, v65\n\ts_nop 1\n\t"
                       "v_pk_fma_f32 v[66:67], %1, v[60:61], v[62:63] op_sel_hi:[1,0,0]\n\t"
                       "s_nop 4\n\tv_mov_b32 %0, v66\n\tv_mov_b32 %4, v67"
                       : "=&v"(r2[0]), "+v"(accp), "+v"(sc_), "+v"(sh_), "=&v"(r2[1]) : : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        res = r2;
      } else {
        float r0, r1;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(addr) : "memory");
        r0 = fmaf(acc0, t[0], t[1]);
        r1 = fmaf(acc1, t[0], t[1]);
        res = f32x2{r0, r1};
      }
      const float x0 = fmaf(acc0, tab_a(it, e), tab_b(it, e)), x1 = fmaf(acc1, tab_a(it, e), tab_b(it, e));
      const float g0 = res[0], g1 = res[1];   // (element copies first: __builtin_bit_cast on an ext_vector ELEMENT expression reads element 0 -- hipcc, ROCm 7.2)
      const bool ok = __builtin_bit_cast(unsigned, g0) == __builtin_bit_cast(unsigned, x0) && __builtin_bit_cast(unsigned, g1) == __builtin_bit_cast(unsigned, x1);
      if (!ok) {
        if (nbad == 0 && atomicAdd(&first[0], 1u) < 8) {
          const unsigned slot = atomicAdd(&first[1], 1u) & 7;
          first[8 + slot * 6 + 0] = lane; first[8 + slot * 6 + 1] = it; first[8 + slot * 6 + 2] = __builtin_bit_cast(unsigned, g0);
          first[8 + slot * 6 + 3] = __builtin_bit_cast(unsigned, x0); first[8 + slot * 6 + 4] = __builtin_bit_cast(unsigned, g1);
          first[8 + slot * 6 + 5] = __builtin_bit_cast(unsigned, x1);
        }
        ++nbad;
      }
    }
    if (sink == 123.456f) bad[1] = 1;
    __syncthreads();
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int per_cu = argc > 2 ? atoi(argv[2]) : 2;
  unsigned *bad, *first;
  float* gsrc;
  hipMalloc(&bad, 8); hipMalloc(&first, 4 * 64); hipMalloc(&gsrc, 64 * 256 * 16);
  hipMemset(gsrc, 0, 64 * 256 * 16);
  const size_t lds = per_cu == 2 ? 1024 + 32768 + 4096 : 100 * 1024;   // one workgroup per CU: a request no second one fits beside
  const int grid = 256 * per_cu;
  for (int mode = 0; mode < 10; ++mode) {
    hipMemset(bad, 0, 8); hipMemset(first, 0, 4 * 64);
    auto fn = mode == 0 ? repro<0> : (mode == 1 ? repro<1> : (mode == 2 ? repro<2> : (mode == 3 ? repro<3> : (mode == 4 ? repro<4> : (mode == 5 ? repro<5> : (mode == 6 ? repro<6> : (mode == 7 ? repro<7> : (mode == 8 ? repro<8> : repro<9>))))))));
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, bad, first, iters, gsrc);
    hipError_t e = hipDeviceSynchronize();
    unsigned hb[2], hf[64];
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 4 * 64, hipMemcpyDeviceToHost);
    const double total = (double)grid * 256 * iters * 16;
    printf("mode %d (%s), %d workgroup(s) per CU: %u mismatches of %.3g (%s)\n", mode,
           mode == 0 ? "v_pk_fma_f32 op_sel behind the wait" : (mode == 1 ? "v_pk_fma_f32 behind s_nop 7" : (mode == 2 ? "v_fma_f32 x 2" : (mode == 3 ? "pk_fma, cvt into the dead high half of the pair, pk_fma" : (mode == 4 ? "the same with s_nop 1 before the second pk_fma" : (mode == 5 ? "mode 3 with four MFMAs in flight" : (mode == 6 ? "mode 4 with four MFMAs in flight" : (mode == 7 ? "v_pk_fma_f32 op_sel:[0,1,1] (scale / shift = HIGH halves)" : (mode == 8 ? "shift / and -> two v_pk_fma_f32 op_sel:[0,1,1] back to back" : "the failing commit slot with the kernel's register numbers")))))))), per_cu, hb[0], total,
           hipGetErrorString(e));
    for (unsigned i = 0; i < (hf[1] < 8 ? hf[1] : 8); ++i)
      printf("   lane %u iter %u: lo got %08x want %08x, hi got %08x want %08x\n", hf[8 + i * 6], hf[8 + i * 6 + 1], hf[8 + i * 6 + 2], hf[8 + i * 6 + 3],
             hf[8 + i * 6 + 4], hf[8 + i * 6 + 5]);
  }
  return 0;
}

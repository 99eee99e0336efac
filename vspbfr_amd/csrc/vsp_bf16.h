// bf16 activations in HBM (the "bf16 kernels" configuration, BASELINE configs[2]): element access helpers shared by the kernels
// that read or write them.  A bf16 tensor is raw 16-bit words, dense NCHW like its fp32 twin; rows of odd width (the (2H+1)^2
// planes between a transposed convolution and its blur) make a plane only 2-byte aligned, which gfx950 global / buffer
// accesses take at any width (unaligned access mode): four elements travel as ONE 8-byte access.
// Conversion: f32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32), bf16 -> f32 is a 16-bit shift (exact).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vsp {

typedef uint16_t bf16_t;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // 4 floats, dword aligned
typedef unsigned u32x2h __attribute__((ext_vector_type(2), aligned(2)));  // 4 bf16, halfword aligned

__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ unsigned bf16_pack(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

template <typename T>
struct Elem;

template <>
struct Elem<float> {
  static __device__ __forceinline__ f32x4u load4(const float* p) { return *reinterpret_cast<const f32x4u*>(p); }
  static __device__ __forceinline__ float load1(const float* p) { return *p; }
  static __device__ __forceinline__ void store4(float* p, f32x4u v) { *reinterpret_cast<f32x4u*>(p) = v; }
  static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
};

template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ f32x4u load4(const bf16_t* p) {
    const u32x2h w = *reinterpret_cast<const u32x2h*>(p);
    return f32x4u{bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1])};
  }
  static __device__ __forceinline__ float load1(const bf16_t* p) { return bf16_lo(*p); }
  static __device__ __forceinline__ void store4(bf16_t* p, f32x4u v) {
    *reinterpret_cast<u32x2h*>(p) = u32x2h{bf16_pack(v[0], v[1]), bf16_pack(v[2], v[3])};
  }
  static __device__ __forceinline__ void store1(bf16_t* p, float v) { *p = (bf16_t)(bf16_pack(v, 0.f) & 0xffffu); }
};

}  // namespace vsp

// Every style modulation of a network in TWO launches (round 5).
//
// A StyleGAN2-shaped network evaluates, per modulated convolution, s = EqualLinear(style_l) (B x Cin from one row of the latent) and the
// demodulation coefficients rsqrt(scale^2 sum_ci s^2 wsq + eps) (B x Cout): reference models/RestoreNet.py:211,376-379,467; ~120 launches
// of a few microseconds per batch, each a dependent round trip on the stream that carries the convolutions.  All of them depend only on the
// latent, so they run up front: one launch for every layer's modulation vector, one for every layer's demodulation coefficients, driven by a
// device table of per-layer pointers.  The arithmetic per output is that of gemv_rows_kernel (gemm_small.hip) and demod_kernel (rowops.hip):
// same operand order, same reductions -- bit-identical to the per-layer launches (tests/test_hip_models.py::test_style_plan_matches_layers).
#include "vsp_common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_mov_s(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_gemv(float v) {   // the reduction of gemv_rows_kernel
  v += dpp_mov_s<0xB1>(v);
  v += dpp_mov_s<0x4E>(v);
  v += dpp_mov_s<0x141>(v);
  v += dpp_mov_s<0x140>(v);
  const int iv = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}
__device__ __forceinline__ float wave_sum_demod(float v) {   // the reduction of demod_kernel
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int MR>
__global__ __launch_bounds__(256) void style_mods_kernel(const vsp_style_layer* __restrict__ tab, const float* __restrict__ src, int B, int64_t bstride, int K) {
  const vsp_style_layer L = tab[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= L.cin) return;
  const float* wrow = L.w + (int64_t)n * K + lane * 4;
  const float* arow = src + L.src_off + lane * 4;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const int ki = K / 256;
  for (int i = 0; i < ki; ++i) {
    const float4 w = *reinterpret_cast<const float4*>(wrow + i * 256);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float4 a = *reinterpret_cast<const float4*>(arow + (int64_t)(m < B ? m : 0) * bstride + i * 256);
      acc[m] = fmaf(a.x, w.x, fmaf(a.y, w.y, fmaf(a.z, w.z, fmaf(a.w, w.w, acc[m]))));
    }
  }
  const float bv = L.bias ? L.bias[n] * L.bias_scale : 0.f;
  float out = 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const float v = wave_sum_gemv(acc[m]);
    if (lane == m) out = v;
  }
  if (lane < B) L.mod[(int64_t)lane * L.cin + n] = out * L.alpha + bv;
}

__global__ __launch_bounds__(256) void style_demods_kernel(const vsp_style_layer* __restrict__ tab, int B, float eps) {
  const vsp_style_layer L = tab[blockIdx.y];
  if (!L.demod) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * L.cout) return;
  const int b = (int)(row / L.cout), co = (int)(row % L.cout);
  const float* sp = L.mod + (int64_t)b * L.cin;
  const float* wp = L.wsq + (int64_t)co * L.cin;
  float s = 0.f;
  for (int c = lane; c < L.cin; c += 64) {
    const float st = sp[c];
    s = fmaf(st * st, wp[c], s);
  }
  s = wave_sum_demod(s);
  if (lane == 0) L.demod[row] = rsqrtf(s * L.wscale2 + eps);
}

}  // namespace

extern "C" int vsp_style_plan_f32(const vsp_style_layer* table, int L, const float* src, int B, int64_t bstride, int K, int max_cin, int max_cout,
                                  float eps, vsp_stream_t stream) {
  VSP_REQUIRE(table && src && L >= 1 && L <= 65535, "style_plan: bad table");
  VSP_REQUIRE(B >= 1 && B <= 16, "style_plan: 1 .. 16 samples (got %d)", B);
  VSP_REQUIRE(K >= 512 && K % 256 == 0 && bstride % 4 == 0 && vsp::aligned16(src), "style_plan: style rows must be 16-byte aligned, K a multiple of 256 >= 512");
  VSP_REQUIRE(max_cin >= 1 && max_cout >= 0, "style_plan: bad maxima");
  hipStream_t st = vsp::as_stream(stream);
  const dim3 g1((unsigned)((max_cin + 3) / 4), (unsigned)L);
  if (B <= 8) style_mods_kernel<8><<<g1, 256, 0, st>>>(table, src, B, bstride, K);
  else style_mods_kernel<16><<<g1, 256, 0, st>>>(table, src, B, bstride, K);
  if (max_cout > 0) {
    const dim3 g2((unsigned)(((int64_t)B * max_cout + 3) / 4), (unsigned)L);
    style_demods_kernel<<<g2, 256, 0, st>>>(table, B, eps);
  }
  return vsp::check_launch("style_plan");
}

// fused bias + activation for gfx950.
//
// Semantics follow the reference native op (op/fused_bias_act_kernel.cu:19-65, launch :68-103):
//   out[i] = f(x[i] + b[(i / step_b) % size_b]) * scale, f selected by act*10+grad.
// Design: pure HBM streaming (8 B/elem fp32), so the kernel is a 16-byte-per-lane grid-stride stream.  When the
// plane size (step_b) is a multiple of 4 one float4 never straddles two channels, so the bias is one scalar
// load per float4 (L1/L2 resident: size_b <= a few thousand floats); otherwise a scalar kernel handles the
// 2-D (B, C) style-MLP case where step_b == 1.
#include "vsp_common.h"

namespace {

__device__ __forceinline__ float act_apply(float v, float r, int mode, float alpha) {
  switch (mode) {
    case 30: return v > 0.f ? v : v * alpha;
    case 31: return r > 0.f ? v : v * alpha;
    case 12:
    case 32: return 0.f;
    default: return v;  // 10, 11 and anything unknown: identity (reference `default:` falls into case 10)
  }
}

template <bool HAS_BIAS, bool HAS_REF>
__global__ __launch_bounds__(256) void fba_vec4_kernel(float4* __restrict__ out, const float4* __restrict__ x,
                                                        const float* __restrict__ bias,
                                                        const float4* __restrict__ ref, int64_t n4, int step_b4,
                                                        int size_b, int mode, float alpha, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = x[i];
    if (HAS_BIAS) {
      const float b = bias[(i / step_b4) % size_b];
      v.x += b; v.y += b; v.z += b; v.w += b;
    }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_REF) r = ref[i];
    float4 o;
    o.x = act_apply(v.x, r.x, mode, alpha) * scale;
    o.y = act_apply(v.y, r.y, mode, alpha) * scale;
    o.z = act_apply(v.z, r.z, mode, alpha) * scale;
    o.w = act_apply(v.w, r.w, mode, alpha) * scale;
    out[i] = o;
  }
}

__global__ __launch_bounds__(256) void fba_scalar_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ ref, int64_t n, int step_b,
                                                          int size_b, int mode, float alpha, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = x[i];
    if (bias) v += bias[(i / step_b) % size_b];
    const float r = ref ? ref[i] : 0.f;
    out[i] = act_apply(v, r, mode, alpha) * scale;
  }
}

}  // namespace

extern "C" int vsp_fused_bias_act_f32(float* out, const float* x, const float* bias, const float* ref,
                                       int64_t n, int step_b, int size_b, int act, int grad, float alpha,
                                       float scale, vsp_stream_t stream) {
  VSP_REQUIRE(n >= 0, "fused_bias_act: negative element count");
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "fused_bias_act: null input/output pointer");
  VSP_REQUIRE(!bias || (size_b > 0 && step_b > 0), "fused_bias_act: bias given but size_b=%d step_b=%d", size_b,
              step_b);
  const int mode = act * 10 + grad;
  hipStream_t s = vsp::as_stream(stream);
  const bool vec = (n % 4 == 0) && (!bias || step_b % 4 == 0) && vsp::aligned16(out) && vsp::aligned16(x) &&
                   (!ref || vsp::aligned16(ref));
  if (vec) {
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > vsp::kMaxStreamBlocks) blocks = vsp::kMaxStreamBlocks;
    const int sb4 = bias ? step_b / 4 : 1;
    const int szb = bias ? size_b : 1;
    auto o4 = reinterpret_cast<float4*>(out);
    auto x4 = reinterpret_cast<const float4*>(x);
    auto r4 = reinterpret_cast<const float4*>(ref);
    if (bias && ref)
      fba_vec4_kernel<true, true><<<(int)blocks, 256, 0, s>>>(o4, x4, bias, r4, n4, sb4, szb, mode, alpha, scale);
    else if (bias)
      fba_vec4_kernel<true, false><<<(int)blocks, 256, 0, s>>>(o4, x4, bias, r4, n4, sb4, szb, mode, alpha, scale);
    else if (ref)
      fba_vec4_kernel<false, true><<<(int)blocks, 256, 0, s>>>(o4, x4, bias, r4, n4, sb4, szb, mode, alpha, scale);
    else
      fba_vec4_kernel<false, false><<<(int)blocks, 256, 0, s>>>(o4, x4, bias, r4, n4, sb4, szb, mode, alpha, scale);
  } else {
    int64_t blocks = (n + 255) / 256;
    if (blocks > vsp::kMaxStreamBlocks) blocks = vsp::kMaxStreamBlocks;
    fba_scalar_kernel<<<(int)blocks, 256, 0, s>>>(out, x, bias, ref, n, step_b > 0 ? step_b : 1,
                                                  size_b > 0 ? size_b : 1, mode, alpha, scale);
  }
  return vsp::check_launch("fused_bias_act");
}

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


// ---- NoiseInjection + FusedLeakyReLU of a styled layer in one stream (training forward; the inference path carries the same chain
// in the conv / blur epilogues):  y[b,c,p] = lrelu(x[b,c,p] + nw * noise[b,p] + bias[c], slope) * gain
// (reference models/RestoreNet.py:558-569 then op/fused_act.py:199-233), and the gradient of the scalar noise weight
//   d nw = sum_{b,c,p} gx[b,c,p] * noise[b,p]        (gx = g * m(y), the slope mask of fused_bias_act act=3 grad=1)
namespace {

__global__ __launch_bounds__(256) void noise_bias_act_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                              const float* __restrict__ noise, const float* __restrict__ nw,
                                                              const float* __restrict__ bias, int C, int64_t hw, float slope, float gain) {
  const int64_t plane = blockIdx.y;                    // b * C + c
  const int b = (int)(plane / C), c = (int)(plane - (int64_t)b * C);
  const float w = nw[0], bc = bias ? bias[c] : 0.f;
  const float* xp = x + plane * hw;
  const float* np = noise + (int64_t)b * hw;
  float* yp = y + plane * hw;
  auto f = [&](float v, float n) { v = fmaf(w, n, v) + bc; return (v > 0.f ? v : v * slope) * gain; };
  if ((hw & 3) == 0) {
    for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < hw; i += 4 * 256 * (int64_t)gridDim.x) {
      const float4 v = *reinterpret_cast<const float4*>(xp + i), n = *reinterpret_cast<const float4*>(np + i);
      *reinterpret_cast<float4*>(yp + i) = make_float4(f(v.x, n.x), f(v.y, n.y), f(v.z, n.z), f(v.w, n.w));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += 256 * (int64_t)gridDim.x) yp[i] = f(xp[i], np[i]);
  }
}

__global__ __launch_bounds__(256) void noise_dot_kernel(float* __restrict__ out, const float* __restrict__ gx,
                                                         const float* __restrict__ noise, int C, int64_t hw) {
  const int64_t plane = blockIdx.y;
  const int b = (int)(plane / C);
  const float* gp = gx + plane * hw;
  const float* np = noise + (int64_t)b * hw;
  float s = 0.f;
  if ((hw & 3) == 0) {
    for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < hw; i += 4 * 256 * (int64_t)gridDim.x) {
      const float4 v = *reinterpret_cast<const float4*>(gp + i), n = *reinterpret_cast<const float4*>(np + i);
      s = fmaf(v.x, n.x, fmaf(v.y, n.y, fmaf(v.z, n.z, fmaf(v.w, n.w, s))));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += 256 * (int64_t)gridDim.x) s = fmaf(gp[i], np[i], s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// ---- backward of the tail of a SMART layer -- FusedLeakyReLU(bias1) -> NoiseInjection -> FusedLeakyReLU(bias2), which the forward
// runs as the fusion conv's epilogue (reference models/RestoreNet.py:220-244) -- as ONE stream over (g, y): both slope masks, the
// gradient that enters the conv, and the three parameter gradients.  Only the final y is kept: the first activation's output is
// recovered from it,  pre2 = y / m2,  y1 = pre2 - nw * noise - bias2  (m = gain or slope * gain by sign).
__global__ __launch_bounds__(256) void smart_tail_bwd_kernel(float* __restrict__ g1, float* __restrict__ db1, float* __restrict__ db2,
                                                             float* __restrict__ dnw, const float* __restrict__ g,
                                                             const float* __restrict__ y, const float* __restrict__ noise,
                                                             const float* __restrict__ nw, const float* __restrict__ bias2, int C,
                                                             int64_t hw, float slope, float gain) {
  const int64_t plane = blockIdx.y;
  const int b = (int)(plane / C), c = (int)(plane - (int64_t)b * C);
  const float w = nw[0], b2 = bias2[c];
  const float mp = gain, mn = slope * gain, ip = 1.f / gain, in = 1.f / (slope * gain);
  const float* gp = g + plane * hw;
  const float* yp = y + plane * hw;
  const float* np = noise + (int64_t)b * hw;
  float* op = g1 + plane * hw;
  float s1 = 0.f, s2 = 0.f, sw = 0.f;
  auto f = [&](float gv, float yv, float nz) {
    const bool pos = yv > 0.f;
    const float g2 = gv * (pos ? mp : mn);
    const float y1 = yv * (pos ? ip : in) - w * nz - b2;
    const float o = g2 * (y1 > 0.f ? mp : mn);
    s2 += g2;
    sw = fmaf(g2, nz, sw);
    s1 += o;
    return o;
  };
  if ((hw & 3) == 0) {
    for (int64_t i = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); i < hw; i += 4 * 256 * (int64_t)gridDim.x) {
      const float4 gv = *reinterpret_cast<const float4*>(gp + i), yv = *reinterpret_cast<const float4*>(yp + i),
                   nz = *reinterpret_cast<const float4*>(np + i);
      *reinterpret_cast<float4*>(op + i) = make_float4(f(gv.x, yv.x, nz.x), f(gv.y, yv.y, nz.y), f(gv.z, yv.z, nz.z), f(gv.w, yv.w, nz.w));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += 256 * (int64_t)gridDim.x) op[i] = f(gp[i], yp[i], np[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
    sw += __shfl_xor(sw, o, 64);
  }
  __shared__ float red[3][4];
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
    red[2][threadIdx.x >> 6] = sw;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(db1 + c, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    unsafeAtomicAdd(db2 + c, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    unsafeAtomicAdd(dnw, (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]));
  }
}

inline int plane_blocks(int64_t planes, int64_t hw) {   // blocks per plane: ~8 workgroups per CU in all, >= 4096 elements each
  int64_t nb = (8 * vsp::kNumCU + planes - 1) / planes;
  if (nb > hw / 4096) nb = hw / 4096;
  return (int)(nb < 1 ? 1 : nb);
}

}  // namespace

extern "C" int vsp_noise_bias_act_f32(float* y, const float* x, const float* noise, const float* noise_w, const float* bias, int B, int C,
                                      int64_t hw, float slope, float gain, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 0 && hw >= 0, "noise_bias_act: negative size");
  if ((int64_t)B * C * hw == 0) return VSP_OK;
  VSP_REQUIRE(y && x && noise && noise_w, "noise_bias_act: null pointer");
  VSP_REQUIRE((int64_t)B * C <= 65535, "noise_bias_act: too many planes for one grid");
  noise_bias_act_kernel<<<dim3((unsigned)plane_blocks((int64_t)B * C, hw), (unsigned)(B * C)), 256, 0, vsp::as_stream(stream)>>>(
      y, x, noise, noise_w, bias, C, hw, slope, gain);
  return vsp::check_launch("noise_bias_act");
}

extern "C" int vsp_noise_dot_f32(float* out, const float* gx, const float* noise, int B, int C, int64_t hw, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 0 && hw >= 0, "noise_dot: negative size");
  VSP_REQUIRE(out != nullptr, "noise_dot: null output");
  hipStream_t st = vsp::as_stream(stream);
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "noise_dot: memset failed");
  if ((int64_t)B * C * hw == 0) return VSP_OK;
  VSP_REQUIRE(gx && noise, "noise_dot: null pointer");
  VSP_REQUIRE((int64_t)B * C <= 65535, "noise_dot: too many planes for one grid");
  noise_dot_kernel<<<dim3((unsigned)plane_blocks((int64_t)B * C, hw), (unsigned)(B * C)), 256, 0, st>>>(out, gx, noise, C, hw);
  return vsp::check_launch("noise_dot");
}

extern "C" int vsp_smart_tail_bwd_f32(float* g1, float* db1, float* db2, float* dnw, const float* g, const float* y, const float* noise,
                                      const float* noise_w, const float* bias2, int B, int C, int64_t hw, float slope, float gain,
                                      vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 1 && hw >= 0, "smart_tail_bwd: bad dims");
  VSP_REQUIRE(db1 && db2 && dnw, "smart_tail_bwd: null output");
  VSP_REQUIRE(slope > 0.f && gain > 0.f, "smart_tail_bwd: slope and gain must be positive (the masks are recovered from signs)");
  hipStream_t st = vsp::as_stream(stream);
  if (hipMemsetAsync(db1, 0, sizeof(float) * C, st) != hipSuccess || hipMemsetAsync(db2, 0, sizeof(float) * C, st) != hipSuccess ||
      hipMemsetAsync(dnw, 0, sizeof(float), st) != hipSuccess)
    return vsp::fail(VSP_ELAUNCH, "smart_tail_bwd: memset failed");
  if ((int64_t)B * C * hw == 0) return VSP_OK;
  VSP_REQUIRE(g1 && g && y && noise && noise_w && bias2, "smart_tail_bwd: null pointer");
  VSP_REQUIRE((int64_t)B * C <= 65535, "smart_tail_bwd: too many planes for one grid");
  smart_tail_bwd_kernel<<<dim3((unsigned)plane_blocks((int64_t)B * C, hw), (unsigned)(B * C)), 256, 0, st>>>(
      g1, db1, db2, dnw, g, y, noise, noise_w, bias2, C, hw, slope, gain);
  return vsp::check_launch("smart_tail_bwd");
}

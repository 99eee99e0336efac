// Operators of the training losses of restoration_train.py (BASELINE configs[4]: "RestoreNet fwd/bwd + id_loss + LPIPS") for gfx950:
//   * max-pooling forward / backward (VGG16 2x2 stride 2, ResNet 3x3 stride 2 pad 1) -- the backward recomputes the window's
//     arg-max (first maximum in row-major scan order, ATen's rule) instead of storing an index plane;
//   * the LPIPS layer distance  d[b] = mean_p sum_c w_c (f0/(|f0|+eps) - f1/(|f1|+eps))^2  (my_lpips/networks_basic.py:73-83 with
//     my_lpips/__init__.py:44-46) as ONE stream over the two feature maps instead of ~10 elementwise passes, and its gradient
//     w.r.t. f1 (the reference calls model.forward(target, pred): in1 is the generated image, my_lpips/__init__.py:42);
//   * the adjoint of the bilinear resize (F.interpolate(size=112) in front of the ArcFace network, Loss/id_loss.py:27,37,41).
// All HBM streams: thread per pixel, coalesced along the contiguous spatial dimension, channels strided by the plane size.
#include "vsp_common.h"

namespace {

inline int stream_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > vsp::kMaxStreamBlocks) b = vsp::kMaxStreamBlocks;
  return (int)(b < 1 ? 1 : b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// arg-max of the window of output (oy, ox): first maximum in scan order (NaN wins, as in ATen's max_pool2d_with_indices)
__device__ __forceinline__ int window_argmax(const float* xp, int H, int W, int oy, int ox, int k, int s, int p, float* vmax) {
  int y0 = oy * s - p, x0 = ox * s - p;
  const int y1 = min(y0 + k, H), x1 = min(x0 + k, W);
  y0 = max(y0, 0);
  x0 = max(x0, 0);
  int best = y0 * W + x0;
  float m = xp[best];
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const float v = xp[y * W + x];
      if (v > m || (v != v && m == m)) {
        m = v;
        best = y * W + x;
      }
    }
  *vmax = m;
  return best;
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t planes, int H,
                                                           int W, int OH, int OW, int k, int s, int p) {
  const int64_t total = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    float m;
    window_argmax(x + pl * H * W, H, W, oy, ox, k, s, p, &m);
    out[i] = m;
  }
}

// gather form: thread per INPUT pixel, over the (at most ceil(k/s)^2) windows that contain it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                           const float* __restrict__ x, int64_t planes, int H, int W, int OH, int OW,
                                                           int k, int s, int p) {
  const int64_t total = planes * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    const int64_t t = i / W;
    const int iy = (int)(t % H);
    const int64_t pl = t / H;
    const float* xp = x + pl * H * W;
    const float* gp = dy + pl * OH * OW;
    // windows oy with oy*s - p <= iy < oy*s - p + k  (a negative numerator truncates towards zero: lower bound 0 either way)
    const int oy_lo = max((iy + p - k + s) / s, 0), oy_hi = min((iy + p) / s, OH - 1);
    const int ox_lo = max((ix + p - k + s) / s, 0), ox_hi = min((ix + p) / s, OW - 1);
    float g = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        float m;
        if (window_argmax(xp, H, W, oy, ox, k, s, p, &m) == iy * W + ix) g += gp[oy * OW + ox];
      }
    dx[i] = g;
  }
}

// ---- LPIPS layer distance.  grid (blocks over pixels, B); out[b] += (1/HW) * sum over the block's pixels
__global__ __launch_bounds__(256) void lpips_layer_fwd_kernel(float* __restrict__ out, const float* __restrict__ f0,
                                                               const float* __restrict__ f1, const float* __restrict__ w, int C,
                                                               int HW, float eps) {
  const int b = blockIdx.y;
  const float* a = f0 + (int64_t)b * C * HW;
  const float* c = f1 + (int64_t)b * C * HW;
  float acc = 0.f;
  for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < HW; px += gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    for (int ch = 0; ch < C; ++ch) {
      const float u = a[(int64_t)ch * HW + px], v = c[(int64_t)ch * HW + px];
      s0 = fmaf(u, u, s0);
      s1 = fmaf(v, v, s1);
    }
    const float r0 = 1.f / (sqrtf(s0) + eps), r1 = 1.f / (sqrtf(s1) + eps);
    float d = 0.f;
    for (int ch = 0; ch < C; ++ch) {
      const float e = a[(int64_t)ch * HW + px] * r0 - c[(int64_t)ch * HW + px] * r1;
      d = fmaf(w[ch] * e, e, d);
    }
    acc += d;
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out + b, (red[0] + red[1] + red[2] + red[3]) / (float)HW);
}

// d out[b] / d f1:  with n = |f1|, r = 1/(n+eps), e_c = f0_c r0 - f1_c r,  g_c = -2 w_c e_c:
//   d/df1_k = g_k r - f1_k (r^2 / n) sum_c g_c f1_c        (second term absent where n = 0)
__global__ __launch_bounds__(256) void lpips_layer_bwd_kernel(float* __restrict__ df1, const float* __restrict__ f0,
                                                               const float* __restrict__ f1, const float* __restrict__ w,
                                                               const float* __restrict__ gout, int C, int HW, float eps) {
  const int b = blockIdx.y;
  const float* a = f0 + (int64_t)b * C * HW;
  const float* c = f1 + (int64_t)b * C * HW;
  float* o = df1 + (int64_t)b * C * HW;
  const float go = gout[b] / (float)HW;
  for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < HW; px += gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    for (int ch = 0; ch < C; ++ch) {
      const float u = a[(int64_t)ch * HW + px], v = c[(int64_t)ch * HW + px];
      s0 = fmaf(u, u, s0);
      s1 = fmaf(v, v, s1);
    }
    const float n1 = sqrtf(s1);
    const float r0 = 1.f / (sqrtf(s0) + eps), r1 = 1.f / (n1 + eps);
    float dot = 0.f;
    for (int ch = 0; ch < C; ++ch) {
      const float v = c[(int64_t)ch * HW + px];
      const float e = a[(int64_t)ch * HW + px] * r0 - v * r1;
      dot = fmaf(-2.f * w[ch] * e, v, dot);
    }
    const float k2 = n1 > 0.f ? dot * r1 * r1 / n1 : 0.f;
    for (int ch = 0; ch < C; ++ch) {
      const float v = c[(int64_t)ch * HW + px];
      const float e = a[(int64_t)ch * HW + px] * r0 - v * r1;
      o[(int64_t)ch * HW + px] = go * (-2.f * w[ch] * e * r1 - v * k2);
    }
  }
}

// Register forms of the two kernels above for C = 64 * LPP, LPP = 1, 2, 4 (the first three LPIPS levels: 94 % of the bytes): LPP lanes share
// a pixel, each keeps its 64 channels of BOTH maps in registers, so every element is read ONCE (the generic kernels walk the channels two
// / three times: 2.4 GB through L2 for the 805 MB of the 64-channel 512^2 level, 533 us).  Lane = pixel + (64 / LPP) * part: the lanes of a
// part read consecutive pixels of one channel plane; channel sums meet through xor-shuffles across the parts.
template <int LPP>
__device__ __forceinline__ float part_sum(float v) {
#pragma unroll
  for (int o = 64 / LPP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPP, bool BWD>
__global__ __launch_bounds__(256) void lpips_layer_reg_kernel(float* __restrict__ outp, const float* __restrict__ f0, const float* __restrict__ f1,
                                                               const float* __restrict__ w, const float* __restrict__ gout, int HW, float eps) {
  constexpr int PXW = 64 / LPP;                       // pixels per wave
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane / PXW, pl = lane % PXW;
  const int C = 64 * LPP;
  const float* a = f0 + ((int64_t)b * C + 64 * part) * HW;
  const float* c = f1 + ((int64_t)b * C + 64 * part) * HW;
  const float* wp = w + 64 * part;
  float acc = 0.f;
  const float go = BWD ? gout[b] / (float)HW : 0.f;
  for (int px0 = (blockIdx.x * 4 + wave) * PXW; px0 < HW; px0 += gridDim.x * 4 * PXW) {
    const int px = px0 + pl;
    const bool ok = px < HW;
    const int pc = ok ? px : HW - 1;
    float u[64], v[64];
#pragma unroll
    for (int ch = 0; ch < 64; ++ch) {
      u[ch] = a[(int64_t)ch * HW + pc];
      v[ch] = c[(int64_t)ch * HW + pc];
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int ch = 0; ch < 64; ++ch) {
      s0 = fmaf(u[ch], u[ch], s0);
      s1 = fmaf(v[ch], v[ch], s1);
    }
    s0 = part_sum<LPP>(s0);
    s1 = part_sum<LPP>(s1);
    const float n1 = sqrtf(s1);
    const float r0 = 1.f / (sqrtf(s0) + eps), r1 = 1.f / (n1 + eps);
    if constexpr (!BWD) {
      float d = 0.f;
#pragma unroll
      for (int ch = 0; ch < 64; ++ch) {
        const float e = u[ch] * r0 - v[ch] * r1;
        d = fmaf(wp[ch] * e, e, d);
      }
      acc += ok ? d : 0.f;
    } else {
      float dot = 0.f;
#pragma unroll
      for (int ch = 0; ch < 64; ++ch) {
        const float e = u[ch] * r0 - v[ch] * r1;
        dot = fmaf(-2.f * wp[ch] * e, v[ch], dot);
      }
      dot = part_sum<LPP>(dot);
      const float k2 = n1 > 0.f ? dot * r1 * r1 / n1 : 0.f;
      float* o = outp + ((int64_t)b * C + 64 * part) * HW;
      if (ok) {
#pragma unroll
        for (int ch = 0; ch < 64; ++ch) {
          const float e = u[ch] * r0 - v[ch] * r1;
          o[(int64_t)ch * HW + px] = go * (-2.f * wp[ch] * e * r1 - v[ch] * k2);
        }
      }
    }
  }
  if constexpr (!BWD) {
    __shared__ float red[4];
    acc = wave_sum(acc);   // (all parts: every lane holds its share of the channels)
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(outp + b, (red[0] + red[1] + red[2] + red[3]) / (float)HW);
  }
}

// adjoint of resize_bilinear_kernel (rowops.hip): scatter of each output gradient onto its four source pixels
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy, int64_t planes,
                                                                   int IH, int IW, int OH, int OW, float sy, float sx) {
  const int64_t total = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > IH - 1) y0 = IH - 1;
    if (x0 > IW - 1) x0 = IW - 1;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    float* xp = dx + pl * IH * IW;
    const float g = dy[i];
    unsafeAtomicAdd(xp + y0 * IW + x0, hy * hx * g);
    unsafeAtomicAdd(xp + y0 * IW + x1, hy * lx * g);
    unsafeAtomicAdd(xp + y1 * IW + x0, ly * hx * g);
    unsafeAtomicAdd(xp + y1 * IW + x1, ly * lx * g);
  }
}

int pool_dims_ok(int H, int W, int OH, int OW, int k, int s, int p) {
  return k >= 1 && s >= 1 && p >= 0 && 2 * p <= k && H >= 1 && W >= 1 && OH == (H + 2 * p - k) / s + 1 && OW == (W + 2 * p - k) / s + 1;
}

}  // namespace

extern "C" {

// 2 x 2 windows at stride 2 without padding on planes whose width is a multiple of 4 (every pooling of the LPIPS VGG): a thread owns two
// neighbouring windows = four columns of two input rows -- 16-byte loads / stores, no index arithmetic per element, the same
// first-maximum-in-scan-order rule as window_argmax (the generic gather form ran the 64-channel 512^2 level at 1.9 TB/s).
__device__ __forceinline__ int argmax4(float a, float b, float c, float d) {   // scan order (y0,x0) (y0,x1) (y1,x0) (y1,x1); NaN wins
  int best = 0;
  float m = a;
  if (b > m || (b != b && m == m)) { m = b; best = 1; }
  if (c > m || (c != c && m == m)) { m = c; best = 2; }
  if (d > m || (d != d && m == m)) { m = d; best = 3; }
  return best;
}
__device__ __forceinline__ float max4(float a, float b, float c, float d) {
  float m = a;
  if (b > m || (b != b && m == m)) m = b;
  if (c > m || (c != c && m == m)) m = c;
  if (d > m || (d != d && m == m)) m = d;
  return m;
}
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t rows, int W) {
  // rows = planes * OH output rows; a thread = two outputs of one output row
  const int W4 = W >> 2;
  const int64_t total = rows * W4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / W4;
    const int c = (int)(i - r * W4);
    const float4 a = *reinterpret_cast<const float4*>(x + (2 * r) * W + 4 * c);
    const float4 b = *reinterpret_cast<const float4*>(x + (2 * r + 1) * W + 4 * c);
    *reinterpret_cast<float2*>(out + r * (W >> 1) + 2 * c) = make_float2(max4(a.x, a.y, b.x, b.y), max4(a.z, a.w, b.z, b.w));
  }
}
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                              const float* __restrict__ x, int64_t rows, int W) {
  const int W4 = W >> 2;
  const int64_t total = rows * W4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / W4;
    const int c = (int)(i - r * W4);
    const float4 a = *reinterpret_cast<const float4*>(x + (2 * r) * W + 4 * c);
    const float4 b = *reinterpret_cast<const float4*>(x + (2 * r + 1) * W + 4 * c);
    const float2 g = *reinterpret_cast<const float2*>(dy + r * (W >> 1) + 2 * c);
    const int k0 = argmax4(a.x, a.y, b.x, b.y), k1 = argmax4(a.z, a.w, b.z, b.w);
    *reinterpret_cast<float4*>(dx + (2 * r) * W + 4 * c) =
        make_float4(k0 == 0 ? g.x : 0.f, k0 == 1 ? g.x : 0.f, k1 == 0 ? g.y : 0.f, k1 == 1 ? g.y : 0.f);
    *reinterpret_cast<float4*>(dx + (2 * r + 1) * W + 4 * c) =
        make_float4(k0 == 2 ? g.x : 0.f, k0 == 3 ? g.x : 0.f, k1 == 2 ? g.y : 0.f, k1 == 3 ? g.y : 0.f);
  }
}
static bool pool2x2_form(const void* a, const void* b, const void* c, int H, int W, int OH, int OW, int k, int s, int p) {
  return k == 2 && s == 2 && p == 0 && (W & 3) == 0 && (H & 1) == 0 && OH == H / 2 && OW == W / 2 && vsp::aligned16(a) && vsp::aligned16(b) &&
         (c == nullptr || (reinterpret_cast<uintptr_t>(c) & 7u) == 0);
}

int vsp_maxpool2d_f32(float* out, const float* x, int64_t planes, int H, int W, int OH, int OW, int k, int s, int p,
                      vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && pool_dims_ok(H, W, OH, OW, k, s, p), "maxpool2d: bad geometry (%dx%d -> %dx%d, k %d s %d p %d; floor mode)",
              H, W, OH, OW, k, s, p);
  const int64_t n = planes * OH * OW;
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "maxpool2d: null pointer");
  if (pool2x2_form(x, x, out, H, W, OH, OW, k, s, p)) {
    maxpool2x2_fwd_kernel<<<stream_blocks(planes * OH * (W >> 2)), 256, 0, vsp::as_stream(stream)>>>(out, x, planes * OH, W);
    return vsp::check_launch("maxpool2d");
  }
  maxpool_fwd_kernel<<<stream_blocks(n), 256, 0, vsp::as_stream(stream)>>>(out, x, planes, H, W, OH, OW, k, s, p);
  return vsp::check_launch("maxpool2d");
}

int vsp_maxpool2d_bwd_f32(float* dx, const float* dy, const float* x, int64_t planes, int H, int W, int OH, int OW, int k, int s,
                          int p, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && pool_dims_ok(H, W, OH, OW, k, s, p), "maxpool2d_bwd: bad geometry");
  const int64_t n = planes * H * W;
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(dx && dy && x, "maxpool2d_bwd: null pointer");
  if (pool2x2_form(x, dx, dy, H, W, OH, OW, k, s, p)) {
    maxpool2x2_bwd_kernel<<<stream_blocks(planes * OH * (W >> 2)), 256, 0, vsp::as_stream(stream)>>>(dx, dy, x, planes * OH, W);
    return vsp::check_launch("maxpool2d_bwd");
  }
  maxpool_bwd_kernel<<<stream_blocks(n), 256, 0, vsp::as_stream(stream)>>>(dx, dy, x, planes, H, W, OH, OW, k, s, p);
  return vsp::check_launch("maxpool2d_bwd");
}

int vsp_lpips_layer_f32(float* out, const float* f0, const float* f1, const float* w, int B, int C, int HW, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 1 && HW >= 1, "lpips_layer: bad dims");
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(out && f0 && f1 && w, "lpips_layer: null pointer");
  hipStream_t st = vsp::as_stream(stream);
  if (hipMemsetAsync(out, 0, sizeof(float) * B, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "lpips_layer: memset failed");
  if (C == 64 || C == 128 || C == 256) {   // register form: every element read once
    const int lpp = C / 64, pxb = 4 * (64 / lpp);
    int bx = (HW + pxb - 1) / pxb;
    if (bx > 2048) bx = 2048;
    if (lpp == 1) lpips_layer_reg_kernel<1, false><<<dim3(bx, B), 256, 0, st>>>(out, f0, f1, w, nullptr, HW, 1e-10f);
    else if (lpp == 2) lpips_layer_reg_kernel<2, false><<<dim3(bx, B), 256, 0, st>>>(out, f0, f1, w, nullptr, HW, 1e-10f);
    else lpips_layer_reg_kernel<4, false><<<dim3(bx, B), 256, 0, st>>>(out, f0, f1, w, nullptr, HW, 1e-10f);
    return vsp::check_launch("lpips_layer");
  }
  int bx = (HW + 255) / 256;
  if (bx > 1024) bx = 1024;
  lpips_layer_fwd_kernel<<<dim3(bx, B), 256, 0, st>>>(out, f0, f1, w, C, HW, 1e-10f);
  return vsp::check_launch("lpips_layer");
}

int vsp_lpips_layer_bwd_f32(float* df1, const float* f0, const float* f1, const float* w, const float* gout, int B, int C, int HW,
                            vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 1 && HW >= 1, "lpips_layer_bwd: bad dims");
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(df1 && f0 && f1 && w && gout, "lpips_layer_bwd: null pointer");
  if (C == 64 || C == 128 || C == 256) {
    const int lpp = C / 64, pxb = 4 * (64 / lpp);
    int bx = (HW + pxb - 1) / pxb;
    if (bx > 4096) bx = 4096;
    hipStream_t st = vsp::as_stream(stream);
    if (lpp == 1) lpips_layer_reg_kernel<1, true><<<dim3(bx, B), 256, 0, st>>>(df1, f0, f1, w, gout, HW, 1e-10f);
    else if (lpp == 2) lpips_layer_reg_kernel<2, true><<<dim3(bx, B), 256, 0, st>>>(df1, f0, f1, w, gout, HW, 1e-10f);
    else lpips_layer_reg_kernel<4, true><<<dim3(bx, B), 256, 0, st>>>(df1, f0, f1, w, gout, HW, 1e-10f);
    return vsp::check_launch("lpips_layer_bwd");
  }
  int bx = (HW + 255) / 256;
  if (bx > 4096) bx = 4096;
  lpips_layer_bwd_kernel<<<dim3(bx, B), 256, 0, vsp::as_stream(stream)>>>(df1, f0, f1, w, gout, C, HW, 1e-10f);
  return vsp::check_launch("lpips_layer_bwd");
}

int vsp_resize_bilinear_bwd_f32(float* dx, const float* dy, int64_t planes, int IH, int IW, int OH, int OW, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "resize_bilinear_bwd: bad dims");
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(dx && dy, "resize_bilinear_bwd: null pointer");
  hipStream_t st = vsp::as_stream(stream);
  if (hipMemsetAsync(dx, 0, sizeof(float) * planes * IH * IW, st) != hipSuccess)
    return vsp::fail(VSP_ELAUNCH, "resize_bilinear_bwd: memset failed");
  const int64_t n = planes * OH * OW;
  resize_bilinear_bwd_kernel<<<stream_blocks(n), 256, 0, st>>>(dx, dy, planes, IH, IW, OH, OW, (float)IH / (float)OH,
                                                              (float)IW / (float)OW);
  return vsp::check_launch("resize_bilinear_bwd");
}

}  // extern "C"

// The two geometric / colour primitives of the ADA augmentation pipeline (reference non_leaking.py:857-934) for gfx950.
//
// affine_sample: F.affine_grid(theta, (B, C, OH, OW), align_corners=False) + F.grid_sample(x, grid, "bilinear", "zeros",
//   align_corners=False) in one pass (reference :891-892, :808-847): the sampling grid is affine, so it is evaluated per output
//   pixel from the six numbers of theta instead of being materialised ((B, OH, OW, 2) floats).
//     xn = (2 ox + 1) / OW - 1, yn = (2 oy + 1) / OH - 1;  gx = t00 xn + t01 yn + t02,  gy = t10 xn + t11 yn + t12
//     ix = ((gx + 1) IW - 1) / 2,  iy = ((gy + 1) IH - 1) / 2;  bilinear over the four neighbours, zero outside
//   The adjoint w.r.t. x (training: the generator's gradient flows back through the augmented fake image) scatters with atomics.
// color_affine: y[b, c, p] = sum_k M[b, c, k] x[b, k, p] + t[b, c] for 3 channels (reference apply_color, :910-918); its adjoint
//   is the same call with M transposed and no offset.
// Both are HBM streams (4 B read + 4 B written per element; the sampler gathers within a few rows).
#include "vsp_common.h"

namespace {

struct Taps {
  int x0, y0;
  float wx1, wy1;  // weights of the +1 neighbours
};

__device__ __forceinline__ Taps taps_of(const float* th, int ox, int oy, int OW, int OH, int IW, int IH) {
  const float xn = (2.f * (float)ox + 1.f) / (float)OW - 1.f, yn = (2.f * (float)oy + 1.f) / (float)OH - 1.f;
  const float gx = th[0] * xn + th[1] * yn + th[2], gy = th[3] * xn + th[4] * yn + th[5];
  const float ix = ((gx + 1.f) * (float)IW - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)IH - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  Taps t;
  // (far-away coordinates: clamp before the int conversion, the taps are outside either way)
  t.x0 = (int)fminf(fmaxf(fx, -2.f), (float)IW + 1.f);
  t.y0 = (int)fminf(fmaxf(fy, -2.f), (float)IH + 1.f);
  t.wx1 = ix - fx;
  t.wy1 = iy - fy;
  return t;
}

__global__ __launch_bounds__(256) void affine_sample_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                             const float* __restrict__ theta, int C, int IH, int IW, int OH,
                                                             int OW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    int64_t t = i / OW;
    const int oy = (int)(t % OH);
    t /= OH;
    const int c = (int)(t % C), b = (int)(t / C);
    const Taps tp = taps_of(theta + b * 6, ox, oy, OW, OH, IW, IH);
    const float* xp = x + ((int64_t)b * C + c) * IH * IW;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < IH && xx >= 0 && xx < IW) ? xp[yy * IW + xx] : 0.f; };
    const float w00 = (1.f - tp.wx1) * (1.f - tp.wy1), w01 = tp.wx1 * (1.f - tp.wy1), w10 = (1.f - tp.wx1) * tp.wy1, w11 = tp.wx1 * tp.wy1;
    out[i] = at(tp.y0, tp.x0) * w00 + at(tp.y0, tp.x0 + 1) * w01 + at(tp.y0 + 1, tp.x0) * w10 + at(tp.y0 + 1, tp.x0 + 1) * w11;
  }
}

__global__ __launch_bounds__(256) void affine_sample_bwd_kernel(float* __restrict__ gx, const float* __restrict__ gout,
                                                                 const float* __restrict__ theta, int C, int IH, int IW, int OH,
                                                                 int OW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    int64_t t = i / OW;
    const int oy = (int)(t % OH);
    t /= OH;
    const int c = (int)(t % C), b = (int)(t / C);
    const Taps tp = taps_of(theta + b * 6, ox, oy, OW, OH, IW, IH);
    float* gp = gx + ((int64_t)b * C + c) * IH * IW;
    const float g = gout[i];
    auto add = [&](int yy, int xx, float w) {
      if (yy >= 0 && yy < IH && xx >= 0 && xx < IW) unsafeAtomicAdd(gp + yy * IW + xx, g * w);
    };
    add(tp.y0, tp.x0, (1.f - tp.wx1) * (1.f - tp.wy1));
    add(tp.y0, tp.x0 + 1, tp.wx1 * (1.f - tp.wy1));
    add(tp.y0 + 1, tp.x0, (1.f - tp.wx1) * tp.wy1);
    add(tp.y0 + 1, tp.x0 + 1, tp.wx1 * tp.wy1);
  }
}

__global__ __launch_bounds__(256) void color_affine_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ M, const float* __restrict__ tr,
                                                            int64_t HW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / HW, p = i - b * HW;
    const float* xp = x + b * 3 * HW + p;
    const float r = xp[0], g = xp[HW], bl = xp[2 * HW];
    const float* m = M + b * 9;
    float* yp = y + b * 3 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) yp[c * HW] = m[c * 3] * r + m[c * 3 + 1] * g + m[c * 3 + 2] * bl + (tr ? tr[b * 3 + c] : 0.f);
  }
}

int blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > vsp::kMaxStreamBlocks ? vsp::kMaxStreamBlocks : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int vsp_affine_sample_f32(float* out, const float* x, const float* theta, int B, int C, int IH, int IW, int OH, int OW,
                                      vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 1 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "affine_sample: bad dims");
  const int64_t total = (int64_t)B * C * OH * OW;
  if (total == 0) return VSP_OK;
  VSP_REQUIRE(out && x && theta, "affine_sample: null pointer");
  affine_sample_kernel<<<blocks_for(total), 256, 0, vsp::as_stream(stream)>>>(out, x, theta, C, IH, IW, OH, OW, total);
  return vsp::check_launch("affine_sample");
}

extern "C" int vsp_affine_sample_bwd_f32(float* gx, const float* gout, const float* theta, int B, int C, int IH, int IW, int OH,
                                          int OW, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 1 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "affine_sample_bwd: bad dims");
  VSP_REQUIRE(gx != nullptr || B == 0, "affine_sample_bwd: null pointer");
  hipStream_t st = vsp::as_stream(stream);
  if (B > 0 && hipMemsetAsync(gx, 0, (size_t)B * C * IH * IW * sizeof(float), st) != hipSuccess)
    return vsp::fail(VSP_ELAUNCH, "affine_sample_bwd: memset failed");
  const int64_t total = (int64_t)B * C * OH * OW;
  if (total == 0) return VSP_OK;
  VSP_REQUIRE(gout && theta, "affine_sample_bwd: null pointer");
  affine_sample_bwd_kernel<<<blocks_for(total), 256, 0, st>>>(gx, gout, theta, C, IH, IW, OH, OW, total);
  return vsp::check_launch("affine_sample_bwd");
}

extern "C" int vsp_color_affine_f32(float* y, const float* x, const float* M, const float* t, int B, int64_t HW, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && HW >= 0, "color_affine: bad dims");
  const int64_t total = (int64_t)B * HW;
  if (total == 0) return VSP_OK;
  VSP_REQUIRE(y && x && M, "color_affine: null pointer");
  color_affine_kernel<<<blocks_for(total), 256, 0, vsp::as_stream(stream)>>>(y, x, M, t, HW, total);
  return vsp::check_launch("color_affine");
}

// Weight gradient of the path's convolutions on fp32 MFMA for gfx950 (the training step's conv backward;
// reference op/conv2d_gradfix.py:176-206 -> aten::cudnn_convolution_backward_weight, and on current torch simply autograd of
// F.conv2d / F.conv_transpose2d, op/conv2d_gradfix.py:78-92).
//
//   dW[g][co][ci][ky][kx] = sum_{b,oy,ox} dY[b, g Cout_g + co, oy, ox] * dys[b, .]  *  X[b, g Cin_g + ci, oy s + ky d - p, ox s + kx d - p] * xs[b, .]
//
// As a GEMM: M = output channels, N = (input channel, tap), K = every output pixel of the batch.  One 4-wave workgroup owns
// 64 output channels x 16 input channels x all KH*KW taps of one group and walks a strided share of the K dimension in
// chunks of one output-row segment (64 pixels): the dY slab [64 co][64 px] and the X slab [16 ci][KH rows][row segment with
// halo] are staged in LDS with coalesced row reads (per-sample scales folded in: the modulate-input / demodulate-output form
// of the style-modulated layers), wave w multiplies its 16 channels against the nine shifted views of the X slab --
// v_mfma_f32_16x16x4_f32, one A fragment serving all taps of a k-step -- and the partial sums of the workgroups that share
// an output block meet through fp32 atomics (dW is a few MB; the order of the additions is not fixed: ~1e-6 relative).
// LDS pitches: co / ci rows 2 (mod 32) words apart, so the 16 x 2 lanes of an access group hit 32 distinct banks.
#include "vsp_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_CO = 64, WG_CI = 16, WG_PX = 64, WG_NT = 256;
constexpr int DPITCH = WG_PX + 2;  // 66 = 2 (mod 32)

struct WgradK {
  const float* x;
  const float* dy;
  float* dw;
  const float* xs;   // [B, x_ch] or null
  const float* dys;  // [B, dy_ch] or null
  int B, Cin_g, H, W, G, Cout_g, OH, OW, KH, KW, stride, dil, pad;
  int x_ch, dy_ch, xw, xwp, xplane, segs, chunks;
};

template <int NTAP>
__global__ __launch_bounds__(WG_NT) void conv_wgrad_kernel(const WgradK p) {
  extern __shared__ float wg_smem[];
  float* Dl = wg_smem;                       // [64 co][DPITCH]
  float* Xl = wg_smem + WG_CO * DPITCH;      // [16 ci][xplane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int g = blockIdx.z;
  const int ci_tiles = (p.Cin_g + WG_CI - 1) / WG_CI;
  const int cot = blockIdx.y / ci_tiles, cit = blockIdx.y - cot * ci_tiles;
  const int co0 = cot * WG_CO, ci0 = cit * WG_CI;
  const int KW = p.KW;

  f32x4 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ch = blockIdx.x; ch < p.chunks; ch += gridDim.x) {
    const int seg = ch % p.segs, row = ch / p.segs;
    const int oy = row % p.OH, b = row / p.OH;
    const int ox0 = seg * WG_PX;
    // ---- dY slab: thread t covers pixel t % 64 of channels t / 64, + 4, + 8, ... (rows are contiguous in memory)
    {
      const int px = tid & 63, ox = ox0 + px;
      for (int c = tid >> 6; c < WG_CO; c += WG_NT / 64) {
        const int co = co0 + c;
        float v = 0.f;
        if (co < p.Cout_g && ox < p.OW) {
          const int ch_ = g * p.Cout_g + co;
          v = p.dy[(((int64_t)b * p.dy_ch + ch_) * p.OH + oy) * p.OW + ox];
          if (p.dys) v *= p.dys[(int64_t)b * p.dy_ch + ch_];
        }
        Dl[c * DPITCH + px] = v;
      }
    }
    // ---- X slab: rows oy s + ky d - p, columns ox0 s - p .. + xw - 1
    {
      const int ix0 = ox0 * p.stride - p.pad;
      const int per_ci = p.KH * p.xw;
      for (int i = tid; i < WG_CI * per_ci; i += WG_NT) {
        const int c = i / per_ci, rem = i - c * per_ci;
        const int ky = rem / p.xw, cx = rem - ky * p.xw;
        const int iy = oy * p.stride + ky * p.dil - p.pad, ix = ix0 + cx;
        const int ci = ci0 + c;
        float v = 0.f;
        if (ci < p.Cin_g && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
          const int ch_ = g * p.Cin_g + ci;
          v = p.x[(((int64_t)b * p.x_ch + ch_) * p.H + iy) * p.W + ix];
          if (p.xs) v *= p.xs[(int64_t)b * p.x_ch + ch_];
        }
        Xl[c * p.xplane + ky * p.xwp + cx] = v;
      }
    }
    __syncthreads();
    const float* ap = Dl + (wave * 16 + r) * DPITCH + kq;
    const float* bp = Xl + r * p.xplane + kq * p.stride;
#pragma unroll 4
    for (int k0 = 0; k0 < WG_PX; k0 += 4) {
      const float a = ap[k0];
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int ky = t / KW, kx = t - ky * KW;
        const float bv = bp[ky * p.xwp + k0 * p.stride + kx * p.dil];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D layout: lane (r, kq) holds rows 4 kq + j (output channel), column r (input channel)
  const int ci = ci0 + r;
  if (ci < p.Cin_g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + wave * 16 + 4 * kq + j;
      if (co >= p.Cout_g) continue;
      float* dst = p.dw + (((int64_t)g * p.Cout_g + co) * p.Cin_g + ci) * NTAP;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) unsafeAtomicAdd(dst + t, acc[t][j]);
    }
  }
}

}  // namespace

extern "C" int vsp_conv2d_wgrad_f32(const vsp_conv_wgrad_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d_wgrad: null params");
  const vsp_conv_wgrad_params& q = *pp;
  VSP_REQUIRE(q.B >= 0 && q.G >= 1 && q.Cin_g >= 1 && q.Cout_g >= 1 && q.H >= 1 && q.W >= 1 && q.OH >= 0 && q.OW >= 0,
              "conv2d_wgrad: bad dimensions");
  VSP_REQUIRE((q.KH == 3 && q.KW == 3) || (q.KH == 1 && q.KW == 1), "conv2d_wgrad: 3x3 and 1x1 kernels only (got %dx%d)", q.KH, q.KW);
  VSP_REQUIRE(q.stride == 1 || q.stride == 2, "conv2d_wgrad: stride must be 1 or 2");
  VSP_REQUIRE(q.dil >= 1 && q.dil <= 64 && q.pad >= 0, "conv2d_wgrad: bad dilation / padding");
  VSP_REQUIRE(q.dw != nullptr, "conv2d_wgrad: null output");
  const size_t dw_bytes = (size_t)q.G * q.Cout_g * q.Cin_g * q.KH * q.KW * sizeof(float);
  hipStream_t st = vsp::as_stream(stream);
  if (hipMemsetAsync(q.dw, 0, dw_bytes, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d_wgrad: memset failed");
  if (q.B == 0 || q.OH == 0 || q.OW == 0) return VSP_OK;
  VSP_REQUIRE(q.x && q.dy, "conv2d_wgrad: null input");
  VSP_REQUIRE((q.OH - 1) * q.stride + (q.KH - 1) * q.dil - q.pad < q.H + q.pad && (q.OW - 1) * q.stride + (q.KW - 1) * q.dil - q.pad < q.W + q.pad,
              "conv2d_wgrad: %dx%d outputs do not fit a %dx%d input with stride %d, dilation %d, padding %d", q.OH, q.OW, q.H, q.W,
              q.stride, q.dil, q.pad);
  WgradK k{};
  k.x = q.x; k.dy = q.dy; k.dw = q.dw; k.xs = q.x_scale; k.dys = q.dy_scale;
  k.B = q.B; k.Cin_g = q.Cin_g; k.H = q.H; k.W = q.W; k.G = q.G; k.Cout_g = q.Cout_g; k.OH = q.OH; k.OW = q.OW;
  k.KH = q.KH; k.KW = q.KW; k.stride = q.stride; k.dil = q.dil; k.pad = q.pad;
  k.x_ch = q.G * q.Cin_g; k.dy_ch = q.G * q.Cout_g;
  k.xw = (WG_PX - 1) * q.stride + (q.KW - 1) * q.dil + 1;
  k.xwp = k.xw;
  int plane = q.KH * k.xwp;
  while (plane % 32 != 2) ++plane;  // ci rows 2 (mod 32) words apart
  k.xplane = plane;
  k.segs = (q.OW + WG_PX - 1) / WG_PX;
  const int64_t chunks = (int64_t)q.B * q.OH * k.segs;
  VSP_REQUIRE(chunks < ((int64_t)1 << 31), "conv2d_wgrad: too many pixels");
  k.chunks = (int)chunks;
  const int tiles = ((q.Cout_g + WG_CO - 1) / WG_CO) * ((q.Cin_g + WG_CI - 1) / WG_CI);
  VSP_REQUIRE((int64_t)tiles <= 65535 && q.G <= 65535, "conv2d_wgrad: grid too large");
  // split the pixel dimension so that ~4 workgroups per CU are in flight, every workgroup keeping >= 8 chunks when it can
  int64_t split = (4 * vsp::kNumCU + (int64_t)tiles * q.G - 1) / ((int64_t)tiles * q.G);
  if (split > chunks / 8) split = chunks / 8;
  if (split < 1) split = 1;
  const size_t lds = ((size_t)WG_CO * DPITCH + (size_t)WG_CI * k.xplane) * sizeof(float);
  VSP_REQUIRE(lds <= 64 * 1024, "conv2d_wgrad: row segment with halo does not fit LDS (dilation %d)", q.dil);
  dim3 grid((unsigned)split, (unsigned)tiles, (unsigned)q.G);
  if (q.KH == 3)
    conv_wgrad_kernel<9><<<grid, WG_NT, lds, st>>>(k);
  else
    conv_wgrad_kernel<1><<<grid, WG_NT, lds, st>>>(k);
  return vsp::check_launch("conv2d_wgrad");
}

// out[plane] = sum_i a[plane, i] * b[plane, i]: the gradients of the per-sample input / output scales of a modulated convolution
// (d in_scale[b, ci] = <dXs, x>, d out_scale[b, co] = <dY, y> / out_scale).  One workgroup per plane, HBM stream.
namespace {
__global__ __launch_bounds__(256) void plane_dot_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                         const float* __restrict__ b, int64_t n) {
  const float* ap = a + (int64_t)blockIdx.x * n;
  const float* bp = b + (int64_t)blockIdx.x * n;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s = fmaf(ap[i], bp[i], s);
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
}  // namespace

extern "C" int vsp_plane_dot_f32(float* out, const float* a, const float* b, int64_t planes, int64_t n, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && n >= 0, "plane_dot: negative size");
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(out && (n == 0 || (a && b)), "plane_dot: null pointer");
  VSP_REQUIRE(planes < ((int64_t)1 << 31), "plane_dot: too many planes");
  plane_dot_kernel<<<(unsigned)planes, 256, 0, vsp::as_stream(stream)>>>(out, a, b, n);
  return vsp::check_launch("plane_dot");
}

// Weight gradient of the path's convolutions on fp32 MFMA for gfx950 (the training step's conv backward;
// reference op/conv2d_gradfix.py:176-206 -> aten::cudnn_convolution_backward_weight, and on current torch simply autograd of
// F.conv2d / F.conv_transpose2d, op/conv2d_gradfix.py:78-92).
//
//   dW[g][co][ci][ky][kx] = sum_{b,oy,ox} dY[b, g Cout_g + co, oy, ox] * dys[b, .]  *  X[b, xc(g) + ci, oy s + ky d_g - p_g, ox s + kx d_g - p_g] * xs[b, .]
//
// As a GEMM: M = output channels, N = (input channel, tap), K = every output pixel of the batch.  One 4-wave workgroup owns a
// (16 WCO) x (16 NB WCI) block of (co, ci) with all KH*KW taps of one group -- wave (wco, wci) its 16 co x 16 NB ci -- and walks a
// strided share of K in chunks of one output-row segment (64 pixels):
//   * tile shapes: 64 co x 32 ci (WCO 4, NB 2) for ordinary layers, 32 x 64 (WCO 2, WCI 2, NB 2) and 16 x 64 (WCI 4) for the
//     narrow dilated branches of a SMART layer (16 / 32 output channels per group: a 64-co tile would be 3/4 empty);
//   * the dY slab [co][64 px] and the X slab [ci][KH rows][row segment + halo] of chunk i+1 are fetched into REGISTERS (16-byte
//     loads; every thread keeps one (row, column quad) of the X slab and walks the channels, so the border masks are one row flag
//     and four column bits per chunk) before the MFMA loop of chunk i and written to LDS after it -- the global latency hides
//     under ~150 MFMAs per wave; the per-sample scales of a modulated layer (modulate-input / demodulate-output form) are folded
//     in at the LDS write;
//   * a k-step = 4 pixels: one A fragment (dY) serves all taps and ci blocks, B fragments are the tap-shifted views of the X slab;
//   * the partial sums of the workgroups that share a (co, ci) block meet through fp32 atomics (dW is a few MB; the order of
//     the additions is not fixed: ~1e-6 relative).
// Groups: true groups (x channels g Cin_g ...), or a SHARED input with per-group dilation / padding (the four SMART branches in
// one launch).  LDS pitches: channel rows 2 (mod 32) words apart: the 16 x 2 lanes of an access group hit 32 distinct banks.
#include "vsp_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));

constexpr int WG_PX = 64, WG_NT = 256;
constexpr int DPITCH = WG_PX + 2;  // 66 = 2 (mod 32)
constexpr size_t kMaxLds = 128 * 1024;

struct WgradK {
  const float* x;
  const float* dy;
  float* dw;
  const float* xs;   // [B, x_ch] or null
  const float* dys;  // [B, dy_ch] or null
  int B, Cin_g, H, W, G, Cout_g, OH, OW, KH, KW, stride;
  int dil[4], pad[4], per_group;
  int x_ch, x_coff, x_gs, dy_ch, dy_coff;
  int xwp, plane, segs, chunks;
};

template <int NTAP, int WCO, int NB, int XJ>
__global__ __launch_bounds__(WG_NT) void conv_wgrad_kernel(const WgradK p) {
  constexpr int WCI = 4 / WCO;
  constexpr int CO_T = 16 * WCO, CI_T = 16 * NB * WCI;
  constexpr int NITX = CI_T / 4;   // X items (one float4 each) per thread: channel = wave + 4 it
  constexpr int NITD = WCO;        // dY items per thread: row = tid / 16 + 16 it
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  float* Dl = wg_smem;                       // [CO_T][DPITCH]
  float* Xl = wg_smem + CO_T * DPITCH;       // [CI_T][plane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int g = blockIdx.z;
  const int ci_tiles = (p.Cin_g + CI_T - 1) / CI_T;
  const int cot = blockIdx.y / ci_tiles, cit = blockIdx.y - cot * ci_tiles;
  const int co0 = cot * CO_T, ci0 = cit * CI_T;
  const int KW = NTAP == 1 ? 1 : p.KW, KH = NTAP == 1 ? 1 : p.KH;
  const int d = p.dil[p.per_group ? g : 0], pad = p.pad[p.per_group ? g : 0], s = p.stride;
  const int XWP = p.xwp, plane = p.plane;
  const int xw = (WG_PX - 1) * s + (KW - 1) * d + 1;            // slab row of this group
  const int xw4 = (xw + 3) >> 2;
  const int xc0 = p.x_coff + g * p.x_gs + ci0, yc0 = p.dy_coff + g * p.Cout_g + co0;
  const int64_t xplane = (int64_t)p.H * p.W, yplane = (int64_t)p.OH * p.OW;

  // ---- staging roles.  X: item = lane + 64 j -> (slab row, column quad), wave + 4 it -> channel (XJ = 2: slab rows of more than
  //      64 / KH quads, i.e. stride 2).  dY: tid / 16 + 16 it -> channel, tid % 16 -> quad
  int x_row[XJ], x_q[XJ];
  bool x_on[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int item = lane + 64 * j;
    x_row[j] = item / xw4;
    x_q[j] = item - x_row[j] * xw4;
    x_on[j] = x_row[j] < KH;
  }
  const int d_row = tid >> 4, d_q = tid & 15;
  float4 xr[XJ][NITX], dr[NITD];
  float xsc[NITX], dsc[NITD];
  int x_mask[XJ], d_mask = 0;       // bit e: element e of the quad lies inside the image (and the row does)
#pragma unroll
  for (int j = 0; j < XJ; ++j) x_mask[j] = 0;
  auto fetch = [&](int ch) {
    const int seg = ch % p.segs, row = ch / p.segs;
    const int oy = row % p.OH, b = row / p.OH;
    const int ox0 = seg * WG_PX;
    {  // dY
      const int ox = ox0 + 4 * d_q;
      d_mask = (ox < p.OW ? 1 : 0) | (ox + 1 < p.OW ? 2 : 0) | (ox + 2 < p.OW ? 4 : 0) | (ox + 3 < p.OW ? 8 : 0);
#pragma unroll
      for (int it = 0; it < NITD; ++it) {
        const int c = d_row + 16 * it;
        const bool cok = co0 + c < p.Cout_g;
        const int ch_ = yc0 + (cok ? c : 0);
        const float* src = p.dy + ((int64_t)b * p.dy_ch + ch_) * yplane + (int64_t)oy * p.OW;
        if (d_mask == 15) {
          const f32x4u v = *reinterpret_cast<const f32x4u*>(src + ox);
          dr[it] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          dr[it].x = src[min(ox, p.OW - 1)]; dr[it].y = src[min(ox + 1, p.OW - 1)];
          dr[it].z = src[min(ox + 2, p.OW - 1)]; dr[it].w = src[min(ox + 3, p.OW - 1)];
        }
        dsc[it] = cok ? (p.dys ? p.dys[(int64_t)b * p.dy_ch + ch_] : 1.f) : 0.f;
      }
    }
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int c = wave + 4 * it;
      xsc[it] = ci0 + c < p.Cin_g ? (p.xs ? p.xs[(int64_t)b * p.x_ch + xc0 + c] : 1.f) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if (!x_on[j]) continue;
      const int iy = oy * s + x_row[j] * d - pad, ix = ox0 * s - pad + 4 * x_q[j];
      const bool rok = iy >= 0 && iy < p.H;
      x_mask[j] = rok ? ((ix >= 0 && ix < p.W ? 1 : 0) | (ix + 1 >= 0 && ix + 1 < p.W ? 2 : 0) | (ix + 2 >= 0 && ix + 2 < p.W ? 4 : 0) |
                         (ix + 3 >= 0 && ix + 3 < p.W ? 8 : 0)) : 0;
      const int iyc = min(max(iy, 0), p.H - 1);
      const float* rowp = p.x + ((int64_t)b * p.x_ch + xc0) * xplane + (int64_t)iyc * p.W;
      const int e0 = min(max(ix, 0), p.W - 1), e1 = min(max(ix + 1, 0), p.W - 1), e2 = min(max(ix + 2, 0), p.W - 1),
                e3 = min(max(ix + 3, 0), p.W - 1);
#pragma unroll
      for (int it = 0; it < NITX; ++it) {
        const int c = wave + 4 * it;
        const bool cok = ci0 + c < p.Cin_g;   // wave-uniform
        const float* src = rowp + (int64_t)(cok ? c : 0) * xplane;
        if (x_mask[j] == 15) {
          const f32x4u v = *reinterpret_cast<const f32x4u*>(src + ix);
          xr[j][it] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          xr[j][it] = make_float4(src[e0], src[e1], src[e2], src[e3]);
        }
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const float sc = dsc[it];
      float* dst = Dl + (d_row + 16 * it) * DPITCH + 4 * d_q;
      *reinterpret_cast<f32x2a*>(dst) = f32x2a{(d_mask & 1) ? dr[it].x * sc : 0.f, (d_mask & 2) ? dr[it].y * sc : 0.f};
      *reinterpret_cast<f32x2a*>(dst + 2) = f32x2a{(d_mask & 4) ? dr[it].z * sc : 0.f, (d_mask & 8) ? dr[it].w * sc : 0.f};
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if (!x_on[j]) continue;
      const int m = x_mask[j];
#pragma unroll
      for (int it = 0; it < NITX; ++it) {
        const float sc = xsc[it];
        float* dst = Xl + (wave + 4 * it) * plane + x_row[j] * XWP + 4 * x_q[j];
        *reinterpret_cast<f32x2a*>(dst) = f32x2a{(m & 1) ? xr[j][it].x * sc : 0.f, (m & 2) ? xr[j][it].y * sc : 0.f};
        *reinterpret_cast<f32x2a*>(dst + 2) = f32x2a{(m & 4) ? xr[j][it].z * sc : 0.f, (m & 8) ? xr[j][it].w * sc : 0.f};
      }
    }
  };

  f32x4 acc[NB][NTAP];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wco = wave % WCO, wci = wave / WCO;
  const float* ap = Dl + (wco * 16 + r) * DPITCH + kq;
  const float* bp = Xl + (wci * NB * 16 + r) * plane + kq * s;

  int ch = blockIdx.x;
  if (ch < p.chunks) fetch(ch);
  for (; ch < p.chunks; ch += gridDim.x) {
    __syncthreads();   // the MFMAs of the previous chunk have read the slabs
    commit();
    __syncthreads();
    if (ch + (int)gridDim.x < p.chunks) fetch(ch + gridDim.x);
#pragma unroll 2
    for (int k0 = 0; k0 < WG_PX; k0 += 4) {
      const float a = ap[k0];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int ky = t / 3, kx = t - 3 * ky;   // (NTAP = 9: 3x3; NTAP = 1: the single tap)
          const float bv = bp[nb * 16 * plane + ky * XWP + k0 * s + kx * d];
          acc[nb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[nb][t], 0, 0, 0);
        }
    }
  }
  // D layout: lane (r, kq) holds rows 4 kq + j (output channel), column r (input channel)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int ci = ci0 + (wci * NB + nb) * 16 + r;
    if (ci >= p.Cin_g) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + wco * 16 + 4 * kq + j;
      if (co >= p.Cout_g) continue;
      float* dst = p.dw + (((int64_t)g * p.Cout_g + co) * p.Cin_g + ci) * NTAP;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) unsafeAtomicAdd(dst + t, acc[nb][t][j]);
    }
  }
}

template <int NTAP, int WCO, int NB>
int launch_wgrad(const WgradK& k, int xj, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr_set = false;  // (slabs beyond the default 64 KB limit: stride-2 rows, the 64-channel X tiles)
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<NTAP, WCO, NB, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<NTAP, WCO, NB, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d_wgrad: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  if (xj == 1)
    conv_wgrad_kernel<NTAP, WCO, NB, 1><<<grid, WG_NT, lds, st>>>(k);
  else
    conv_wgrad_kernel<NTAP, WCO, NB, 2><<<grid, WG_NT, lds, st>>>(k);
  return VSP_OK;
}

}  // namespace

extern "C" int vsp_conv2d_wgrad_f32(const vsp_conv_wgrad_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d_wgrad: null params");
  const vsp_conv_wgrad_params& q = *pp;
  VSP_REQUIRE(q.B >= 0 && q.G >= 1 && q.Cin_g >= 1 && q.Cout_g >= 1 && q.H >= 1 && q.W >= 1 && q.OH >= 0 && q.OW >= 0,
              "conv2d_wgrad: bad dimensions");
  VSP_REQUIRE((q.KH == 3 && q.KW == 3) || (q.KH == 1 && q.KW == 1), "conv2d_wgrad: 3x3 and 1x1 kernels only (got %dx%d)", q.KH, q.KW);
  VSP_REQUIRE(q.stride == 1 || q.stride == 2, "conv2d_wgrad: stride must be 1 or 2");
  VSP_REQUIRE(q.dw != nullptr, "conv2d_wgrad: null output");
  VSP_REQUIRE(!q.per_group_geometry || q.G <= 4, "conv2d_wgrad: per-group dilation / padding for at most 4 groups");
  VSP_REQUIRE(q.x_ch >= 0 && q.dy_ch >= 0 && q.x_coff >= 0 && q.dy_coff >= 0, "conv2d_wgrad: negative channel count / offset");
  WgradK k{};
  int dmax = 1;
  for (int g = 0; g < 4; ++g) {
    k.dil[g] = q.per_group_geometry ? q.dil_g[g] : q.dil;
    k.pad[g] = q.per_group_geometry ? q.pad_g[g] : q.pad;
    if (g < q.G || !q.per_group_geometry) {
      VSP_REQUIRE(k.dil[g] >= 1 && k.dil[g] <= 64 && k.pad[g] >= 0, "conv2d_wgrad: bad dilation / padding");
      VSP_REQUIRE((q.OH - 1) * q.stride + (q.KH - 1) * k.dil[g] - k.pad[g] < q.H + k.pad[g] &&
                      (q.OW - 1) * q.stride + (q.KW - 1) * k.dil[g] - k.pad[g] < q.W + k.pad[g],
                  "conv2d_wgrad: %dx%d outputs do not fit a %dx%d input with stride %d, dilation %d, padding %d", q.OH, q.OW, q.H, q.W,
                  q.stride, k.dil[g], k.pad[g]);
      dmax = k.dil[g] > dmax ? k.dil[g] : dmax;
    }
  }
  k.per_group = q.per_group_geometry ? 1 : 0;
  const size_t dw_bytes = (size_t)q.G * q.Cout_g * q.Cin_g * q.KH * q.KW * sizeof(float);
  hipStream_t st = vsp::as_stream(stream);
  if (!q.accumulate && hipMemsetAsync(q.dw, 0, dw_bytes, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d_wgrad: memset failed");
  if (q.B == 0 || q.OH == 0 || q.OW == 0) return VSP_OK;
  VSP_REQUIRE(q.x && q.dy, "conv2d_wgrad: null input");
  k.x = q.x; k.dy = q.dy; k.dw = q.dw; k.xs = q.x_scale; k.dys = q.dy_scale;
  k.B = q.B; k.Cin_g = q.Cin_g; k.H = q.H; k.W = q.W; k.G = q.G; k.Cout_g = q.Cout_g; k.OH = q.OH; k.OW = q.OW;
  k.KH = q.KH; k.KW = q.KW; k.stride = q.stride;
  k.x_gs = q.x_shared ? 0 : q.Cin_g;
  k.x_coff = q.x_coff; k.dy_coff = q.dy_coff;
  k.x_ch = q.x_ch > 0 ? q.x_ch : (q.x_shared ? q.Cin_g : q.G * q.Cin_g);
  k.dy_ch = q.dy_ch > 0 ? q.dy_ch : q.G * q.Cout_g;
  VSP_REQUIRE(k.x_coff + (q.G - 1) * k.x_gs + q.Cin_g <= k.x_ch && k.dy_coff + q.G * q.Cout_g <= k.dy_ch,
              "conv2d_wgrad: channel window exceeds the tensor (x %d+%d of %d, dy %d+%d of %d)", k.x_coff, (q.G - 1) * k.x_gs + q.Cin_g,
              k.x_ch, k.dy_coff, q.G * q.Cout_g, k.dy_ch);
  const int xw = (WG_PX - 1) * q.stride + (q.KW - 1) * dmax + 1;
  const int xw4 = (xw + 3) / 4;
  VSP_REQUIRE(q.KH * xw4 <= 128, "conv2d_wgrad: row segment with halo too wide for the staging layout (stride %d, dilation %d)", q.stride, dmax);
  const int xj = q.KH * xw4 <= 64 ? 1 : 2;
  k.xwp = 4 * xw4;
  int plane = q.KH * k.xwp;
  while (plane % 32 != 2) plane += 2;  // ci rows 2 (mod 32) words apart, 8-byte aligned
  k.plane = plane;
  k.segs = (q.OW + WG_PX - 1) / WG_PX;
  const int64_t chunks = (int64_t)q.B * q.OH * k.segs;
  VSP_REQUIRE(chunks < ((int64_t)1 << 31), "conv2d_wgrad: too many pixels");
  k.chunks = (int)chunks;
  // tile shape by the group's channel counts
  // (two-item staging -- wide stride-2 slabs -- doubles the prefetch registers: it keeps to the 16 / 32-channel X tiles)
  int wco, nb;
  if (q.Cout_g > 32 || xj == 2) { wco = 4; nb = q.Cin_g > 16 ? 2 : 1; }
  else if (q.Cout_g > 16) { wco = 2; nb = q.Cin_g > 32 ? 2 : 1; }
  else { wco = 1; nb = 1; }
  const int co_t = 16 * wco, ci_t = 16 * nb * (4 / wco);
  const int tiles = ((q.Cout_g + co_t - 1) / co_t) * ((q.Cin_g + ci_t - 1) / ci_t);
  VSP_REQUIRE((int64_t)tiles <= 65535 && q.G <= 65535, "conv2d_wgrad: grid too large");
  // split the pixel dimension so that ~3 workgroups per CU are in flight, every workgroup keeping >= 8 chunks when it can
  int64_t split = (3 * vsp::kNumCU + (int64_t)tiles * q.G - 1) / ((int64_t)tiles * q.G);
  if (split > chunks / 8) split = chunks / 8;
  if (split < 1) split = 1;
  const size_t lds = ((size_t)co_t * DPITCH + (size_t)ci_t * k.plane) * sizeof(float);
  VSP_REQUIRE(lds <= kMaxLds, "conv2d_wgrad: row segment with halo does not fit LDS (dilation %d)", dmax);
  dim3 grid((unsigned)split, (unsigned)tiles, (unsigned)q.G);
  const bool k3 = q.KH == 3;
  int rc;
  if (wco == 4 && nb == 2) rc = k3 ? launch_wgrad<9, 4, 2>(k, xj, grid, lds, st) : launch_wgrad<1, 4, 2>(k, xj, grid, lds, st);
  else if (wco == 4) rc = k3 ? launch_wgrad<9, 4, 1>(k, xj, grid, lds, st) : launch_wgrad<1, 4, 1>(k, xj, grid, lds, st);
  else if (wco == 2 && nb == 2) rc = k3 ? launch_wgrad<9, 2, 2>(k, xj, grid, lds, st) : launch_wgrad<1, 2, 2>(k, xj, grid, lds, st);
  else if (wco == 2) rc = k3 ? launch_wgrad<9, 2, 1>(k, xj, grid, lds, st) : launch_wgrad<1, 2, 1>(k, xj, grid, lds, st);
  else rc = k3 ? launch_wgrad<9, 1, 1>(k, xj, grid, lds, st) : launch_wgrad<1, 1, 1>(k, xj, grid, lds, st);
  if (rc != VSP_OK) return rc;
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

// Weight gradient of the path's convolutions on fp32 MFMA for gfx950 (the training step's conv backward;
// reference op/conv2d_gradfix.py:176-206 -> aten::cudnn_convolution_backward_weight, and on current torch simply autograd of
// F.conv2d / F.conv_transpose2d, op/conv2d_gradfix.py:78-92).
//
//   dW[g][co][ci][ky][kx] = sum_{b,oy,ox} dY[b, g Cout_g + co, oy, ox] * dys[b, .]  *  X[b, xc(g) + ci, oy s + ky d_g - p_g, ox s + kx d_g - p_g] * xs[b, .]
//
// As a GEMM: M = output channels, N = (input channel, tap), K = every output pixel of the batch.  One 4-wave workgroup owns a
// (16 WCO) x (16 NB WCI) block of (co, ci) with all KH*KW taps of one group -- wave (wco, wci) its 16 co x 16 NB ci -- and walks a
// consecutive run of K (down one column segment) in chunks of one output-row segment (64 pixels; 2 x 32 / 4 x 16 on small maps):
//   * tile shapes: 64 co x 32 ci (WCO 4, NB 2) for ordinary layers, 32 x 64 (WCO 2, WCI 2, NB 2) and 16 x 64 (WCI 4) for the
//     narrow dilated branches of a SMART layer (16 / 32 output channels per group: a 64-co tile would be 3/4 empty);
//   * the dY slab [co][64 px] and the X slab [ci][KH rows][row segment + halo] of chunk i+1 are fetched into REGISTERS (16-byte
//     loads; every thread keeps one (row, column quad) of the X slab and walks the channels, so the border masks are one row flag
//     and four column bits per chunk) before the MFMA loop of chunk i and written to LDS after it -- the global latency hides
//     under ~150 MFMAs per wave; the per-sample scales of a modulated layer (modulate-input / demodulate-output form) are folded
//     in at the LDS write;
//   * a k-step = 4 pixels: one A fragment (dY) serves all taps and ci blocks, B fragments are the tap-shifted views of the X slab;
//     3x3 layers run the chunk's MFMA loop FULLY UNROLLED (round 3; template arguments S = stride, TCL = chunk shape): one LDS base
//     register per (ci block, tap), the pixel offset an immediate, the nine B fragments of unit u + 1 read under the MFMAs of unit u;
//   * the partial sums of the workgroups that share a (co, ci) block go to private copies of dW in a caller-provided workspace and a
//     second kernel adds them (sliced over the copies when dW is small); without a workspace: fp32 atomics (order not fixed, ~1e-6);
//     the split matches the RESIDENT workgroups (2 per CU at 208 registers) so that the grid is one round;
//   * 1x1 layers with at most 4 input channels (FromRGB) are a stream, not a GEMM: wgrad_fewin_kernel.
// Groups: true groups (x channels g Cin_g ...), or a SHARED input with per-group dilation / padding (the four SMART branches in
// one launch).  LDS pitches: channel rows 2 (mod 32) words apart: the 16 x 2 lanes of an access group hit 32 distinct banks.
#include "vsp_common.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));

#ifndef VSP_WG_UNROLL
#define VSP_WG_UNROLL 2
#endif
#ifndef VSP_WG_SGB   // 1: interleave one LDS read per MFMA in the unrolled loop (sched_group_barrier)
#define VSP_WG_SGB 1
#endif
#ifndef VSP_WG_WALK  // 1: a workgroup walks a consecutive run of chunks down a column segment; 0: chunks strided by the grid
#define VSP_WG_WALK 1
#endif
#ifndef VSP_WG_ABL   // tuning builds only: 1 no atomics, 2 no MFMAs, 4 no staging after the first chunk
#define VSP_WG_ABL 0
#endif
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
  int tcl, rbs, nrows, rstep, rmul;   // chunk = 2^tcl columns x (64 >> tcl) rows; slab rows, input-row step of a slab row, slab rows per tap row
  float* work;       // null: fp32 atomics into dw; else [gridDim.x][dw elements] partial copies, plain stores
  int64_t dw_elems;
  float scale;       // every partial sum is multiplied by it (the equalized-lr factor of the layer: dL/dW_param = scale dL/dW_conv)
};

// S: 0 = any chunk shape and stride (run-time LDS addresses), 1 / 2 = ROW-SEGMENT chunks (tcl = 6) of a stride-S layer: the MFMA loop
// is fully unrolled with one LDS base per (ci block, tap) and the pixel offset as an immediate, B fragments fetched one unit ahead.
// TCL (S != 0): log2 of the chunk width -- 6: one row segment of 64 pixels; 5 / 4: 2 x 32 / 4 x 16 pixels of a 32- / 16-wide map (dense
// slab rows): the row part of the offset is added to the bases once per chunk row.
template <int NTAP, int WCO, int NB, int XJ, int S = 0, int TCL = 6>
__global__ __launch_bounds__(WG_NT, S != 0 ? 2 : 1) void conv_wgrad_kernel(const WgradK p) {
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
  const int KW = NTAP == 1 ? 1 : p.KW;
  const int d = p.dil[p.per_group ? g : 0], pad = p.pad[p.per_group ? g : 0], s = S ? S : p.stride;
  const int XWP = p.xwp, plane = p.plane;
  const int TC = 1 << p.tcl, TR = WG_PX >> p.tcl;                 // chunk: TR output rows x TC columns (TR > 1: small maps)
  const int xw = (TC - 1) * s + (KW - 1) * d + 1 + ((4 - (pad & 3)) & 3);   // slab row of this group (aligned origin)
  const int xw4 = (xw + 3) >> 2;
  const int rstep = p.rstep ? p.rstep : d, rmul = p.rmul ? p.rmul : d;   // compact tap rows (step d) / dense rows (tap ky at row ky d)
  const int NR = p.nrows;                                         // slab rows: KH (tap rows, one output row) or every input row of TR rows
  const int xc0 = p.x_coff + g * p.x_gs + ci0, yc0 = p.dy_coff + g * p.Cout_g + co0;
  const int64_t xplane = (int64_t)p.H * p.W, yplane = (int64_t)p.OH * p.OW;

  // ---- staging roles.  X: item = lane + 64 j -> (slab row, column quad), wave + 4 it -> channel (XJ = 2: slab rows of more than
  //      64 / KH quads, i.e. stride 2).  dY: tid / 16 + 16 it -> channel, tid % 16 -> quad.
  //      Every load is a 16-byte load at a CLAMPED address and every border case is arithmetic on its result -- no branch: a
  //      divergent scalar-fallback path made the compiler wait for the loads in flight at each join (vmcnt(2) between the
  //      loads of one prefetch: the staging alone took as long as all the MFMAs).  The slab starts at a column that is a
  //      multiple of 4 in image coordinates, so a quad is either left of the image, inside it, or cut by the RIGHT border only:
  //      that one loads the last four pixels of the row and shifts (sh = 1..3 pixels, zeros enter from the right).
  const int xs_al = (4 - (pad & 3)) & 3;                         // slab column of input column ox0 s - pad (ox0 s is a multiple of 64)
  int x_row[XJ], x_q[XJ];
  bool x_on[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int item = lane + 64 * j;
    x_row[j] = item / xw4;
    x_q[j] = item - x_row[j] * xw4;
    x_on[j] = x_row[j] < NR;
  }
  const int d_row = tid >> 4, d_q = tid & 15;
  float4 xr[XJ][NITX], dr[NITD];
  float xsc[NITX], dsc[NITD], x_okf[XJ];
  int x_sh[XJ], d_sh = 0;
  float d_okf = 0.f;
  auto shifted = [](float4 v, int sh) {   // v holds pixels c .. c+3, the quad wants c+sh .. c+sh+3 (beyond the row: zero)
    if (sh & 1) v = make_float4(v.y, v.z, v.w, 0.f);
    if (sh & 2) v = make_float4(v.z, v.w, 0.f, 0.f);
    return v;
  };
  auto fetch = [&](int ch) {
#if VSP_WG_WALK
    // chunk order: rows of one column segment are consecutive, and a workgroup owns a consecutive run: two of the three slab rows of a
    // chunk are the previous chunk's (L2), and the run stays inside a few pages of every channel plane
    const int rb = ch % p.rbs, cs = ch / p.rbs;
    const int seg = cs % p.segs, b = cs / p.segs;
#else
    const int seg = ch % p.segs, row = ch / p.segs;
    const int rb = row % p.rbs, b = row / p.rbs;
#endif
    const int oy0 = rb * TR, ox0 = seg * TC;
    {  // dY: quad d_q = pixels 4 d_q .. + 3 of the chunk = row ty, columns tx .. tx + 3
      const int ty = (4 * d_q) >> p.tcl, tx = (4 * d_q) & (TC - 1);
      const int oy = oy0 + ty, ox = ox0 + tx;
      const int oxc = min(ox, p.OW - 4), oyc = min(oy, p.OH - 1);
      d_sh = ox - oxc;
      d_okf = (ox < p.OW && oy < p.OH) ? 1.f : 0.f;
#pragma unroll
      for (int it = 0; it < NITD; ++it) {
        const int c = d_row + 16 * it;
        const bool cok = co0 + c < p.Cout_g;
        const int ch_ = yc0 + (cok ? c : 0);
        const float* src = p.dy + ((int64_t)b * p.dy_ch + ch_) * yplane + (int64_t)oyc * p.OW;
        const f32x4u v = *reinterpret_cast<const f32x4u*>(src + oxc);
        dr[it] = make_float4(v[0], v[1], v[2], v[3]);
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
      const int iy = oy0 * s + x_row[j] * rstep - pad, ix = ox0 * s - pad - xs_al + 4 * x_q[j];
      const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 4);
      x_sh[j] = ix - ixc;                                   // > 0 only at the right border (>= 4: the quad is outside)
      x_okf[j] = (x_on[j] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? 1.f : 0.f;
      const float* rowp = p.x + ((int64_t)b * p.x_ch + xc0) * xplane + (int64_t)iyc * p.W + ixc;
#pragma unroll
      for (int it = 0; it < NITX; ++it) {
        const int c = wave + 4 * it;
        const bool cok = ci0 + c < p.Cin_g;   // wave-uniform
        const f32x4u v = *reinterpret_cast<const f32x4u*>(rowp + (int64_t)(cok ? c : 0) * xplane);
        xr[j][it] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const float sc = dsc[it] * d_okf;
      const float4 v = shifted(dr[it], d_sh);
      float* dst = Dl + (d_row + 16 * it) * DPITCH + 4 * d_q;
      *reinterpret_cast<f32x2a*>(dst) = f32x2a{v.x * sc, v.y * sc};
      *reinterpret_cast<f32x2a*>(dst + 2) = f32x2a{v.z * sc, v.w * sc};
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      if (!x_on[j]) continue;
#pragma unroll
      for (int it = 0; it < NITX; ++it) {
        const float sc = xsc[it] * x_okf[j];
        const float4 v = shifted(xr[j][it], x_sh[j]);
        float* dst = Xl + (wave + 4 * it) * plane + x_row[j] * XWP + 4 * x_q[j];
        *reinterpret_cast<f32x2a*>(dst) = f32x2a{v.x * sc, v.y * sc};
        *reinterpret_cast<f32x2a*>(dst + 2) = f32x2a{v.z * sc, v.w * sc};
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
  const float* bp = Xl + (wci * NB * 16 + r) * plane + kq * s + xs_al;
  const float* bt[NB][NTAP];   // (S != 0) LDS base of (ci block, tap): the pixel offset of a k-step is an immediate
  if constexpr (S != 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < NTAP; ++t) bt[nb][t] = bp + nb * 16 * plane + (t / 3) * rmul * XWP + (t % 3) * d;
  }

#if VSP_WG_WALK
  const int ch_lo = (int)((int64_t)p.chunks * blockIdx.x / gridDim.x), ch_hi = (int)((int64_t)p.chunks * (blockIdx.x + 1) / gridDim.x);
  const int ch_step = 1;
#else
  const int ch_lo = blockIdx.x, ch_hi = p.chunks, ch_step = gridDim.x;
#endif
  int ch = ch_lo;
  if (ch < ch_hi) fetch(ch);
  for (; ch < ch_hi; ch += ch_step) {
#if VSP_WG_ABL & 4
    if (ch == ch_lo) { commit(); __syncthreads(); }
#else
    __syncthreads();   // the MFMAs of the previous chunk have read the slabs
    commit();
    __syncthreads();
    if (ch + ch_step < ch_hi) fetch(ch + ch_step);
#endif
#if VSP_WG_ABL & 2
    continue;
#endif
    if constexpr (S != 0 && NTAP == 9) {
      // unit u = (k-step u / NB, ci block u % NB): 9 MFMAs.  The nine B fragments of unit u + 1 (and the A fragment of the next
      // k-step) are read while the MFMAs of unit u issue: a wave alone keeps the matrix pipe fed (with the reads issued right before
      // their MFMAs the LDS latency showed once per 8 MFMAs: 82 % of the MFMA rate with the staging removed, VSP_WG_ABL = 4).
      constexpr int NU = (WG_PX / 4) * NB, SPR = (1 << TCL) / 4;   // units; k-steps per chunk row
      float bq[2][NTAP], aq[2];
      const float* br[NB][NTAP];   // bases of the chunk row the NEXT unit reads
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) br[nb][t] = bt[nb][t];
      aq[0] = ap[0];
#pragma unroll
      for (int t = 0; t < NTAP; ++t) bq[0][t] = br[0][t][0];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int step = u / NB, nb = u % NB;
        if (u + 1 < NU) {
          const int step1 = (u + 1) / NB, nb1 = (u + 1) % NB;
          if (nb1 == 0) {
            aq[step1 & 1] = ap[4 * step1];
            if (TCL < 6 && step1 % SPR == 0) {   // next chunk row: S slab rows down
#pragma unroll
              for (int n2 = 0; n2 < NB; ++n2)
#pragma unroll
                for (int t = 0; t < NTAP; ++t) br[n2][t] += S * XWP;
            }
          }
#pragma unroll
          for (int t = 0; t < NTAP; ++t) bq[(u + 1) & 1][t] = br[nb1][t][4 * (step1 % SPR) * S];
        }
#pragma unroll
        for (int t = 0; t < NTAP; ++t) acc[nb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[step & 1], bq[u & 1][t], acc[nb][t], 0, 0, 0);
#if VSP_WG_SGB
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll VSP_WG_UNROLL
    for (int k0 = 0; k0 < WG_PX; k0 += 4) {
      const float a = ap[k0];
      const int kb = ((k0 >> p.tcl) * s) * XWP + (k0 & (TC - 1)) * s;   // pixel k0 = chunk row k0 / TC, column k0 % TC
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int ky = t / 3, kx = t - 3 * ky;   // (NTAP = 9: 3x3; NTAP = 1: the single tap)
          const float bv = bp[nb * 16 * plane + ky * rmul * XWP + kb + kx * d];
          acc[nb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[nb][t], 0, 0, 0);
        }
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
      const int64_t off = (((int64_t)g * p.Cout_g + co) * p.Cin_g + ci) * NTAP;
#if VSP_WG_ABL & 1
      if (p.chunks < 0)
#endif
      if (p.work) {   // this workgroup's private copy: every (split index, tile) pair is written exactly once
        float* dst = p.work + (int64_t)blockIdx.x * p.dw_elems + off;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) dst[t] = acc[nb][t][j] * p.scale;
      } else {
#pragma unroll
        for (int t = 0; t < NTAP; ++t) unsafeAtomicAdd(p.dw + off + t, acc[nb][t][j] * p.scale);
      }
    }
  }
}

// dw[i] (+)= sum over the split copies
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ work, int64_t n, int copies,
                                                            int accumulate) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = accumulate ? reinterpret_cast<const float4*>(dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < copies; ++c) {
      const float4 v = reinterpret_cast<const float4*>(work + (int64_t)c * n)[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[i] = a;
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float a = accumulate ? dw[i] : 0.f;
    for (int c = 0; c < copies; ++c) a += work[(int64_t)c * n + i];
    dw[i] = a;
  }
}

// 1x1, stride 1, at most 4 input channels (FromRGB of the discriminator, the RGB side of a skip: 3 -> 16 / 64 at 512^2): a stream, not
// a GEMM -- the MFMA kernel above spends a 16 x 64 tile on 3 columns (0.27 ms for 80 MB).  One workgroup = a pixel range of one image and
// 16 output channels; every thread keeps 16 x CIN sums over its pixels (16-byte loads), wave reduction, one fp32 atomic per wave and sum.
template <int CIN>
__global__ __launch_bounds__(256) void wgrad_fewin_kernel(const WgradK p, int px_per_block) {
  const int b = blockIdx.y, co0 = blockIdx.z * 16;
  const int64_t hw = (int64_t)p.H * p.W;
  const int64_t p0 = (int64_t)blockIdx.x * px_per_block, p1 = p0 + px_per_block < hw ? p0 + px_per_block : hw;
  const float* xb = p.x + ((int64_t)b * p.x_ch + p.x_coff) * hw;
  const float* yb = p.dy + ((int64_t)b * p.dy_ch + p.dy_coff + co0) * hw;
  const int nco = p.Cout_g - co0 < 16 ? p.Cout_g - co0 : 16;
  float acc[16][CIN];
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int i = 0; i < CIN; ++i) acc[c][i] = 0.f;
  const bool vec = (hw & 3) == 0 && (px_per_block & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.dy)) & 15) == 0;
  if (vec) {
    for (int64_t q = p0 + 4 * threadIdx.x; q < p1; q += 4 * 256) {
      float4 xv[CIN];
#pragma unroll
      for (int i = 0; i < CIN; ++i) xv[i] = i < p.Cin_g ? *reinterpret_cast<const float4*>(xb + i * hw + q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c >= nco) break;
        const float4 g = *reinterpret_cast<const float4*>(yb + c * hw + q);
#pragma unroll
        for (int i = 0; i < CIN; ++i) acc[c][i] += g.x * xv[i].x + g.y * xv[i].y + g.z * xv[i].z + g.w * xv[i].w;
      }
    }
  } else {
    for (int64_t q = p0 + threadIdx.x; q < p1; q += 256) {
      float xv[CIN];
#pragma unroll
      for (int i = 0; i < CIN; ++i) xv[i] = i < p.Cin_g ? xb[i * hw + q] : 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c >= nco) break;
        const float g = yb[c * hw + q];
#pragma unroll
        for (int i = 0; i < CIN; ++i) acc[c][i] += g * xv[i];
      }
    }
  }
  __shared__ float red[4][16 * CIN];
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int i = 0; i < CIN; ++i) {
      float v = acc[c][i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c * CIN + i] = v;
    }
  __syncthreads();
  if (threadIdx.x < 16 * CIN) {
    const int c = threadIdx.x / CIN, i = threadIdx.x - c * CIN;
    if (c < nco && i < p.Cin_g) {
      const float sc = (p.xs ? p.xs[(int64_t)b * p.x_ch + p.x_coff + i] : 1.f) * (p.dys ? p.dys[(int64_t)b * p.dy_ch + p.dy_coff + co0 + c] : 1.f) * p.scale;
      const float v = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * sc;
      const int64_t off = (int64_t)(co0 + c) * p.Cin_g + i;
      if (p.work) p.work[((int64_t)b * gridDim.x + blockIdx.x) * p.dw_elems + off] = v;   // one copy per (image, pixel range)
      else unsafeAtomicAdd(p.dw + off, v);
    }
  }
}

// dw[e] (+)= sum over the copies: one wave per element (a few hundred elements, up to ~1000 copies)
__global__ __launch_bounds__(64) void fewin_reduce_kernel(float* __restrict__ dw, const float* __restrict__ work, int64_t n, int copies,
                                                           int accumulate) {
  const int64_t e = blockIdx.x;
  float a = 0.f;
  for (int c = threadIdx.x; c < copies; c += 64) a += work[(int64_t)c * n + e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (threadIdx.x == 0) dw[e] = (accumulate ? dw[e] : 0.f) + a;
}

// pixel range per workgroup of the stream form: ~1 workgroup per CU (the partial sums meet in `copies` = ranges * B copies of dw)
inline bool fewin_form(const vsp_conv_wgrad_params& q) {
  return q.KH == 1 && q.KW == 1 && q.G == 1 && q.stride == 1 && q.Cin_g <= 4 && q.pad == 0 && !q.per_group_geometry && q.OH == q.H && q.OW == q.W;
}
inline int64_t fewin_range(const vsp_conv_wgrad_params& q) {
  const int64_t hw = (int64_t)q.H * q.W;
  const int co_blocks = (q.Cout_g + 15) / 16;
  int64_t per = hw * q.B * co_blocks / vsp::kNumCU / co_blocks / q.B;
  per = per < 4096 ? 4096 : per;
  return (per + 1023) / 1024 * 1024;
}

// The same sum when dw is small and the copies are many (64-channel layers at 512^2: 9216 quads, 384 copies -- one thread per quad
// walking every copy is a 36-workgroup launch bound by its own load latency, 90 us): SL threads share the copies of a quad.
template <int SL>
__global__ __launch_bounds__(64 * SL) void wgrad_reduce_sliced_kernel(float* __restrict__ dw, const float* __restrict__ work, int64_t n4,
                                                                       int copies, int accumulate) {
  __shared__ float4 red[SL][64];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int sl = threadIdx.y;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
#pragma unroll 4
    for (int c = sl; c < copies; c += SL) {
      const float4 v = reinterpret_cast<const float4*>(work)[(int64_t)c * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[sl][threadIdx.x] = a;
  __syncthreads();
  if (sl == 0 && i < n4) {
    a = accumulate ? reinterpret_cast<const float4*>(dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const float4 v = red[k][threadIdx.x];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[i] = a;
  }
}

template <int NTAP, int WCO, int NB, int XJ, int S, int TCL>
int launch_wgrad_fast(const WgradK& k, dim3 grid, size_t lds, hipStream_t st) {
  static vsp::LdsAttrOnce attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wgrad_kernel<NTAP, WCO, NB, XJ, S, TCL>), (int)kMaxLds, "conv2d_wgrad")) return rc;
  conv_wgrad_kernel<NTAP, WCO, NB, XJ, S, TCL><<<grid, WG_NT, lds, st>>>(k);
  return VSP_OK;
}

template <int NTAP, int WCO, int NB>
int launch_wgrad(const WgradK& k, int xj, dim3 grid, size_t lds, hipStream_t st) {
  static vsp::LdsAttrOnce attr1, attr2;  // (slabs beyond the default 64 KB limit: stride-2 rows, the 64-channel X tiles); per device
  // 3x3 layers on chunks of 64 / 2 x 32 / 4 x 16 pixels: the unrolled loop (stride 1 stages one item per thread, stride 2 two)
  if constexpr (NTAP == 9) {
    if (k.tcl >= 4 && xj == k.stride && !vsp::tune_env("VSP_WGRAD_GENERIC")) {
      if (k.stride == 1) {
        if (k.tcl == 6) return launch_wgrad_fast<NTAP, WCO, NB, 1, 1, 6>(k, grid, lds, st);
        if (k.tcl == 5) return launch_wgrad_fast<NTAP, WCO, NB, 1, 1, 5>(k, grid, lds, st);
        return launch_wgrad_fast<NTAP, WCO, NB, 1, 1, 4>(k, grid, lds, st);
      }
      if constexpr (WCO == 4) {   // (two staging items per lane always take the 64-co tile)
        if (k.tcl == 6) return launch_wgrad_fast<NTAP, WCO, NB, 2, 2, 6>(k, grid, lds, st);
        if (k.tcl == 5) return launch_wgrad_fast<NTAP, WCO, NB, 2, 2, 5>(k, grid, lds, st);
        return launch_wgrad_fast<NTAP, WCO, NB, 2, 2, 4>(k, grid, lds, st);
      }
    }
  }
  if (xj == 1) {
    if (int rc = attr1.ensure(reinterpret_cast<const void*>(conv_wgrad_kernel<NTAP, WCO, NB, 1>), (int)kMaxLds, "conv2d_wgrad")) return rc;
    conv_wgrad_kernel<NTAP, WCO, NB, 1><<<grid, WG_NT, lds, st>>>(k);
  } else {
    if (int rc = attr2.ensure(reinterpret_cast<const void*>(conv_wgrad_kernel<NTAP, WCO, NB, 2>), (int)kMaxLds, "conv2d_wgrad")) return rc;
    conv_wgrad_kernel<NTAP, WCO, NB, 2><<<grid, WG_NT, lds, st>>>(k);
  }
  return VSP_OK;
}

struct Plan {
  int dmax, xw4, xj, wco, nb, tiles, segs, tcl, nrows, rbs;
  int64_t chunks, split;
};

// tile shape and split of a (validated) parameter block
Plan make_plan(const vsp_conv_wgrad_params& q) {
  Plan pl{};
  pl.dmax = 1;
  for (int g = 0; g < (q.per_group_geometry ? (q.G < 4 ? q.G : 4) : 1); ++g) {
    const int d = q.per_group_geometry ? q.dil_g[g] : q.dil;
    pl.dmax = d > pl.dmax ? d : pl.dmax;
  }
  // chunk shape: one row segment of 64 pixels, or -- maps narrower than 64 -- TR rows x TC columns (TC a power of two) when the dense
  // slab of those rows (every input row between the first and the last tap row) still fits the staging layout
  pl.tcl = 6;
  for (int tcl = 2; tcl < 6; ++tcl)
    if ((1 << tcl) >= q.OW) { pl.tcl = tcl; break; }
  for (;; pl.tcl = 6) {
    const int tc = 1 << pl.tcl, tr = WG_PX >> pl.tcl;
    const int xw = (tc - 1) * q.stride + (q.KW - 1) * pl.dmax + 1 + 3;   // + alignment slack of the slab origin
    pl.xw4 = (xw + 3) / 4;
    pl.nrows = tr == 1 ? q.KH : (tr - 1) * q.stride + (q.KH - 1) * pl.dmax + 1;
    if (tr == 1 || pl.nrows * pl.xw4 <= 128) break;
  }
  pl.xj = pl.nrows * pl.xw4 <= 64 ? 1 : 2;
  // (two-item staging -- wide stride-2 slabs -- doubles the prefetch registers: it keeps to the 16 / 32-channel X tiles)
  if (q.Cout_g > 32 || pl.xj == 2) { pl.wco = 4; pl.nb = q.Cin_g > 16 ? 2 : 1; }
  else if (q.Cout_g > 16) { pl.wco = 2; pl.nb = q.Cin_g > 32 ? 2 : 1; }
  else { pl.wco = 1; pl.nb = 1; }
  const int co_t = 16 * pl.wco, ci_t = 16 * pl.nb * (4 / pl.wco);
  pl.tiles = ((q.Cout_g + co_t - 1) / co_t) * ((q.Cin_g + ci_t - 1) / ci_t);
  pl.segs = (q.OW + (1 << pl.tcl) - 1) >> pl.tcl;
  pl.rbs = (q.OH + (WG_PX >> pl.tcl) - 1) / (WG_PX >> pl.tcl);
  pl.chunks = (int64_t)q.B * pl.rbs * pl.segs;
  // split the pixel dimension so that every CU slot holds a workgroup, every workgroup keeping >= 8 chunks when it can
  // (resident workgroups per CU: two at the 200+ registers of the 3x3 kernels with 32 accumulator columns or a 64-channel X tile, else 3.
  //  A grid of 3 per CU on 2 slots ran as one full round and one half-empty one: 64 -> 64 at 512^2 833 -> 775 us, 64 -> 4 x 16: 1138 -> 931 us.
  //  Stride 2 measured the other way round -- 3 per CU: 408 / 376 us, 2 per CU: 433 / 399 us on 64 -> 128 at 256^2 / 256 -> 512 at 64^2.)
  const int per_cu = (q.KH == 3 && q.stride == 1 && (pl.nb == 2 || pl.wco != 4)) ? 2 : 3;
  pl.split = (per_cu * vsp::kNumCU + (int64_t)pl.tiles * q.G - 1) / ((int64_t)pl.tiles * q.G);
  if (pl.split > pl.chunks / 8) pl.split = pl.chunks / 8;
  if (pl.split < 1) pl.split = 1;
  return pl;
}

}  // namespace

extern "C" size_t vsp_conv2d_wgrad_work_floats(const vsp_conv_wgrad_params* pp) {
  if (!pp) return 0;
  const vsp_conv_wgrad_params& q = *pp;
  if (q.B <= 0 || q.G < 1 || q.Cin_g < 1 || q.Cout_g < 1 || q.OH <= 0 || q.OW <= 0 || q.KH < 1 || q.KW < 1 || q.stride < 1) return 0;
  if (fewin_form(q)) return (size_t)((((int64_t)q.H * q.W + fewin_range(q) - 1) / fewin_range(q)) * q.B) * q.Cout_g * q.Cin_g;   // stream form
  const Plan pl = make_plan(q);
  return (size_t)pl.split * (size_t)q.G * q.Cout_g * q.Cin_g * q.KH * q.KW;
}

extern "C" int vsp_conv2d_wgrad_f32(const vsp_conv_wgrad_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d_wgrad: null params");
  const vsp_conv_wgrad_params& q = *pp;
  VSP_REQUIRE(q.B >= 0 && q.G >= 1 && q.Cin_g >= 1 && q.Cout_g >= 1 && q.H >= 1 && q.W >= 1 && q.OH >= 0 && q.OW >= 0,
              "conv2d_wgrad: bad dimensions");
  VSP_REQUIRE((q.KH == 3 && q.KW == 3) || (q.KH == 1 && q.KW == 1), "conv2d_wgrad: 3x3 and 1x1 kernels only (got %dx%d)", q.KH, q.KW);
  VSP_REQUIRE(q.stride == 1 || q.stride == 2, "conv2d_wgrad: stride must be 1 or 2");
  VSP_REQUIRE(q.dw != nullptr, "conv2d_wgrad: null output");
  VSP_REQUIRE(q.W >= 4 && (q.OW >= 4 || q.OW == 0), "conv2d_wgrad: rows shorter than 4 pixels are not supported (W %d, OW %d)", q.W, q.OW);
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
  const int64_t dw_elems = (int64_t)q.G * q.Cout_g * q.Cin_g * q.KH * q.KW;
  const size_t dw_bytes = (size_t)dw_elems * sizeof(float);
  hipStream_t st = vsp::as_stream(stream);
  const bool use_work = q.work != nullptr && q.work_floats > 0;
  if (!q.accumulate && (!use_work || q.B == 0 || q.OH == 0 || q.OW == 0) && hipMemsetAsync(q.dw, 0, dw_bytes, st) != hipSuccess)
    return vsp::fail(VSP_ELAUNCH, "conv2d_wgrad: memset failed");
  if (q.B == 0 || q.OH == 0 || q.OW == 0) return VSP_OK;
  VSP_REQUIRE(q.x && q.dy, "conv2d_wgrad: null input");
  k.x = q.x; k.dy = q.dy; k.dw = q.dw; k.xs = q.x_scale; k.dys = q.dy_scale;
  k.scale = q.dw_scale != 0.f ? q.dw_scale : 1.f;
  k.B = q.B; k.Cin_g = q.Cin_g; k.H = q.H; k.W = q.W; k.G = q.G; k.Cout_g = q.Cout_g; k.OH = q.OH; k.OW = q.OW;
  k.KH = q.KH; k.KW = q.KW; k.stride = q.stride;
  k.x_gs = q.x_shared ? 0 : q.Cin_g;
  k.x_coff = q.x_coff; k.dy_coff = q.dy_coff;
  k.x_ch = q.x_ch > 0 ? q.x_ch : (q.x_shared ? q.Cin_g : q.G * q.Cin_g);
  k.dy_ch = q.dy_ch > 0 ? q.dy_ch : q.G * q.Cout_g;
  VSP_REQUIRE(k.x_coff + (q.G - 1) * k.x_gs + q.Cin_g <= k.x_ch && k.dy_coff + q.G * q.Cout_g <= k.dy_ch,
              "conv2d_wgrad: channel window exceeds the tensor (x %d+%d of %d, dy %d+%d of %d)", k.x_coff, (q.G - 1) * k.x_gs + q.Cin_g,
              k.x_ch, k.dy_coff, q.G * q.Cout_g, k.dy_ch);
  if (fewin_form(q)) {   // the stream form (wgrad_fewin_kernel)
    const int64_t hw = (int64_t)q.H * q.W, per = fewin_range(q);
    const int co_blocks = (q.Cout_g + 15) / 16;
    const int64_t ranges = (hw + per - 1) / per, copies = ranges * q.B;
    VSP_REQUIRE(ranges <= 0x7fffffff && q.B <= 65535 && co_blocks <= 65535, "conv2d_wgrad: grid too large");
    const bool copies_fit = use_work && (int64_t)q.work_floats >= copies * dw_elems && vsp::aligned16(q.work);
    if (copies_fit) {   // every (copy, element) is written exactly once: no memset
      k.work = q.work;
      k.dw_elems = dw_elems;
    } else if (!q.accumulate && use_work && hipMemsetAsync(q.dw, 0, dw_bytes, st) != hipSuccess) {
      return vsp::fail(VSP_ELAUNCH, "conv2d_wgrad: memset failed");
    }
    dim3 grid((unsigned)ranges, (unsigned)q.B, (unsigned)co_blocks);
    wgrad_fewin_kernel<4><<<grid, 256, 0, st>>>(k, (int)per);
    if (copies_fit) {
      fewin_reduce_kernel<<<(unsigned)dw_elems, 64, 0, st>>>(q.dw, q.work, dw_elems, (int)copies, q.accumulate ? 1 : 0);
    }
    return vsp::check_launch("conv2d_wgrad");
  }
  const Plan pl = make_plan(q);
  const int xw4 = pl.xw4, xj = pl.xj, wco = pl.wco, nb = pl.nb, tiles = pl.tiles;
  const int64_t chunks = pl.chunks;
  int64_t split = pl.split;
  VSP_REQUIRE(pl.nrows * xw4 <= 128, "conv2d_wgrad: row segment with halo too wide for the staging layout (stride %d, dilation %d)", q.stride, dmax);
  k.xwp = 4 * xw4;
  k.tcl = pl.tcl; k.rbs = pl.rbs; k.nrows = pl.nrows;
  const bool dense = pl.tcl < 6;                  // dense slab rows: row r = input row r of the chunk; else row ky = tap row ky
  k.rstep = dense ? 1 : 0;                        // 0: the group's dilation (set in the kernel)
  k.rmul = dense ? 0 : 1;                         // 0: the group's dilation
  int plane = pl.nrows * k.xwp;
  while (plane % 32 != 2) plane += 2;  // ci rows 2 (mod 32) words apart, 8-byte aligned
  k.plane = plane;
  k.segs = pl.segs;
  VSP_REQUIRE(chunks < ((int64_t)1 << 31), "conv2d_wgrad: too many pixels");
  k.chunks = (int)chunks;
  const int co_t = 16 * wco, ci_t = 16 * nb * (4 / wco);
  VSP_REQUIRE((int64_t)tiles <= 65535 && q.G <= 65535, "conv2d_wgrad: grid too large");
  if (use_work) {   // the copies must fit the caller's workspace; a workspace that is too small for even one copy is an error
    VSP_REQUIRE(vsp::aligned16(q.work) && vsp::aligned16(q.dw), "conv2d_wgrad: dw and work must be 16-byte aligned");
    VSP_REQUIRE((int64_t)q.work_floats >= dw_elems, "conv2d_wgrad: workspace holds %zu floats, one copy of dw needs %lld",
                (size_t)q.work_floats, (long long)dw_elems);
    if (split > (int64_t)q.work_floats / dw_elems) split = (int64_t)q.work_floats / dw_elems;
    k.work = q.work;
    k.dw_elems = dw_elems;
    // a tile the channel counts do not fill leaves holes in a copy, as do blocks without a chunk: those copies start from zero
    // (full tiles and split <= chunks: every element of every copy is stored -- the memset was 56 MB per launch on the big layers)
    const bool holes = q.Cout_g % co_t != 0 || q.Cin_g % ci_t != 0 || split > chunks;
    if (holes && hipMemsetAsync(q.work, 0, (size_t)split * dw_bytes, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d_wgrad: memset failed");
  }
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
  if (use_work) {
    int blocks = (int)((dw_elems / 4 + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > vsp::kMaxStreamBlocks ? vsp::kMaxStreamBlocks : blocks);
    if (dw_elems % 4 == 0 && blocks < 2 * vsp::kNumCU && split >= 32)
      wgrad_reduce_sliced_kernel<16><<<(unsigned)((dw_elems / 4 + 63) / 64), dim3(64, 16), 0, st>>>(q.dw, q.work, dw_elems / 4, (int)split,
                                                                                               q.accumulate ? 1 : 0);
    else
      wgrad_reduce_kernel<<<blocks, 256, 0, st>>>(q.dw, q.work, dw_elems, (int)split, q.accumulate ? 1 : 0);
  }
  return vsp::check_launch("conv2d_wgrad");
}

// out[plane] = sum_i a[plane, i] * b[plane, i]: the gradients of the per-sample input / output scales of a modulated convolution
// (d in_scale[b, ci] = <dXs, x>, d out_scale[b, co] = <dY, y> / out_scale); out[c] = sum_{b, i} x[b, c, i]: bias gradients.
// HBM streams: a plane is split over several workgroups (16-byte loads, block reduction, one fp32 atomic per workgroup) so that a
// handful of 512^2 planes still fills the chip.
namespace {
__device__ __forceinline__ float block_sum_256(float s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void plane_dot_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                         const float* __restrict__ b, int64_t n, int64_t per) {
  const float* ap = a + (int64_t)blockIdx.x * n;
  const float* bp = b + (int64_t)blockIdx.x * n;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < n ? lo + per : n;     // (per is a multiple of 4)
  float s = 0.f;
  if ((n & 3) == 0) {
    for (int64_t i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
      const float4 u = *reinterpret_cast<const float4*>(ap + i), v = *reinterpret_cast<const float4*>(bp + i);
      s = fmaf(u.x, v.x, fmaf(u.y, v.y, fmaf(u.z, v.z, fmaf(u.w, v.w, s))));
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s = fmaf(ap[i], bp[i], s);
  }
  s = block_sum_256(s);
  if (threadIdx.x == 0) {
    if (gridDim.y == 1) out[blockIdx.x] = s;
    else unsafeAtomicAdd(out + blockIdx.x, s);
  }
}

// out[plane] = <a, b> and a[plane, :] *= scale[plane] in the same pass (data gradient of a modulated layer: a = d/d(x s), b = x:
// d/ds = <a, x>, d/dx = a s)
__global__ __launch_bounds__(256) void plane_dot_scale_kernel(float* __restrict__ out, float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ scale, int64_t n, int64_t per) {
  float* ap = a + (int64_t)blockIdx.x * n;
  const float* bp = b + (int64_t)blockIdx.x * n;
  const float sc = scale[blockIdx.x];
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  if ((n & 3) == 0) {
    for (int64_t i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
      float4 u = *reinterpret_cast<const float4*>(ap + i);
      const float4 v = *reinterpret_cast<const float4*>(bp + i);
      s = fmaf(u.x, v.x, fmaf(u.y, v.y, fmaf(u.z, v.z, fmaf(u.w, v.w, s))));
      u.x *= sc; u.y *= sc; u.z *= sc; u.w *= sc;
      *reinterpret_cast<float4*>(ap + i) = u;
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const float u = ap[i];
      s = fmaf(u, bp[i], s);
      ap[i] = u * sc;
    }
  }
  s = block_sum_256(s);
  if (threadIdx.x == 0) {
    if (gridDim.y == 1) out[blockIdx.x] = s;
    else unsafeAtomicAdd(out + blockIdx.x, s);
  }
}

// grid (C, splits): workgroup (c, sp) sums its share of the B planes of channel c
__global__ __launch_bounds__(256) void channel_sum_kernel(float* __restrict__ out, const float* __restrict__ x, int B, int C, int64_t hw,
                                                           int64_t per) {
  const int c = blockIdx.x;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < hw ? lo + per : hw;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* xp = x + ((int64_t)b * C + c) * hw;
    if ((hw & 3) == 0) {
      for (int64_t i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
        const float4 u = *reinterpret_cast<const float4*>(xp + i);
        s += (u.x + u.y) + (u.z + u.w);
      }
    } else {
      for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += xp[i];
    }
  }
  s = block_sum_256(s);
  if (threadIdx.x == 0) {
    if (gridDim.y == 1) out[c] = s;
    else unsafeAtomicAdd(out + c, s);
  }
}

// splits of an n-element plane so that planes * splits ~ 8 workgroups per CU, each at least 4096 elements (multiple of 4)
inline int plane_splits(int64_t planes, int64_t n, int64_t* per) {
  int64_t sp = (8 * vsp::kNumCU + planes - 1) / planes;
  if (sp > n / 4096) sp = n / 4096;
  if (sp < 1) sp = 1;
  if (sp > 65535) sp = 65535;
  int64_t p = (n + sp - 1) / sp;
  p = (p + 3) / 4 * 4;
  *per = p;
  return (int)((n + p - 1) / p);
}
}  // namespace

extern "C" int vsp_plane_dot_f32(float* out, const float* a, const float* b, int64_t planes, int64_t n, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && n >= 0, "plane_dot: negative size");
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(out && (n == 0 || (a && b)), "plane_dot: null pointer");
  VSP_REQUIRE(planes < ((int64_t)1 << 31), "plane_dot: too many planes");
  hipStream_t st = vsp::as_stream(stream);
  int64_t per;
  const int sp = plane_splits(planes, n, &per);
  if (sp > 1 && hipMemsetAsync(out, 0, sizeof(float) * planes, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "plane_dot: memset failed");
  plane_dot_kernel<<<dim3((unsigned)planes, (unsigned)sp), 256, 0, st>>>(out, a, b, n, per);
  return vsp::check_launch("plane_dot");
}

extern "C" int vsp_plane_dot_scale_f32(float* out, float* a, const float* b, const float* scale, int64_t planes, int64_t n,
                                       vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && n >= 0, "plane_dot_scale: negative size");
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(out && scale && (n == 0 || (a && b)), "plane_dot_scale: null pointer");
  VSP_REQUIRE(planes < ((int64_t)1 << 31), "plane_dot_scale: too many planes");
  hipStream_t st = vsp::as_stream(stream);
  int64_t per;
  const int sp = plane_splits(planes, n, &per);
  if (sp > 1 && hipMemsetAsync(out, 0, sizeof(float) * planes, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "plane_dot_scale: memset failed");
  plane_dot_scale_kernel<<<dim3((unsigned)planes, (unsigned)sp), 256, 0, st>>>(out, a, b, scale, n, per);
  return vsp::check_launch("plane_dot_scale");
}

extern "C" int vsp_channel_sum_f32(float* out, const float* x, int B, int C, int64_t hw, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 0 && hw >= 0, "channel_sum: negative size");
  if (C == 0) return VSP_OK;
  VSP_REQUIRE(out && (B == 0 || hw == 0 || x), "channel_sum: null pointer");
  VSP_REQUIRE(C <= 65535 * 32, "channel_sum: too many channels");
  hipStream_t st = vsp::as_stream(stream);
  int64_t per;
  const int sp = plane_splits(C, hw, &per);
  if (sp > 1 && hipMemsetAsync(out, 0, sizeof(float) * C, st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "channel_sum: memset failed");
  channel_sum_kernel<<<dim3((unsigned)C, (unsigned)sp), 256, 0, st>>>(out, x, B, C, hw, per);
  return vsp::check_launch("channel_sum");
}

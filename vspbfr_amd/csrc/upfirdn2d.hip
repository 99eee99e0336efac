// upfirdn2d (zero-insert upsample -> pad/crop -> 2-D FIR -> decimate) for gfx950, with an optional fused epilogue.
//
// Definition (reference CPU statement op/upfirdn2d.py:365-406, CUDA kernels op/upfirdn2d_kernel.cu:49-207):
//   U[s,t]  = x[s/up_y, t/up_x] when s%up_y==0 && t%up_x==0 && inside, else 0      (zero insertion)
//   out[oy,ox] = sum_{ky,kx} k[kh-1-ky][kw-1-kx] * U[oy*down_y + ky - pad_y0, ox*down_x + kx - pad_x0]
// (the taps are flipped: a true convolution; negative pads crop because the index simply shifts).
//
// Two kernels:
//  * fir_tile_kernel<KH,KW>: up=down=1 (the `Blur` of every up/down-sampling StyledConv: the hot case, up to
//    (B,64,513,513) / (B,32,1025,1025) planes).  One 256-thread block produces a 32x64 output tile of one plane:
//    the (32+KH-1)x(64+KW-1) input window is staged once in LDS with coalesced row reads and zero fill, the
//    flipped taps sit in registers (loaded through SGPRs, uniform), and each lane walks 8 rows of one column
//    with a sliding register window, so every LDS word is read KW times and every HBM byte once.
//    Roofline: HBM, 4 B read + 4 B written per output element (+ epilogue operands).
//  * fir_generic_kernel: any up/down/tap count/minor (RGB-skip Upsample: 3 channels, negligible bytes).
#include "vsp_common.h"
#include "vsp_bf16.h"
#include <type_traits>
#include <cstdlib>

namespace {

struct Epi {
  const float* plane_scale;
  const float* noise;
  const float* noise_w;
  const float* act_bias;
  const void* res1;   // same element type as the output (fp32, or bf16 through vsp_upfirdn2d_bf16)
  const void* res2;
  int channels;
  int act;
  float slope;
  float gain;
  int enabled;
  int separable;   // VSP_FIR_SEPARABLE: the taps are an outer product (the caller's promise)
};

__device__ __forceinline__ float epi_apply(const Epi& e, float v, int plane, int oy, int ox, int out_h, int out_w) {
  if (!e.enabled) return v;
  const int c = plane % e.channels;
  const int b = plane / e.channels;
  if (e.plane_scale) v *= e.plane_scale[plane];
  if (e.noise) v += e.noise[((int64_t)b * out_h + oy) * out_w + ox] * e.noise_w[0];
  if (e.act) {
    if (e.act_bias) v += e.act_bias[c];
    v = (v > 0.f ? v : v * e.slope) * e.gain;
  }
  const int64_t o = ((int64_t)plane * out_h + oy) * out_w + ox;
  if (e.res1) v += static_cast<const float*>(e.res1)[o];
  if (e.res2) v += static_cast<const float*>(e.res2)[o];
  return v;
}

// Output tile of one 256-thread block (each thread 2 rows x 4 columns): 32 x 64.  (16 x 128 -- longer, better aligned row
// segments -- measured identical on the large maps and worse at 64^2: the plain blur is bound by bytes in flight per workgroup,
// not by cache-line efficiency, and forcing 8 waves/SIMD (<= 64 VGPRs) spills: 15 % slower; tools/bench_fir.py.)
#ifndef VSP_FIR_TOW
#define VSP_FIR_TOW 64
#endif
constexpr int TOW = VSP_FIR_TOW;          // output tile cols
#ifndef VSP_FIR_RPT
#define VSP_FIR_RPT 2
#endif
constexpr int RPT = VSP_FIR_RPT;          // output rows per thread (x 4 columns)
// tiles per block: template argument of the kernel (1 ships; see the launcher)
constexpr int TOH = RPT * (256 / (TOW / 4)); // output tile rows

// (Round 4, bf16 planes: the window kept PACKED -- 8-byte loads, a tile of dwords in LDS, one 16-byte LDS read per window row and thread, two
// horizontal taps as ONE v_dot2c_f32_bf16 on a pixel pair, i.e. 8 dot products + 3 funnel shifts per window row instead of 16 FMAs per output
// + the unpacking; exact for the path's taps, results within one bf16 unit of the fp32 arithmetic on < 1 % of the elements -- measured 2.26-2.40
// TB/s against 2.4-2.6: the bf16 blur is NOT bound by its arithmetic either.  A tile lives ~7 us with six workgroups per CU: what is in
// flight per CU (6 x 6 KB of window) over that latency IS the 2.4 TB/s; removed.)
// 4 floats that are only 4-byte aligned (image rows of odd width): gfx950 global loads/stores take any dword alignment
using vsp::f32x4u;

// One 256-thread block produces a 32x64 output tile of one plane.  Throughput shape: (1) the (32+KH-1) x 68 input window is
// fetched as three 16-byte loads per thread, ALL issued before the first LDS write (the first version ran ten dependent
// load -> LDS round trips per block); (2) each thread owns a 2x4 output patch: (KH+1) x 2 ds_read_b128 feed 8 outputs;
// (3) the epilogue operands (noise, two residuals) are fetched as 16-byte vectors before the FMAs, the result leaves as
// 16-byte stores.
// T = float or vsp::bf16_t: element type of x, out and the two residuals (arithmetic is fp32 either way; bf16: 2 B read + 2 B
// written per output element, four elements per 8-byte access at halfword alignment).
template <int KH, int KW, typename T, int NTB>
__global__ __launch_bounds__(256) void fir_tile_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                        const float* __restrict__ kern, int in_h, int in_w,
                                                        int out_h, int out_w, int pad_x0, int pad_y0, int tiles_x,
                                                        int tiles_y, int total_tiles, Epi epi) {
  constexpr int TIH = TOH + KH - 1;
  constexpr int TIW4 = (TOW + KW - 1 + 3) / 4;  // float4 per staged row (17: one spare column for KW = 4)
  constexpr int LDW = TIW4 * 4;
  constexpr int NLD = (TIH * TIW4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float tile[TIH * LDW];

  auto decode = [&](int t, int& plane, int& oy0, int& ox0) {
    const int tx_i = t % tiles_x;
    const int ty_i = (t / tiles_x) % tiles_y;
    plane = t / (tiles_x * tiles_y);
    oy0 = ty_i * TOH;
    ox0 = tx_i * TOW;
  };
  // Elements ix .. ix+3 of a row of `w` elements (w >= 4) as ONE 4-element load at a clamped address plus a register shift: no
  // branch around a load.  A divergent scalar-fallback branch for the quads the image border cuts made the compiler wait for
  // every load in flight at each join (`s_waitcnt vmcnt(0)` between the three window loads of a thread: three dependent round
  // trips instead of one -- the same finding as in conv_wgrad.hip).  Positions outside [0, w) come back as zeros.
  auto quad = [&](const T* row, int ix, int w, bool row_ok) -> f32x4u {
    const int ixc = min(max(ix, 0), w - 4);
    f32x4u q = vsp::Elem<T>::load4(row + ixc);
    const int sh = ix - ixc;                 // > 0: cut by the right border, < 0: by the left one
    const int a = sh < 0 ? -sh : sh;
    if (sh > 0) {
      if (a & 1) q = f32x4u{q[1], q[2], q[3], 0.f};
      if (a & 2) q = f32x4u{q[2], q[3], 0.f, 0.f};
    } else if (sh < 0) {
      if (a & 1) q = f32x4u{0.f, q[0], q[1], q[2]};
      if (a & 2) q = f32x4u{0.f, 0.f, q[0], q[1]};
    }
    return (row_ok && a < 4) ? q : f32x4u{0.f, 0.f, 0.f, 0.f};
  };
  auto load_window = [&](int t, f32x4u (&v)[NLD]) {
    int plane, oy0, ox0;
    decode(t, plane, oy0, ox0);
    const int iy0 = oy0 - pad_y0, ix0 = ox0 - pad_x0;
    const T* xp = x + (int64_t)plane * in_h * in_w;
    // (uniform) interior tile: the whole window lies inside the plane -> plain quads, no clamp, no shift, no select.  The blur is bound
    // by its instruction count per output (round 4: ~1000 instructions per 8 outputs in the border form, 0.62 outputs / ns on bf16 planes at
    // 2.5 TB/s, nowhere near HBM), and almost every tile of a large plane is interior.
    const bool interior = iy0 >= 0 && iy0 + TIH <= in_h && ix0 >= 0 && ix0 + LDW <= in_w;
    if (interior) {
      const T* wp = xp + (int64_t)iy0 * in_w + ix0;
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const int idx = min((int)threadIdx.x + 256 * it, TIH * TIW4 - 1);
        const int r = idx / TIW4, c4 = idx - r * TIW4;
        v[it] = vsp::Elem<T>::load4(wp + (int64_t)r * in_w + 4 * c4);
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int idx = min((int)threadIdx.x + 256 * it, TIH * TIW4 - 1);   // (threads past the window re-load its last quad and drop it)
      const int r = idx / TIW4, c4 = idx - r * TIW4;
      const int iy = iy0 + r, ix = ix0 + 4 * c4;
      const int iyc = min(max(iy, 0), in_h - 1);
      v[it] = quad(xp + (int64_t)iyc * in_w, ix, in_w, iy == iyc);
    }
  };
  // flipped taps -> registers (wave-uniform loads)
  float taps[KH][KW];
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) taps[ky][kx] = kern[(KH - 1 - ky) * KW + (KW - 1 - kx)];

  auto process = [&](int t, const f32x4u (&v)[NLD]) {
    int plane, oy0, ox0;
    decode(t, plane, oy0, ox0);
    // this thread's 2 x 4 outputs and their epilogue operands (requested now, consumed after the FMAs)
    const int tx = threadIdx.x % (TOW / 4), ty = threadIdx.x / (TOW / 4);
    const int ox = ox0 + 4 * tx, oyb = oy0 + RPT * ty;
    const bool full = ox + 3 < out_w;
    const bool tile_inside = oy0 + TOH <= out_h && ox0 + TOW <= out_w;
    const int c = plane % epi.channels, b = plane / epi.channels;
    f32x4u nz[RPT], r1[RPT], r2[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      nz[rr] = r1[rr] = r2[rr] = f32x4u{0.f, 0.f, 0.f, 0.f};
      const int oy = oyb + rr;
      if (!epi.enabled) continue;
      // (clamped addresses, uniform branches only: a thread whose quad lies outside the image loads something valid and never stores)
      const int oyc = min(oy, out_h - 1);
      const bool rok = oy < out_h;
      const T* res1 = static_cast<const T*>(epi.res1);
      const T* res2 = static_cast<const T*>(epi.res2);
      if (tile_inside) {   // (uniform) the whole 32 x 64 output tile lies inside the plane: plain 16-byte operand loads
        if (epi.noise) nz[rr] = *reinterpret_cast<const f32x4u*>(epi.noise + ((int64_t)b * out_h + oy) * out_w + ox);
        if (res1) r1[rr] = vsp::Elem<T>::load4(res1 + ((int64_t)plane * out_h + oy) * out_w + ox);
        if (res2) r2[rr] = vsp::Elem<T>::load4(res2 + ((int64_t)plane * out_h + oy) * out_w + ox);
        continue;
      }
      if (epi.noise) {
        const float* nrow = epi.noise + ((int64_t)b * out_h + oyc) * out_w;
        const int oxc = min(max(ox, 0), out_w - 4), sh = ox - oxc;
        f32x4u q = *reinterpret_cast<const f32x4u*>(nrow + oxc);
        if (sh & 1) q = f32x4u{q[1], q[2], q[3], 0.f};
        if (sh & 2) q = f32x4u{q[2], q[3], 0.f, 0.f};
        nz[rr] = (rok && sh < 4) ? q : f32x4u{0.f, 0.f, 0.f, 0.f};
      }
      if (res1) r1[rr] = quad(res1 + ((int64_t)plane * out_h + oyc) * out_w, ox, out_w, rok);
      if (res2) r2[rr] = quad(res2 + ((int64_t)plane * out_h + oyc) * out_w, ox, out_w, rok);
    }
    float pscale = 1.f, nw = 0.f, ab = 0.f;
    if (epi.enabled) {
      if (epi.plane_scale) pscale = epi.plane_scale[plane];
      if (epi.noise) nw = epi.noise_w[0];
      if (epi.act && epi.act_bias) ab = epi.act_bias[c];
    }

#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int idx = threadIdx.x + 256 * it;
      if (idx < TIH * TIW4) *reinterpret_cast<float4*>(tile + idx * 4) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
    }
    __syncthreads();

    float acc[RPT][4];
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[rr][j] = 0.f;
#pragma unroll
    for (int wr = 0; wr < KH + RPT - 1; ++wr) {  // window row wr feeds output row rr with tap row wr - rr
      const float* rp = tile + (RPT * ty + wr) * LDW + 4 * tx;
      const float4 lo = *reinterpret_cast<const float4*>(rp);
      const float4 hi = (KW > 1) ? *reinterpret_cast<const float4*>(rp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) {
        const int ky = wr - rr;
        if (ky < 0 || ky >= KH) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) acc[rr][j] = fmaf(taps[ky][kx], w[j + kx], acc[rr][j]);
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int oy = oyb + rr;
      if (oy >= out_h || ox >= out_w) continue;
      f32x4u o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = acc[rr][j];
        if (epi.enabled) {
          a = fmaf(nz[rr][j], nw, a * pscale);
          if (epi.act) {
            a += ab;
            a = (a > 0.f ? a : a * epi.slope) * epi.gain;
          }
          a += r1[rr][j];
          a += r2[rr][j];
        }
        o4[j] = a;
      }
      T* dst = out + ((int64_t)plane * out_h + oy) * out_w + ox;
      if (full) {
        vsp::Elem<T>::store4(dst, o4);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ox + j < out_w) vsp::Elem<T>::store1(dst + j, o4[j]);
      }
    }
  };
  // NTB adjacent tiles per block, all windows requested up front.  In the micro-benchmark (tools/bench_fir.py) a second window
  // lifts the plain blur from 3.05 to 3.4-3.9 TB/s (bytes in flight); inside the pipeline -- the input was just written by the
  // transposed conv, small maps need the blocks -- one tile per block measures best (rocprofv3: 10.0 ms against 13.3 ms for the
  // 76 blur launches of four steps), so NTB = 1 ships.
  f32x4u v[NTB][NLD];
  const int t0 = NTB * blockIdx.x;
#pragma unroll
  for (int k = 0; k < NTB; ++k)
    if (t0 + k < total_tiles) load_window(t0 + k, v[k]);
#pragma unroll
  for (int k = 0; k < NTB; ++k) {
    if (t0 + k >= total_tiles) break;
    if (k > 0) __syncthreads();
    process(t0 + k, v[k]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// bf16 planes, 4 x 4 taps, up = down = 1 (round 5): the blur as a COLUMN STRIP walk without LDS.  The tile kernel above spends ~49 vector
// instructions per output on bf16 planes (PMC: 394 per wave of 8 outputs per lane: unpacking, LDS staging, border shifts, address
// arithmetic around the 16 FMAs) and holds 2.2-2.9 TB/s: it is bound by its instruction count, not by HBM.  Here a lane owns 8 output
// columns and walks RC output rows: every input row is fetched ONCE per lane as 24 bytes straight from global memory (neighbouring lanes
// overlap by 3 pixels: L1 / L2), converted once (12 shifts / ands), kept with its three predecessors in registers (a rotating window of
// four rows, the loop unrolled by four so that the rotation is a renaming), and feeds 128 FMAs per output row; the epilogue operands and
// the store are 16-byte accesses.  ~24 vector instructions per output, no barrier.  Zero padding: rows outside the plane read through an
// out-of-range buffer offset; columns outside a row (the lanes at a row's ends only) are cleared by six per-lane masks on the packed words.
// EN / ACT: epilogue present / with activation -- compile-time: a uniform `if` in the row loop is a control-flow join, and hipcc waits for
// every load in flight at a join (the two rows the walk keeps ahead).  Planes 1 .. major - 1 only: the first rows of plane 0 start at a
// NEGATIVE offset (left padding before the tensor's first byte), which no range check expresses -- plane 0 goes to the tile kernel.
#ifndef VSP_STRIP_OCC
#define VSP_STRIP_OCC 3
#endif
#ifndef VSP_STRIP_RC
#define VSP_STRIP_RC 32
#endif
// SEP (round 6): the taps are an outer product k[ky][kx] = ty[ky] tx[kx] (every blur of the path: make_kernel([1, 3, 3, 1])) -- an input row is
// reduced along x ONCE when it arrives (32 multiply-adds for the lane's 8 columns) and the rotating window holds those row sums (4 x 8
// instead of 4 x 12 registers); an output row is their 4-tap column sum (32 multiply-adds): 64 instead of 128 per output row.  tx = the
// flipped taps' first row, ty = first column / corner: exact for dyadic taps, otherwise within rounding of the 2-D form.
// TAIL: the columns past the last whole 8-column strip (down-sampling blurs: rows of 2^n + 1 outputs), one lane per (plane, row chunk),
// 2-byte stores predicated by their offset; no epilogue operands.  (ox0 = first column of the strip grid: 0, or the tail's first column.)
template <int RC, bool EN, bool ACT, bool SEP = false, bool TAIL = false>
__global__ __launch_bounds__(256, VSP_STRIP_OCC) void fir_strip_bf16_kernel(vsp::bf16_t* __restrict__ out, const vsp::bf16_t* __restrict__ x, const float* __restrict__ kern,
                                                             int in_h, int in_w, int out_h, int out_w, int pad_x0, int pad_y0, int strips_x,
                                                             int chunks_y, int total, int x_bytes, int out_bytes, int ox0, Epi epi) {
  static_assert(RC % 4 == 0, "the row loop is unrolled by the four window slots");
  static_assert(!(TAIL && EN), "the tail strip carries no epilogue");
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int sx = gid % strips_x;
  const int t = gid / strips_x;
  const int cy = t % chunks_y, plane = 1 + t / chunks_y;
  const int ox = ox0 + 8 * sx, oy0 = cy * RC;
  const int ix0 = ox - pad_x0;
  float taps[4][4];   // flipped: a true convolution
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) taps[ky][kx] = kern[(3 - ky) * 4 + (3 - kx)];
  unsigned cmask[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int a = ix0 + 2 * k;
    cmask[k] = ((a >= 0 && a < in_w) ? 0xffffu : 0u) | ((a + 1 >= 0 && a + 1 < in_w) ? 0xffff0000u : 0u);
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<vsp::bf16_t*>(x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_bytes, 0x00020000);
  const int c = plane % epi.channels, b = plane / epi.channels;
  float pscale = 1.f, nw = 0.f, ab = 0.f;
  if (EN) {
    if (epi.plane_scale) pscale = epi.plane_scale[plane];
    if (epi.noise) nw = epi.noise_w[0];
    if (ACT && epi.act_bias) ab = epi.act_bias[c];
  }
  const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi.noise ? epi.noise : kern), 0,
                                                                       epi.noise ? (int)((int64_t)(out_bytes / 2 / epi.channels) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(epi.res1 ? epi.res1 : (const void*)kern), 0, epi.res1 ? out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(epi.res2 ? epi.res2 : (const void*)kern), 0, epi.res2 ? out_bytes : 0, 0x00020000);
  typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
  // input row r of the chunk (image row oy0 - pad_y0 + r): six packed words, in flight until `unpack`
  u32x4s qa[2];
  u32x2s qb[2];
  const int row_b = in_w * 2;
  const int voff0 = (int)(((int64_t)plane * in_h + (oy0 - pad_y0)) * in_w + ix0) * 2;     // may be negative at the first plane's first rows: out of range, zeros
  auto fetch = [&](int r, int slot) {
    const int iy = oy0 - pad_y0 + r;
    const int vo = (iy >= 0 && iy < in_h) ? voff0 + r * row_b : 0x7ffffff0;
    qa[slot] = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
    qb[slot] = __builtin_bit_cast(u32x2s, __builtin_amdgcn_raw_buffer_load_b64(xrs, vo, 16, 0));
  };
  constexpr int WN = SEP ? 8 : 12;
  float win[4][WN];   // the last four input rows: twelve columns (eleven used), or (SEP) their eight row sums
  float txf[4], tyf[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    txf[k] = taps[0][k];
    tyf[k] = k == 0 ? 1.f : taps[k][0] / taps[0][0];
  }
  auto unpack = [&](int slot, float (&wo)[WN]) {
    const unsigned d[6] = {qa[slot][0] & cmask[0], qa[slot][1] & cmask[1], qa[slot][2] & cmask[2], qa[slot][3] & cmask[3], qb[slot][0] & cmask[4],
                           qb[slot][1] & cmask[5]};
    if constexpr (SEP) {
      float w[12];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        w[2 * k] = vsp::bf16_lo(d[k]);
        w[2 * k + 1] = vsp::bf16_hi(d[k]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) wo[j] = fmaf(txf[3], w[j + 3], fmaf(txf[2], w[j + 2], fmaf(txf[1], w[j + 1], txf[0] * w[j])));
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        wo[2 * k] = vsp::bf16_lo(d[k]);
        wo[2 * k + 1] = vsp::bf16_hi(d[k]);
      }
    }
  };
  const int orow0 = (int)(((int64_t)plane * out_h + oy0) * out_w + ox);     // element index of the chunk's first output
  const int nrow0 = (int)(((int64_t)b * out_h + oy0) * out_w + ox);
  auto emit = [&](int ro, const float (&w0)[WN], const float (&w1)[WN], const float (&w2)[WN], const float (&w3)[WN]) {   // output row ro of the chunk
    const int oy = oy0 + ro;
    const bool ok = oy < out_h;
    f32x4u nz0 = {0.f, 0.f, 0.f, 0.f}, nz1 = nz0;
    u32x4s r1 = {0u, 0u, 0u, 0u}, r2 = r1;
    if (EN) {
      const int no = ok ? (nrow0 + ro * out_w) * 4 : 0x7ffffff0;
      nz0 = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(nrs, no, 0, 0));
      nz1 = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(nrs, no, 16, 0));
      const int eo = ok ? (orow0 + ro * out_w) * 2 : 0x7ffffff0;
      r1 = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(r1rs, eo, 0, 0));
      r2 = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(r2rs, eo, 0, 0));
    }
    float acc[8];
    if constexpr (SEP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(tyf[3], w3[j], fmaf(tyf[2], w2[j], fmaf(tyf[1], w1[j], w0[j])));   // (tyf[0] = 1)
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(taps[0][kx], w0[j + kx], acc[j]);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(taps[1][kx], w1[j + kx], acc[j]);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(taps[2][kx], w2[j + kx], acc[j]);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(taps[3][kx], w3[j + kx], acc[j]);
    }
    if (EN) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = fmaf(j < 4 ? nz0[j & 3] : nz1[j & 3], nw, acc[j] * pscale);
        const unsigned w1r = r1[j >> 1], w2r = r2[j >> 1];
        const float r1f = (j & 1) ? vsp::bf16_hi(w1r) : vsp::bf16_lo(w1r), r2f = (j & 1) ? vsp::bf16_hi(w2r) : vsp::bf16_lo(w2r);
        if (ACT) {
          a += ab;
          a = (a > 0.f ? a : a * epi.slope) * epi.gain;
        }
        {
          // (the tile kernel adds the residual to the ROUNDED product: hipcc contracts `x * gain + res1` to one fused multiply-add in the
          //  vectorised form of this loop but not in the tile kernel's scalar form -- the two then differ in the last bit of one output
          //  in 20 000; contraction off for the two additions)
#pragma clang fp contract(off)
          a = a + r1f;
        }
        a += r2f;
        acc[j] = a;
      }
    }
    const u32x4s o = {vsp::bf16_pack(acc[0], acc[1]), vsp::bf16_pack(acc[2], acc[3]), vsp::bf16_pack(acc[4], acc[5]), vsp::bf16_pack(acc[6], acc[7])};
    if constexpr (TAIL) {
#pragma unroll
      for (int j = 0; j < 8; ++j)   // (no branch: a column past the row leaves through an out-of-range offset)
        __builtin_amdgcn_raw_buffer_store_b16((short)((j & 1) ? (o[j >> 1] >> 16) : (o[j >> 1] & 0xffffu)), ors,
                                              (ok && ox + j < out_w) ? (orow0 + ro * out_w + j) * 2 : 0x7ffffff0, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(o, ors, ok ? (orow0 + ro * out_w) * 2 : 0x7ffffff0, 0, 0);
    }
  };
  // rows 0..2 fill the window; from then on: row r + 2 in flight, row r + 1 ... hmm: two rows ahead
  fetch(0, 0);
  fetch(1, 1);
  unpack(0, win[0]);
  fetch(2, 0);
  unpack(1, win[1]);
  fetch(3, 1);
  unpack(0, win[2]);
  fetch(4, 0);
#pragma unroll 1
  for (int r0 = 0; r0 < RC; r0 += 4) {
    // input rows r0 + 3 .. r0 + 6 complete output rows r0 .. r0 + 3; packed slots alternate: row r lives in slot r & 1
    unpack(1, win[3]);                 // input row r0 + 3
    fetch(r0 + 5, 1);
    emit(r0 + 0, win[0], win[1], win[2], win[3]);
    unpack(0, win[0]);                 // input row r0 + 4
    fetch(r0 + 6, 0);
    emit(r0 + 1, win[1], win[2], win[3], win[0]);
    unpack(1, win[1]);                 // input row r0 + 5
    fetch(r0 + 7, 1);
    emit(r0 + 2, win[2], win[3], win[0], win[1]);
    unpack(0, win[2]);                 // input row r0 + 6
    fetch(r0 + 8, 0);
    emit(r0 + 3, win[3], win[0], win[1], win[2]);
  }
}

__global__ __launch_bounds__(256) void fir_generic_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                           const float* __restrict__ kern, int major, int in_h,
                                                           int in_w, int minor, int kh, int kw, int up_x, int up_y,
                                                           int down_x, int down_y, int pad_x0, int pad_y0, int out_h,
                                                           int out_w, Epi epi) {
  extern __shared__ float ktaps[];  // flipped taps staged once per block
  for (int i = threadIdx.x; i < kh * kw; i += blockDim.x) {
    const int ky = i / kw, kx = i - ky * kw;
    ktaps[i] = kern[(kh - 1 - ky) * kw + (kw - 1 - kx)];
  }
  __syncthreads();
  const int64_t total = (int64_t)major * out_h * out_w * minor;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int uh = in_h * up_y, uw = in_w * up_x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % minor);
    int64_t t = i / minor;
    const int ox = (int)(t % out_w);
    t /= out_w;
    const int oy = (int)(t % out_h);
    const int m = (int)(t / out_h);
    const float* xp = x + (int64_t)m * in_h * in_w * minor + c;
    float acc = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int s = oy * down_y + ky - pad_y0;
      if (s < 0 || s >= uh || (s % up_y) != 0) continue;
      const int iy = s / up_y;
      for (int kx = 0; kx < kw; ++kx) {
        const int u = ox * down_x + kx - pad_x0;
        if (u < 0 || u >= uw || (u % up_x) != 0) continue;
        const int ix = u / up_x;
        acc = fmaf(ktaps[ky * kw + kx], xp[((int64_t)iy * in_w + ix) * minor], acc);
      }
    }
    if (epi.enabled) acc = epi_apply(epi, acc, m, oy, ox, out_h, out_w);
    out[i] = acc;
  }
}

}  // namespace

template <typename T>
static int upfirdn2d_impl(T* out, const T* x, const float* kernel, int major, int in_h, int in_w, int minor, int kh, int kw,
                          int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                          const vsp_fir_epilogue* epi_in, vsp_stream_t stream) {
  constexpr bool BF = !std::is_same<T, float>::value;
  VSP_REQUIRE(major >= 0 && in_h >= 0 && in_w >= 0 && minor >= 0, "upfirdn2d: negative dimension");
  VSP_REQUIRE(kh >= 1 && kw >= 1 && kh * kw <= 1024, "upfirdn2d: unsupported kernel size %dx%d", kh, kw);
  VSP_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d: up/down factors must be >= 1");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  VSP_REQUIRE(out_h >= 0 && out_w >= 0, "upfirdn2d: negative output size %dx%d", out_h, out_w);
  const int64_t total = (int64_t)major * out_h * out_w * minor;
  if (total == 0) return VSP_OK;
  VSP_REQUIRE(out && x && kernel, "upfirdn2d: null pointer");
  VSP_REQUIRE((int64_t)major * in_h * in_w * minor < (int64_t)1 << 40, "upfirdn2d: tensor too large");

  Epi e{};
  e.channels = 1;
  if (epi_in) {
    VSP_REQUIRE(minor == 1, "upfirdn2d: fused epilogue needs minor == 1");
    VSP_REQUIRE(epi_in->channels >= 1 && major % epi_in->channels == 0,
                "upfirdn2d: epilogue channels=%d does not divide major=%d", epi_in->channels, major);
    VSP_REQUIRE(!epi_in->noise || epi_in->noise_w, "upfirdn2d: noise given without noise_w");
    e.plane_scale = epi_in->plane_scale;
    e.noise = epi_in->noise;
    e.noise_w = epi_in->noise_w;
    e.act_bias = epi_in->act_bias;
    e.res1 = epi_in->res1;
    e.res2 = epi_in->res2;
    e.channels = epi_in->channels;
    e.act = epi_in->act;
    e.slope = epi_in->slope;
    e.gain = epi_in->gain;
    e.separable = (epi_in->flags & VSP_FIR_SEPARABLE) ? 1 : 0;
    // an epilogue that names no operand and no activation is the identity: the kernels then take their plain form
    e.enabled = (e.plane_scale || e.noise || e.act || e.res1 || e.res2) ? 1 : 0;
  }
  hipStream_t s = vsp::as_stream(stream);
  const bool tile_ok = (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && minor == 1 && out_w >= 16);
  // (fp32 planes stay on the tile kernel: the same strip walk measured 4.6 against 4.2 TB/s on 32 channels at 1024^2 but 3.4-4.0 against
  //  4.1-4.6 everywhere else -- 151 registers, three waves per SIMD, twice the bytes per lane in flight; default bench 195.6 against 195.1)
  if constexpr (BF) {
    // the strip walk (no LDS, ~24 vector instructions per output): 4 x 4 taps on planes whose rows are whole 16-byte output segments
    static const int strip_env = vsp::tune_env("VSP_FIR_STRIP") ? atoi(vsp::tune_env("VSP_FIR_STRIP")) : 1;
    const int64_t xb = (int64_t)major * in_h * in_w * 2, ob = (int64_t)major * out_h * out_w * 2;
    // rows that are not whole strips (the down-sampling blurs: 2^n + 1 outputs per row): whole strips + a TAIL launch, plain form only
    const bool ragged = out_w % 8 != 0;
    if (strip_env && tile_ok && kh == 4 && kw == 4 && (!ragged || !e.enabled) && out_h >= 4 && xb < 0x7ffffff0ll && ob < 0x7ffffff0ll &&
        (ragged || (reinterpret_cast<uintptr_t>(out) & 15) == 0) && (!e.res1 || (reinterpret_cast<uintptr_t>(e.res1) & 15) == 0) &&
        (!e.res2 || (reinterpret_cast<uintptr_t>(e.res2) & 15) == 0) && (!e.noise || (reinterpret_cast<uintptr_t>(e.noise) & 15) == 0)) {
      constexpr int RC = VSP_STRIP_RC;
      const int strips_x = out_w / 8, chunks_y = (out_h + RC - 1) / RC;
      const int64_t total_t = (int64_t)(major - 1) * chunks_y * strips_x;
      if (total_t < ((int64_t)1 << 31)) {
        if (total_t > 0) {
          const unsigned gridn = (unsigned)((total_t + 255) / 256);
#define VSP_STRIP_LAUNCH(EN_, ACT_, SEP_)                                                                                                         \
  fir_strip_bf16_kernel<RC, EN_, ACT_, SEP_><<<gridn, 256, 0, s>>>(out, x, kernel, in_h, in_w, out_h, out_w, pad_x0, pad_y0, strips_x, chunks_y, \
                                                                   (int)total_t, (int)xb, (int)ob, 0, e)
          if (e.separable) {
            if (!e.enabled) VSP_STRIP_LAUNCH(false, false, true);
            else if (e.act) VSP_STRIP_LAUNCH(true, true, true);
            else VSP_STRIP_LAUNCH(true, false, true);
          } else {
            if (!e.enabled) VSP_STRIP_LAUNCH(false, false, false);
            else if (e.act) VSP_STRIP_LAUNCH(true, true, false);
            else VSP_STRIP_LAUNCH(true, false, false);
          }
#undef VSP_STRIP_LAUNCH
        }
        if (ragged && major > 1) {
          const int total_tail = (major - 1) * chunks_y;
          const unsigned gridt = (unsigned)((total_tail + 255) / 256);
          if (e.separable)
            fir_strip_bf16_kernel<RC, false, false, true, true><<<gridt, 256, 0, s>>>(out, x, kernel, in_h, in_w, out_h, out_w, pad_x0, pad_y0, 1, chunks_y,
                                                                                     total_tail, (int)xb, (int)ob, strips_x * 8, e);
          else
            fir_strip_bf16_kernel<RC, false, false, false, true><<<gridt, 256, 0, s>>>(out, x, kernel, in_h, in_w, out_h, out_w, pad_x0, pad_y0, 1, chunks_y,
                                                                                      total_tail, (int)xb, (int)ob, strips_x * 8, e);
        }
        // plane 0 (see the kernel's header): the tile kernel on one plane
        const int tiles_x = (out_w + TOW - 1) / TOW, tiles_y = (out_h + TOH - 1) / TOH;
        const int blocks0 = tiles_x * tiles_y;
        fir_tile_kernel<4, 4, T, 1><<<(unsigned)blocks0, 256, 0, s>>>(out, x, kernel, in_h, in_w, out_h, out_w, pad_x0, pad_y0, tiles_x, tiles_y, blocks0, e);
        return vsp::check_launch("upfirdn2d(strip)");
      }
    }
  }
  if (tile_ok && ((kh == 4 && kw == 4) || (kh == 3 && kw == 3) || (kh == 2 && kw == 2))) {
    const int tiles_x = (out_w + TOW - 1) / TOW, tiles_y = (out_h + TOH - 1) / TOH;
    const int64_t blocks = (int64_t)tiles_x * tiles_y * major;
    VSP_REQUIRE(blocks < ((int64_t)1 << 31), "upfirdn2d: grid too large");
    // One tile per block.  Two (both windows requested up front; VSP_FIR_NTB = 2, tuning) measured SLOWER on bf16 planes too (round 4:
    // 32 ch at 1024^2, B = 16: 2.50 -> 2.24 TB/s, C3 402 -> 391 img/s): half the bytes per tile is not what holds the bf16 blur at
    // 2.3-2.9 TB/s.  Nor is the halfword alignment of its 8-byte window loads on the (2H+1)-wide rows: the same quads fetched as aligned
    // dwords + a 16-bit funnel shift measured 2-3 % slower (2.51 -> 2.45 TB/s; tried and removed).  Per element the bf16 blur already runs
    // 1.5 x the fp32 one (0.62 vs 0.40 outputs / ns).  PMC (32 channels at 1024^2, bf16, B = 16): 1.05 M wavefronts, 394 VALU +
    // 228 SALU instructions per wave, 5.6 of the 7 admissible waves resident per SIMD (SQ_WAVE_CYCLES counts quad-cycles).  PERSISTENT blocks (8 per
    // CU walking the tile list with the next window requested one tile ahead, before or after the epilogue operands) measured 25-45 %
    // SLOWER on both element types (fp32 plain 4.08 -> 3.1 TB/s, bf16 2.36 -> 1.65): tried and removed -- two barriers per tile in a
    // resident block cost more than launching a fresh one.  The interior fast path below is what paid (fp32 3.0-3.5 -> 4.1-4.5 TB/s).
    static const int ntb_env = vsp::tune_env("VSP_FIR_NTB") ? atoi(vsp::tune_env("VSP_FIR_NTB")) : 0;
    const int ntb = ntb_env == 2 ? 2 : 1;
#define VSP_FIR_LAUNCH(KH_, NTB_)                                                                                                     \
  fir_tile_kernel<KH_, KH_, T, NTB_><<<(unsigned)((blocks + NTB_ - 1) / NTB_), 256, 0, s>>>(out, x, kernel, in_h, in_w, out_h, out_w, \
                                                                                         pad_x0, pad_y0, tiles_x, tiles_y, (int)blocks, e)
    if (kh == 4) {
      if (ntb == 2) VSP_FIR_LAUNCH(4, 2); else VSP_FIR_LAUNCH(4, 1);
    } else if (kh == 3) {
      VSP_FIR_LAUNCH(3, 1);
    } else {
      VSP_FIR_LAUNCH(2, 1);
    }
#undef VSP_FIR_LAUNCH
    return vsp::check_launch("upfirdn2d(tile)");
  }
  if constexpr (BF) {
    return vsp::fail(VSP_ENOTSUP, "upfirdn2d_bf16: only the blur form (up = down = 1, minor = 1, 2x2 / 3x3 / 4x4 taps, width >= 16) has a bf16 kernel");
  } else {
    int64_t blocks = (total + 255) / 256;
    if (blocks > vsp::kMaxStreamBlocks) blocks = vsp::kMaxStreamBlocks;
    fir_generic_kernel<<<(unsigned)blocks, 256, kh * kw * sizeof(float), s>>>(
        out, x, kernel, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, e);
    return vsp::check_launch("upfirdn2d(generic)");
  }
}

extern "C" int vsp_upfirdn2d_f32(float* out, const float* x, const float* kernel, int major, int in_h, int in_w,
                                  int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, const vsp_fir_epilogue* epi_in,
                                  vsp_stream_t stream) {
  return upfirdn2d_impl<float>(out, x, kernel, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                               pad_y1, epi_in, stream);
}

extern "C" int vsp_upfirdn2d_bf16(uint16_t* out, const uint16_t* x, const float* kernel, int major, int in_h, int in_w,
                                   int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                   int pad_x0, int pad_x1, int pad_y0, int pad_y1, const vsp_fir_epilogue* epi_in,
                                   vsp_stream_t stream) {
  return upfirdn2d_impl<vsp::bf16_t>(out, x, kernel, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                     pad_y0, pad_y1, epi_in, stream);
}

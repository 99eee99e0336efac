// conv2d as an fp32-MFMA implicit GEMM for gfx950 (MI355X), NCHW, with fused prologue/epilogue.
//
// Mapping (per image b, per output-channel group g):
//   GEMM  M = output channels (weights = MFMA A operand)
//         N = output pixels   (input patch = MFMA B operand)
//         K = Cin * KH * KW   (4 consecutive input channels of one tap per v_mfma_f32_16x16x4_f32)
// One workgroup (WM*WN wave64s) owns a CO_T x NPIX output tile of one image:
//   CO_T = 16*MB*WM channels, NPIX = 16*NB*WN pixels arranged as TH rows x TW cols (TW a power of two).
// K is walked in chunks of CK input channels.  Per chunk the block stages into LDS
//   * the weight slab   Wl[tap][ci][co]   (CO_T contiguous, pitch WS == 16 mod 32 -> conflict-free A reads), and
//   * the input patch   P[ci][PH][PW]     (PH = (TH-1)*sy+(KH-1)*d+1 rows incl. halo, zero-filled outside the image,
//                                           already multiplied by the per-(b,ci) style / BN scale; plane pitch PS
//                                           == 16 mod 32 (stride 1) or odd (stride 2) -> conflict-free B reads),
// then every wave runs KH*KW*(CK/4) k-steps of MB*NB MFMAs, reading A/B fragments with ds_read_b32: each staged
// input word feeds KH*KW taps x CO_T channels, each staged weight word feeds NPIX pixels.  fp32 MFMA issues at
// 32 cycles/instruction/SIMD (MI355X_MICROARCH.md), i.e. (MB+NB) LDS reads per MB*NB*32 cycles: the kernel is
// MFMA-bound by construction and staging of chunk i+1 by one resident block overlaps the MFMAs of another
// (>= 2 blocks per CU; LDS per block <= 64 KB).
// Numerics: v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain (no TF32 on gfx950) -> parity with the reference's
// fp32 conv is summation-order noise only.
//
// Reference semantics implemented here are listed on vsp_conv2d_f32 in include/vspbfr_hip.h.
#pragma once
#include "vsp_common.h"

namespace vspconv {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct ConvK {
  const float* x;
  const float* w;
  float* y;
  int B, Cin, H, W, G, cout_g, OH, OW, KH, KW, sy, sx;
  int dil[4], pady[4], padx[4];
  int y_ch, y_coff, y_h, y_w, osy, osx, ooy, oox;
  const float* in_scale;
  int in_scale_bstride;
  const float* in_shift;
  // epilogue operands, resolved on the host: an absent operand points at a device constant (1 or 0) and has
  // stride 0, so the kernel issues the same unconditional loads for every epilogue flavour.
  const float* osp; int oss;   // out_scale  [B, Cout]
  const float* csp; int css;   // ch_scale   [Cout]
  const float* cbp; int cbs;   // ch_bias    [Cout]
  const float* b1p; int b1s;   // bias1      [Cout]
  float s1, g1;
  const float* nzp; int nzs;   // noise      [B, OH, OW]
  const float* nwp;            // noise weight (device scalar; constant 0 when absent)
  const float* b2p; int b2s;   // bias2      [Cout]
  const float* s2p; int s2s;   // negative slope of the second activation: per channel (PReLU) or constant
  float g2;
  const float* r1p; int r1s;   // residuals: [B, res_ch, y_h, y_w]
  const float* r2p; int r2s;
  int res_ch, res_coff;
  // derived on the host
  int tw_log2, th, tiles_x, tiles_y, co_tiles;  // co_tiles = tiles per group
  int w_vec4;                                    // weight rows may be read as float4
  int ps_odd;                                    // plane pitch parity target (stride-2 reads)
  int x_ch, x_gs;                                // input channels per image, input-channel stride between groups
  int strip_col;                                 // transposed mode: edge-strip blocks after the tiles_x*tiles_y main tiles (-1: none)
  int dbg;                                       // ablation switches for kernel tuning (env VSP_CONV_DBG; 0 in production)
  int bf_pitch, bf_plane;                        // conv_bf16.hip: patch row pitch and plane size in positions
  int bf_isc_s, bf_ish_s;                        // conv_bf16.hip: channel stride of in_scale / in_shift (0: absent -> constant)
  int wg_order;                                  // conv_wino.hip: 0 = dispatch order, 1 = pixel-tile-major per XCD, 2 = channel-tile-major per XCD
  // conv_wino.hip: per-input-channel operands resolved to pointer + strides (absent -> a device constant, strides 0), so that the
  // interval body is branch-free: wt* = style scale applied on V (in_scale without in_shift); wc* / wsh* = the affine input of
  // folded BatchNorm layers, applied to in-image pixels when the patch is committed (in_shift given)
  const float* wtp; int wt_cs, wt_bs;
  const float* wcp; int wc_cs, wc_bs;
  const float* wshp; int wsh_cs;
  int io_bf16;                                   // conv_bf16.hip: x, y, res1, res2 are bf16 in HBM
  int64_t w_bs;                                  // conv_bf16.hip: 16-byte units between the per-image weight sets (0: one set); then no input scale
  int bf_tab;                                    // conv_bf16.hip: byte offset of the per-channel operand table in dynamic LDS
  int wg_cgs;                                    // conv_pipe.hip, wg_order 2: channel tiles per group (their weights fit half an XCD's L2)
  int rv_copad, rv_cbase, rv_ctot;               // conv_bf16_rv.hip: channel rows of a weight slab (cout_g rounded up to 32); first channel of the
                                                 // launched dilation group in the layer's operands and in y; channels of the layer (G * cout_g)
};

__device__ __forceinline__ int round_pitch(int n, int odd) {
  // smallest p >= n with p % 32 == 16 (unit-stride B reads) or p odd (stride-2 B reads)
  if (odd) return n | 1;
  int p = (n & ~31) + 16;
  return p >= n ? p : p + 32;
}

template <int MB, int NB, int WM, int WN, int CK, int WK, int PMAX, int PF, int OCC, bool TC = false, bool DG = false>
__global__ __launch_bounds__(64 * WM * WN * WK, OCC) void conv_igemm_kernel(const ConvK p) {
  // TC = true: stride-2 transposed 3x3 convolution (conv_transpose2d, padding 0) in ONE pass.  Output (2m+py, 2n+px) only
  // sees taps with ky = py, kx = px (mod 2), so every tap belongs to exactly one of the four sub-pixel phases: the wave's
  // four N-blocks are the four phases of ONE group of 16 input positions, tap (ky, kx) multiplies the staged input shifted
  // by (-(ky>>1), -(kx>>1)) into phase (ky&1, kx&1).  The staged patch (halo 1 up/left) feeds all 9 taps like a 3x3 conv,
  // instead of four launches that each re-stage the input for 4, 2, 2 and 1 taps.
  // NB = 4 * NP there: the wave owns NP groups of 16 input positions and N-block nb = np * 4 + phase.  Each weight tap
  // serves one phase only, so per MFMA a transposed block stages 4x the weights of a plain conv with the same pixel
  // count: NP > 1 (more positions per block) is what buys that back.
  static_assert(!TC || (NB % 4 == 0 && WK == 1), "transposed mode: N-blocks come in groups of four sub-pixel phases");
  constexpr int NP = TC ? NB / 4 : NB;  // 16-wide groups of patch positions per wave
  // DG = true: the four dilated branches of a SMART / LargeConv layer (G = 4 groups over the SAME input, dilation = padding
  // = 1, 2, 4, 8) fused in one block: the block's four 16-channel M-blocks are the four GROUPS (16 channels of each), all
  // fed from ONE staged patch with the largest halo.  Per group the separate launch geometry stages an 18^2 ... 32^2
  // patch for 256 pixels and reuses a weight slab for 16/32 channels only; fused, one 32^2 patch serves 64 channels.
  static_assert(!DG || (MB == 4 && WM == 1 && WK == 1 && !TC), "dilation-group mode: M-block = group");
  // PF = 1: one-chunk register prefetch (ILP hides global latency, ~250 VGPRs, 2 waves/SIMD);
  // PF = 0: loads are consumed in the staging phase itself, the register budget (OCC = min waves/SIMD) buys occupancy and
  //         other blocks' MFMAs hide the latency (TLP).
  // WK > 1: the block's waves are additionally split along K -- wave slice wk runs k-steps wk, wk+WK, ... of every chunk
  // on the SAME output tile and the partial accumulators are summed through LDS before the epilogue.  With CK = 32 this
  // cuts the serial chunk count of deep-K / tiny-map layers (4x4 ... 16x16 maps, 512 channels) by 4 while small
  // 16/32-channel tiles keep >= 256 blocks in flight.
  constexpr int NT = 64 * WM * WN * WK;
  constexpr int CO_T = 16 * MB * WM;
  constexpr int WS = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WM * WN);
  const int wmn = wave - wk * (WM * WN);
  const int wm = wmn / WN, wn = wmn % WN;
  const int lr = lane & 15;  // MFMA row (A) / column (B, D) index inside a 16x16 block
  const int kq = lane >> 4;  // MFMA k slot (A, B); D row group

  // XCD-aware work order (round 2).  In dispatch order a workgroup's neighbours are other PIXEL tiles of the same channel
  // tile, the dispatcher deals them round-robin over the 8 XCDs, and the channel tiles that re-read a patch come a whole grid
  // row later: no L2 ever sees a patch twice (the e4e style-head stem, 44 channel tiles x 64 pixel tiles, fetched 6.8 GB for a
  // 67 MB input).  Bijective remap: every XCD walks a contiguous range of (image, pixel tile, channel tile) with the channel
  // tiles of one pixel tile adjacent -- they read the same patch back to back out of one L2.
  int tile = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.wg_order == 1) {
    const int GX = gridDim.x, GY = gridDim.y, GN = GX * GY, GT = GN * gridDim.z;
    const int wgid = blockIdx.x + GX * (blockIdx.y + GY * blockIdx.z);
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    bz = lid / GN;
    const int lrem = lid - bz * GN;
    tile = lrem / GY;
    by = lrem - tile * GY;
  }
  const int tx_i = tile % p.tiles_x, ty_i = tile / p.tiles_x;
  const int g = DG ? 0 : by / p.co_tiles;
  const int co0 = DG ? by * 16 : (by % p.co_tiles) * CO_T;  // within the group
  const int b = bz;

  // Transposed mode, H and W multiples of the tile: the main tiles cover positions [0,H) x [0,W) exactly and the odd
  // edge (column n = W, row m = H of the (H+1) x (W+1) position grid) is served by strip blocks -- 1 x NPIX column
  // strips, then NPIX x 1 row strips -- instead of a ragged last tile row and column that are 1/TH and 1/TW full.
  int twl = p.tw_log2, TH = p.th, oy0 = ty_i * p.th, ox0 = tx_i << p.tw_log2;
  int mlim = p.H + 1;  // transposed: first invalid position row of this block
  if (TC && p.strip_col >= 0) {
    constexpr int NPIXB = 16 * WN * (NB / 4);
    mlim = p.H;
    int j = tile - p.tiles_x * p.tiles_y;
    if (j >= p.strip_col) {  // row strip (includes the corner)
      j -= p.strip_col;
      twl = __builtin_ctz(NPIXB); TH = 1; oy0 = p.H; ox0 = j * NPIXB; mlim = p.H + 1;
    } else if (j >= 0) {     // column strip
      twl = 0; TH = NPIXB; oy0 = j * NPIXB; ox0 = p.W;
    }
  }
  const int TW = 1 << twl;
  const int gi = p.G > 4 ? 0 : g;  // more than 4 groups = true grouped conv with uniform geometry
  const int D = DG ? max(max(p.dil[0], p.dil[1]), max(p.dil[2], p.dil[3])) : p.dil[gi];
  const int T = p.KH * p.KW;
  const int PH = TC ? TH + 1 : (TH - 1) * p.sy + (p.KH - 1) * D + 1;
  const int PW = TC ? TW + 1 : (TW - 1) * p.sx + (p.KW - 1) * D + 1;
  const int PS = round_pitch(PH * PW, p.ps_odd);
  const int iy0 = TC ? oy0 - 1 : DG ? oy0 - D : oy0 * p.sy - p.pady[gi];
  const int ix0 = TC ? ox0 - 1 : DG ? ox0 - D : ox0 * p.sx - p.padx[gi];

  float* Wl = smem;                // [T][CK][WS]
  float* Pl = smem + T * CK * WS;  // [CK][PS]   (PF = 2: two of them, Pl and Pl + CK * PS, used alternately)

  // per-lane patch offsets of the NB pixel blocks this wave owns
  int pixoff[NP];
#pragma unroll
  for (int nb = 0; nb < NP; ++nb) {
    const int n = (wn * NP + nb) * 16 + lr;
    const int py = n >> twl, px = n & (TW - 1);
    pixoff[nb] = py * p.sy * PW + px * p.sx + (kq + 4 * wk) * PS;
  }
  const int a_lane = (kq + 4 * wk) * WS + wm * MB * 16 + lr;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* xb = p.x + ((int64_t)b * p.x_ch + (int64_t)g * p.x_gs) * p.H * p.W;
  const float* wg = p.w + (int64_t)g * T * p.Cin * p.cout_g;
  // buffer resources: a load is (resource, wave-uniform scalar byte offset, 32-bit lane byte offset) -- with flat pointers every
  // staged word carries a 64-bit lane address (a VALU add per load and two VGPRs per hoisted offset)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7fffffff, 0x00020000);
  const int plane = PH * PW;
  // exact floor(idx / PW) for idx < 2^16, PW <= 2^8 (host guarantees both)
  const unsigned pw_magic = (unsigned)(((1ull << 32) + PW - 1) / PW);

  // ---- staging with a one-chunk register prefetch: the global loads of chunk i+1 are issued right before the MFMA
  // phase of chunk i and only waited for when they are written to LDS, so HBM/L2 latency hides under ~9k cycles of MFMA
  // instead of stalling the block 4-6 dependent round trips per chunk.  Everything that does not depend on the chunk
  // (patch element -> image offset and in-image flag, weight element -> offset) is computed ONCE per lane; per chunk a
  // staged word costs one load (uniform base + lane offset), one fma/select and one ds_write.
  constexpr int NW = WM * WN * WK;
  constexpr int PCH = (CK + NW - 1) / NW;        // patch channels per wave per chunk
  // PMAX (template): prefetched patch words per lane per channel; the rest of a large plane takes the direct path
  constexpr int V = CO_T / 4;
  constexpr int WMAX = (9 * CK * V + NT - 1) / NT;  // prefetched weight float4 per thread (covers 3x3 taps)
  float4 wreg[WMAX];
  float preg[PF == 2 ? 1 : PCH][PF == 2 ? 1 : PMAX];
  float psc[PCH], psh[PCH];
  const int wtotal = T * CK * V;
  const int chw = p.H * p.W;

  auto patch_src = [&](int i, int& off) -> bool {  // element i of the patch plane -> offset inside the channel image
    const int r = (int)__umulhi((unsigned)i, pw_magic);
    const int c = i - r * PW;
    const int iy = iy0 + r, ix = ix0 + c;
    off = iy * p.W + ix;
    return iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
  };

  int poff[PMAX];        // image offset of patch word (lane + 64 e); 0 when outside (the word is zeroed at commit)
  unsigned pin = 0;      // bit e: word e lies inside the image
#pragma unroll
  for (int e = 0; e < PMAX; ++e) {
    const int i = lane + 64 * e;
    int off;
    const bool in = (i < plane) && patch_src(i, off);
    poff[e] = in ? off * 4 : 0;   // byte offset inside the channel plane
    pin |= in ? (1u << e) : 0u;
  }
  int woff[WMAX];        // weight word offset relative to the chunk's first input channel; -1: zero fill
  int wdst[WMAX];        // LDS destination (float index), -1: nothing to write
#pragma unroll
  for (int w = 0; w < WMAX; ++w) {
    const int i = tid + w * NT;
    const int row = i / V, c4 = i - row * V;
    const int tap = row / CK, cl = row - tap * CK;
    const bool ok = i < wtotal;
    wdst[w] = ok ? row * WS + c4 * 4 : -1;
    if constexpr (DG) {  // slab column block c4 >> 2 = group, 16 channels co0.. of each
      const int cc = co0 + (c4 & 3) * 4;
      woff[w] = (ok && cc < p.cout_g) ? ((c4 >> 2) * T * p.Cin + tap * p.Cin + cl) * p.cout_g + cc : -1;
    } else {
      woff[w] = (ok && co0 + c4 * 4 < p.cout_g) ? (tap * p.Cin + cl) * p.cout_g + co0 + c4 * 4 : -1;
    }
  }

  auto issue = [&](int ci0) {
    if constexpr (PF == 2) return;
    if (p.w_vec4) {
      const int wsoff = ci0 * p.cout_g * 4;
#pragma unroll
      for (int w = 0; w < WMAX; ++w) {
        const int row = (tid + w * NT) / V;
        const int cl = row & (CK - 1);
        const bool ok = woff[w] >= 0 && ci0 + cl < p.Cin;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, ok ? woff[w] * 4 : 0, wsoff, 0));
        wreg[w] = ok ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int pc = 0; pc < PCH; ++pc) {
      const int cl = wave + pc * NW;
      const int ci = ci0 + cl;
      const bool chok = cl < CK && ci < p.Cin;  // wave-uniform
      const int cic = chok ? ci : 0;
      psc[pc] = chok ? (p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_bstride + g * p.x_gs + cic] : 1.f) : 0.f;
      psh[pc] = chok ? (p.in_shift ? p.in_shift[cic] : 0.f) : 0.f;
      const int xsoff = cic * chw * 4;
#pragma unroll
      for (int e = 0; e < PMAX; ++e) preg[pc][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, poff[e], xsoff, 0));
    }
  };

  auto commit = [&](int ci0) {
    if constexpr (PF == 2) return;
    if (p.w_vec4) {
#pragma unroll
      for (int w = 0; w < WMAX; ++w)
        if (wdst[w] >= 0) *reinterpret_cast<float4*>(Wl + wdst[w]) = wreg[w];
      for (int i = tid + WMAX * NT; i < wtotal; i += NT) {  // only when KH*KW > 9
        const int row = i / V, c4 = i - row * V;
        const int tap = row / CK, cl = row - tap * CK;
        const int ci = ci0 + cl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci < p.Cin && co0 + c4 * 4 < p.cout_g)
          v = *reinterpret_cast<const float4*>(wg + ((int64_t)tap * p.Cin + ci) * p.cout_g + co0 + c4 * 4);
        *reinterpret_cast<float4*>(Wl + row * WS + c4 * 4) = v;
      }
    } else {
#pragma unroll 4
      for (int i = tid; i < T * CK * CO_T; i += NT) {
        const int row = i / CO_T, c = i - row * CO_T;
        const int tap = row / CK, cl = row - tap * CK;
        const int ci = ci0 + cl;
        float v = 0.f;
        if (ci < p.Cin && co0 + c < p.cout_g) v = wg[((int64_t)tap * p.Cin + ci) * p.cout_g + co0 + c];
        Wl[row * WS + c] = v;
      }
    }
#pragma unroll
    for (int pc = 0; pc < PCH; ++pc) {
      const int cl = wave + pc * NW;
      if (cl >= CK) continue;
      const int ci = ci0 + cl;
      float* dst = Pl + cl * PS;
#pragma unroll
      for (int e = 0; e < PMAX; ++e) {
        const int i = lane + 64 * e;
        if (i < plane) dst[i] = ((pin >> e) & 1u) ? fmaf(preg[pc][e], psc[pc], psh[pc]) : 0.f;
      }
      if (plane > 64 * PMAX) {  // large halos (dilation 4/8, stride 2): the tail of the plane is loaded directly
        // Uniform trip count, clamped offsets, the in-image test as a select AFTER the load: with the load inside a divergent test
        // (and a per-lane loop bound) the compiler waited for each of these loads before issuing the next one.
        const bool chok = ci < p.Cin;
        const float* xc = xb + (int64_t)(chok ? ci : 0) * chw;
#pragma unroll 4
        for (int i0 = 64 * PMAX; i0 < plane; i0 += 64) {
          const int i = i0 + lane;
          int off;
          const bool in = patch_src(i < plane ? i : plane - 1, off) && chok && i < plane;
          const float xv = xc[in ? off : 0];
          if (i < plane) dst[i] = in ? fmaf(xv, psc[pc], psh[pc]) : 0.f;
        }
      }
    }
  };

  // PF = 2: asynchronous staging.  The input patch of chunk i+1 goes global -> LDS directly (global_load_lds: the wave's 64
  // lanes land on 64 consecutive LDS words, which is exactly the plane layout; lanes outside the image are masked off and
  // their words keep the zero written once at kernel start), into the second patch buffer, while chunk i is multiplied.
  // No patch registers, no fma/select/ds_write per staged word; the style scale moves onto the weight rows (w * s[b, ci]),
  // which still pass through registers (the slab rows are padded, an LDS-DMA wave writes contiguous words only).
  auto issue_async = [&](int ci0, float* Pdst) {
    const float* wc = wg + (int64_t)ci0 * p.cout_g;
#pragma unroll
    for (int w = 0; w < WMAX; ++w) {
      const int row = (tid + w * NT) / V;
      const int cl = row & (CK - 1);
      const bool ok = woff[w] >= 0 && ci0 + cl < p.Cin;
      const float4 v = *reinterpret_cast<const float4*>(wc + (ok ? woff[w] : 0));
      const float sc = (ok && p.in_scale) ? p.in_scale[(int64_t)b * p.in_scale_bstride + g * p.x_gs + ci0 + cl] : 1.f;
      wreg[w] = ok ? make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int pc = 0; pc < PCH; ++pc) {
      const int cl = wave + pc * NW;
      const int ci = ci0 + cl;
      if (cl >= CK || ci >= p.Cin) continue;  // wave-uniform; a stale plane meets zero weight rows
      const float* xc = xb + (int64_t)ci * chw;
      float* dstc = Pdst + cl * PS;
#pragma unroll
      for (int e = 0; e < PMAX; ++e)
        if ((pin >> e) & 1u) __builtin_amdgcn_global_load_lds(xc + (poff[e] >> 2), dstc + 64 * e, 4, 0, 0);
      if (plane > 64 * PMAX) {
        for (int i0 = 64 * PMAX; i0 < plane; i0 += 64) {  // i0 is wave-uniform: the LDS base of the instruction
          int off;
          if (i0 + lane < plane && patch_src(i0 + lane, off)) __builtin_amdgcn_global_load_lds(xc + off, dstc + i0, 4, 0, 0);
        }
      }
    }
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int w = 0; w < WMAX; ++w)
      if (wdst[w] >= 0) *reinterpret_cast<float4*>(Wl + wdst[w]) = wreg[w];
  };
  if constexpr (PF == 2) {
    for (int i = tid; i < 2 * CK * PS; i += NT) Pl[i] = 0.f;
    __syncthreads();
    issue_async(0, Pl);
  } else if (PF) {
    issue(0);
  }
  int pbuf = 0;
  for (int ci0 = 0; ci0 < p.Cin; ci0 += CK) {
    __syncthreads();  // previous chunk's fragment reads are done (PF = 2: and this wave's LDS-DMA of chunk ci0 has landed)
    if constexpr (PF == 2) {
      commit_w();
    } else if (!(p.dbg & 1) || ci0 == 0) {
      if (!PF) issue(ci0);
      commit(ci0);
    }
    __syncthreads();
    if constexpr (PF == 2) {
      Pl = smem + T * CK * WS + pbuf * CK * PS;
      pbuf ^= 1;
      if (ci0 + CK < p.Cin) issue_async(ci0 + CK, smem + T * CK * WS + pbuf * CK * PS);
    } else {
      if (PF && ci0 + CK < p.Cin && !(p.dbg & 1)) issue(ci0 + CK);
    }
    if (p.dbg & 2) continue;
    // ---- MFMA over taps x (CK/4) k-steps
    if constexpr (TC) {
      // Each tap feeds ONE phase, so a tap is only MB*NP MFMAs per k-step against MB+NP fragment reads: without help the
      // compiler waits for an LDS round trip in front of every pair of MFMAs.  Software pipeline by hand: the fragments
      // of tap t+1 are requested before the MFMAs of tap t are issued (two register sets, all indices compile-time).
      constexpr int KS = CK / 4;
      float af[2][KS][MB], bf[2][KS][NP];
      auto load_tap = [&](int tap, float (&a)[KS][MB], float (&bq)[KS][NP]) {
        const int ky = tap / 3, kx = tap % 3;
        const int boff = (1 - (ky >> 1)) * PW + (1 - (kx >> 1));
        const float* wt = Wl + tap * CK * WS + a_lane;
#pragma unroll
        for (int c4 = 0; c4 < KS; ++c4) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) a[c4][mb] = wt[c4 * 4 * WS + mb * 16];
#pragma unroll
          for (int np = 0; np < NP; ++np) bq[c4][np] = Pl[c4 * 4 * PS + pixoff[np] + boff];
        }
      };
      load_tap(0, af[0], bf[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1;
        if (tap < 8) load_tap(tap + 1, af[cur ^ 1], bf[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int ph = ((tap / 3) & 1) * 2 + ((tap % 3) & 1);
#pragma unroll
        for (int c4 = 0; c4 < KS; ++c4)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int np = 0; np < NP; ++np)
              acc[mb][np * 4 + ph] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][c4][mb], bf[cur][c4][np], acc[mb][np * 4 + ph], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (DG) {
      int gbase[4];  // centre offset of group g's taps inside the shared patch
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) gbase[gg] = (D - p.dil[gg]) * (PW + 1);
      for (int ky = 0; ky < 3; ++ky) {
        for (int kx = 0; kx < 3; ++kx) {
          const float* wt = Wl + (ky * 3 + kx) * CK * WS + a_lane;
          int boff[4];
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) boff[gg] = gbase[gg] + (ky * PW + kx) * p.dil[gg];
#pragma unroll
          for (int c4 = 0; c4 < CK / 4; ++c4) {
            float a[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) a[gg] = wt[c4 * 4 * WS + gg * 16];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
              float bv[NB];
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) bv[nb] = Pl[c4 * 4 * PS + pixoff[nb] + boff[gg]];
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[gg][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gg], bv[nb], acc[gg][nb], 0, 0, 0);
            }
          }
        }
      }
    } else {
      for (int ky = 0; ky < p.KH; ++ky) {
        for (int kx = 0; kx < p.KW; ++kx) {
          const int boff = ky * D * PW + kx * D;
          const float* wt = Wl + (ky * p.KW + kx) * CK * WS + a_lane;
#pragma unroll
          for (int c4 = 0; c4 < CK / 4; c4 += WK) {  // this wave's k-steps: c4 + wk (folded into a_lane / pixoff)
            float a[MB], bv[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = wt[c4 * 4 * WS + mb * 16];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = Pl[c4 * 4 * PS + pixoff[nb] + boff];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], bv[nb], acc[mb][nb], 0, 0, 0);
          }
        }
      }
    }
  }

  if (WK > 1) {  // sum the K slices: slices 1..WK-1 park their accumulators in LDS, slice 0 adds them and finishes
    __syncthreads();
    float* red = smem;  // [(WK-1)][WM*WN][MB*NB*4][64]
    if (wk > 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            red[((((wk - 1) * (WM * WN) + wmn) * (MB * NB * 4)) + (mb * NB + nb) * 4 + r) * 64 + lane] = acc[mb][nb][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < WK; ++k2)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[mb][nb][r] += red[((((k2 - 1) * (WM * WN) + wmn) * (MB * NB * 4)) + (mb * NB + nb) * 4 + r) * 64 + lane];
  }

  if (p.dbg & 4) return;
  // ---- epilogue: lane holds pixel column lr of block nb, channel rows kq*4 + r of block mb.
  // Branch-free: absent operands read a constant (1 or 0) through a zero stride so that every load of a channel
  // group is issued back to back (a null-pointer branch per element serialises ~10 dependent loads per output).
  const int Cout = p.G * p.cout_g;
  const float* osp = p.osp + (int64_t)b * Cout * p.oss;
  const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW * p.nzs;
  const float nw = p.nwp[0];
  const float s1 = p.s1, g1 = p.g1, g2 = p.g2;
  const int oss = p.oss, css = p.css, cbs = p.cbs, b1s = p.b1s, b2s = p.b2s, s2s = p.s2s, nzs = p.nzs;

  // 32-bit offsets inside one image (host checks C*H*W < 2^31); 64-bit only for the per-image bases
  float* yb = p.y + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const float* r1b = p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r1s;
  const float* r2b = p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r2s;
  const int r1s = p.r1s, r2s = p.r2s;
  const int y_plane = p.y_h * p.y_w;

  if constexpr (TC) {
    // position (m, n) of group np feeds outputs (2m+py, 2n+px); the two px phases of a lane are neighbours in memory and
    // leave as one 8-byte store (rows of 2W+1 floats are only 4-byte aligned: global_store_dwordx2 takes that)
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
    int yoff[NP][2];     // offset of (2m+py, 2n) inside the plane, < 0: row outside
    bool pair[NP];       // column 2n+1 exists
#pragma unroll
    for (int np = 0; np < NP; ++np) {
      const int n = (wn * NP + np) * 16 + lr;
      const int m = oy0 + (n >> twl), c = ox0 + (n & (TW - 1));
      const bool cok = c <= p.W;
      pair[np] = c < p.W;
#pragma unroll
      for (int py = 0; py < 2; ++py)
        yoff[np][py] = (cok && m < mlim && m < p.H + 1 - py) ? (2 * m + py) * p.y_w + 2 * c : -1;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cg = co0 + (wm * MB + mb) * 16 + kq * 4 + r;
        const bool cok = cg < p.cout_g;
        const int co = cok ? cg : 0;
        const float os = osp[co * oss];
        const float cs = p.csp[co * css];
        const float cb = p.cbp[co * cbs];
        const float b1 = p.b1p[co * b1s];
        const float b2 = p.b2p[co * b2s];
        const float sl2 = p.s2p[co * s2s];
        float* yc = yb + (int64_t)co * y_plane;
        auto fin = [&](float v) {
          v = v * os * cs + cb + b1;
          v = (v > 0.f ? v : v * s1) * g1 + b2;
          return (v > 0.f ? v : v * sl2) * g2;
        };
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            if (yoff[np][py] < 0 || !cok) continue;
            const float v0 = fin(acc[mb][np * 4 + py * 2][r]), v1 = fin(acc[mb][np * 4 + py * 2 + 1][r]);
            if (pair[np])
              *reinterpret_cast<f32x2u*>(yc + yoff[np][py]) = f32x2u{v0, v1};
            else
              yc[yoff[np][py]] = v0;
          }
      }
    }
    return;
  }
  int yoff[NB];  // < 0: pixel outside the image
  float nz[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (wn * NB + nb) * 16 + lr;
    const int oy = oy0 + (n >> twl), ox = ox0 + (n & (TW - 1));
    const bool ok = (oy < p.OH && ox < p.OW);
    const int oyc = ok ? oy : 0, oxc = ok ? ox : 0;
    yoff[nb] = ok ? (oyc * p.osy + p.ooy) * p.y_w + oxc * p.osx + p.oox : -1;
    nz[nb] = nzp[(oyc * p.OW + oxc) * nzs] * nw;
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // one channel at a time: 6 parameter registers live instead of 24 (occupancy budget)
      const int cg = DG ? co0 + kq * 4 + r : co0 + (wm * MB + mb) * 16 + kq * 4 + r;  // channel within the group
      const bool cok = cg < p.cout_g;
      const int co = (DG ? mb : g) * p.cout_g + (cok ? cg : 0);
      const float os = osp[co * oss];
      const float cs = p.csp[co * css];
      const float cb = p.cbp[co * cbs];
      const float b1 = p.b1p[co * b1s];
      const float b2 = p.b2p[co * b2s];
      const float sl2 = p.s2p[co * s2s];
      const int cbase = co * y_plane;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int ro = cbase + (yoff[nb] < 0 ? 0 : yoff[nb]);
        const float r1v = r1b[ro * r1s];
        const float r2v = r2b[ro * r2s];
        float v = acc[mb][nb][r] * os;
        v = v * cs + cb;
        v += b1;
        v = (v > 0.f ? v : v * s1) * g1;
        v += nz[nb];
        v += b2;
        v = (v > 0.f ? v : v * sl2) * g2;
        v += r1v;
        v += r2v;
        if (yoff[nb] >= 0 && cok) yb[ro] = v;
      }
    }
  }
}


// conv_wino.hip: Winograd F(2x2,3x3) launch (q.w = transformed weights [16][Cin][Cout]); tiles are set inside
int wino_launch(ConvK q, int form, hipStream_t stream);
// conv_smallmap.hip: K-split GEMM form for layers with few output positions (q filled by fill_convk: no tile plan needed)
bool smallmap_eligible(const ConvK& q, bool transposed);
int smallmap_launch(const ConvK& q, hipStream_t stream);
int wino_ro_launch(ConvK q, int mbw, int ivc, hipStream_t stream);
bool wino_ro_eligible(const ConvK& q);
int wino_rod_launch(ConvK q, int mbw, hipStream_t stream);            // conv_wino_rod.hip: row-owner form, dilation groups
bool wino_rod_eligible(const ConvK& q);
int wino_rs_launch(ConvK q, hipStream_t stream);                      // conv_wino_rs.hip: register-resident U, polyphase staging (Cin <= 64)
bool wino_rs_eligible(const ConvK& q);
bool wino_rs_profitable(const ConvK& q);   // conv_wino_ro.hip: row-owner form, undilated groups
// conv_wino4.hip: Winograd F(4x4,3x3) as input transform + barrier-free GEMM (deep layers)
bool wino4_eligible(const ConvK& q);
size_t wino4_weight_floats(int cin, int cout);
size_t wino4_work_floats(int B, int cin, int H, int W);
int wino4_weight_launch(float* U, const float* wp, int cin, int cout, hipStream_t stream);
int wino4_launch(ConvK q, float* V, const float* scale, int scale_bs, hipStream_t stream);
// conv_wino4f.hip: Winograd F(4x4,3x3) fused in registers (shallow wide layers)
bool wino4f_eligible(const ConvK& q);
size_t wino4f_weight_floats(int cin, int cout);
int wino4f_weight_launch(float* U, const float* wp, int cin, int cout, hipStream_t stream);
int wino4f_launch(ConvK q, hipStream_t stream);
int wino_chunk();           // input channels per chunk the transformed-weight layout is built for
int wino_mbw(int cout_g);   // 16-channel blocks per workgroup (fragment layout) for a layer with cout_g channels per group

// conv_bf16.hip: 3x3 stride-1 convolution on the bf16 matrix pipe (q.w = bf16 weights in LDS-image order)
int bf16_launch(const ConvK& q, int mode, int variant, hipStream_t stream);
int bf16_launch_split(const ConvK& q, int mode, int variant, hipStream_t stream);  // hi + lo bf16 operands ("bf16x3")
// conv_bf16_rv.hip: the row-vector-K form for low-channel large-map stride-1 layers with bf16 activations
bool bf16rv_eligible(const ConvK& q);
int bf16rv_launch(const ConvK& q, int variant, hipStream_t stream);
bool bf16dg_eligible(const ConvK& q);                                 // conv_bf16_dg.hip: dilation groups of <= 16 channels, bf16 activations, Cin <= 64
int bf16dg_launch(const ConvK& q, bool full, hipStream_t stream);

struct Cfg {
  int MB, NB, WM, WN, CK, WK, PMAX, PF, OCC;  // a name ending in "t" marks a transposed-conv kernel
  const char* name;
  void (*kern)(const ConvK);
};

// name = MBxNBxWMxWNxCK [k<WK>] p<PF>o<OCC>
#define VSP_CFG(MB, NB, WM, WN, CK, WK, PMAX, PF, OCC)                                                     \
  {                                                                                                        \
    MB, NB, WM, WN, CK, WK, PMAX, PF, OCC, #MB "x" #NB "x" #WM "x" #WN "x" #CK "k" #WK "p" #PF "o" #OCC,  \
        conv_igemm_kernel<MB, NB, WM, WN, CK, WK, PMAX, PF, OCC>                                           \
  }

// transposed (stride-2, 3x3) variants: name suffix "t"
// same as VSP_CFG with the patch-prefetch depth in the name ("m16": 16 hoisted patch words per lane and channel)
#define VSP_CFGM(MB, NB, WM, WN, CK, WK, PMAX, PF, OCC)                                                          \
  {                                                                                                              \
    MB, NB, WM, WN, CK, WK, PMAX, PF, OCC, #MB "x" #NB "x" #WM "x" #WN "x" #CK "k" #WK "p" #PF "o" #OCC "m" #PMAX, \
        conv_igemm_kernel<MB, NB, WM, WN, CK, WK, PMAX, PF, OCC>                                                 \
  }

#define VSP_CFGD(NB, WN, CK, PMAX, PF, OCC)                                                                \
  {                                                                                                        \
    4, NB, 1, WN, CK, 1, PMAX, PF, OCC, "4x" #NB "x1x" #WN "x" #CK "k1p" #PF "o" #OCC "d",                  \
        conv_igemm_kernel<4, NB, 1, WN, CK, 1, PMAX, PF, OCC, false, true>                                 \
  }

#define VSP_CFGT(MB, NB, WM, WN, CK, PMAX, PF, OCC)                                                        \
  {                                                                                                        \
    MB, NB, WM, WN, CK, 1, PMAX, PF, OCC, #MB "x" #NB "x" #WM "x" #WN "x" #CK "k1p" #PF "o" #OCC "t",       \
        conv_igemm_kernel<MB, NB, WM, WN, CK, 1, PMAX, PF, OCC, true>                                      \
  }

}  // namespace vspconv

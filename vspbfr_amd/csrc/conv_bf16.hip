// 3x3 stride-1 convolution on the bf16 matrix pipe of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the "bf16 kernels"
// configuration of the path (BASELINE configs[2]).  Activations stay fp32 NCHW in HBM -- the kernel converts while it stages
// -- so it drops into the same call sites as vsp_conv2d_f32 with the same fused prologue / epilogue (ConvK).
//
// The bf16 pipe is 16x the fp32 MFMA rate, so the layer is no longer bound by the matrix pipe but by what feeds it:
//   * implicit GEMM, M = output channels (weights = A operand), N = 32-pixel row segments (patch = B operand),
//     K = 16 input channels of one tap per MFMA.  A lane's fragment is 8 consecutive channels of one row/pixel = 16 bytes,
//     so both LDS images are kept in CHANNEL-OCTET planes:  P[octet][patch position][8 ch], W[tap][octet][co][8 ci].
//     A fragment read is one ds_read_b128; the 16 lanes of an LDS access group always hold 16 consecutive positions
//     (mod 16) of ONE plane -> conflict-free without padding or swizzles, for every tap offset.
//   * weights are packed on the host in exactly that order (vsp_conv2d_bf16 documents it) and go global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction): no registers, no ds_write.
//   * the input patch is loaded one chunk ahead into registers (8 channels x PT positions per lane; the channel is
//     wave-uniform, lanes run along the row -> coalesced), scaled by the per-sample style in fp32, converted with
//     v_cvt_pk_bf16_f32 and written as one ds_write_b128 per position after the MFMAs of the current chunk.
//   * weights and patch are double-buffered: ONE barrier per 16-channel chunk.
//   * dilation d is polyphase in ROWS only: a workgroup owns rows ry, ry + d, ry + 2d, ... (consecutive rows of the patch in
//     LDS, so the vertical taps are one patch row apart) and a DENSE run of 32 columns with a halo of d columns, the
//     horizontal taps being d positions apart in LDS (fragment reads are conflict-free for any offset).  Global loads and
//     stores therefore stay whole 128-byte row segments for every dilation (a column-polyphase form reads and writes 4 bytes
//     per 32-byte sector at d = 8: measured 6.1 GB fetched for a 0.7 GB input on the 64 -> 4x16 branch at 512^2).  The (up to
//     four) dilation groups of a SMART branch launch differ only in d and their weight / channel base.
// Modes (template MODE): 0 = stride-1 convolution as above; 1 = stride-2 convolution: the patch is staged as four PARITY
// planes (row parity x column parity), tap (ky, kx) reads plane (ky & 1, kx & 1) at unit stride, so the fragment reads stay
// conflict-free and the global loads stay coalesced (lanes run along the input row and scatter into two planes);
// 2 = stride-2 TRANSPOSED convolution in one pass, as in conv_kernel.h: a 32-position block keeps four sub-pixel phase
// accumulators, tap (ky, kx) multiplies the input shifted by (-(ky >> 1), -(kx >> 1)) into phase (ky & 1, kx & 1); the two
// column phases of a position leave as one 8-byte store.
// Numerics: operands rounded to bf16 (RNE, 8 significant bits), products exact, accumulation fp32; the epilogue chain is
// the fp32 one of the direct kernel.  Not a parity path: tests bound its error against the fp32 kernels.
#include "conv_kernel.h"
#include "vsp_bf16.h"
#include <type_traits>

namespace vspconv {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BCK = 16;     // input channels per chunk = one MFMA k-step
constexpr int BNT = 256;    // threads per workgroup (4 waves)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}

enum { M_CONV = 0, M_S2 = 1, M_TC = 2 };

// DEEP (PT <= 3, 64-channel tiles and up): two patch register sets, loads two chunks ahead, conversion spread over the tap loop.  Larger planes
// (narrow tiles, 5-7 positions per lane) keep ONE set: loads one chunk ahead, conversion after the taps.
// SPLIT ("bf16x3"): both operands are carried as hi + lo bf16 pairs (hi = bf16(v), lo = bf16(v - hi): 16 significant bits) and a
// product is a_hi b_hi + a_hi b_lo + a_lo b_hi on the same fp32 accumulator -- fp32-grade results (relative error ~2^-16 per
// product instead of 2^-8) at three MFMAs per tile pair; the LDS images and the weight DMA double ([part][...]).
// IOB: the activations are bf16 IN HBM (x, y and the two residuals; 2 B per element instead of 4: the bf16-activation
// configuration, hip_ops.ACT_BF16): the patch is fetched as 16-bit loads (same instruction count, half the bytes: the 512^2 and
// 256^2 layers are bound by what the fabric delivers), the epilogue reads / writes four bf16 per 8-byte access.
// PAIR (bf16 activations, even W): a staging task is TWO neighbouring pixels (an aligned 4-byte load) instead of one 2-byte load.  A wave
// load instruction costs the CU's vector-memory path 32 cycles whatever its width (tools/ubench/mfma_valu_gap.hip): with one bf16 element per
// lane the staging of the whole chip is capped at 256 CUs x 4 B/clk = 2.1 TB/s -- which is what every layer of this kernel ran at (512 -> 512 at
// 64^2, B = 16: 700 MB staged in 343 us) -- with the matrix pipe a third busy.
// NOSC (round 6; bf16 activations, pair staging): the style lives in PER-IMAGE weights (p.w_bs; vsp_modulate_weight_bf16: bf16(W * style[b]), the
// reference's own fused form), so a staged pixel is copied, not computed: the eight 4-byte loads of a pair task are interleaved into the two
// pixels' channel octets by eight v_perm_b32 -- no widening, no multiply-add, no rounding, no scale loads (4.5 instead of ~28 vector
// instructions per staged octet; PMC round 5: 4.9 vector + 3.7 scalar instructions per MFMA on 512 -> 512 at 64^2, the pipe 40 % busy).
// Pixels outside the image come back as zeros from the buffer range check (the descriptor covers ONE image, the lane offset of an outside
// pixel lies past it), so the commit has no padding select either.
template <int MB, int NB, int WM, int WN, int PT, int MODE, bool SPLIT = false, bool IOB = false, bool PAIR = false, bool NOSC = false>
__global__ __launch_bounds__(BNT, (MB == 1 && MODE == 0 && PT <= 3 && !SPLIT) ? 4 : 2) void conv_bf16_kernel(const ConvK p) {
  static_assert(WM * WN == 4, "four waves per workgroup");
  static_assert(!(SPLIT && IOB), "the split-precision form keeps fp32 activations");
  static_assert(!PAIR || IOB, "pixel-pair staging is the bf16-activation form");
  static_assert(!NOSC || PAIR, "the copy-only commit is a pair-staging form");
  using AT = typename std::conditional<IOB, vsp::bf16_t, float>::type;  // activation element in HBM
  constexpr unsigned ES = sizeof(AT);
  constexpr int NPART = SPLIT ? 2 : 1;
  constexpr bool S2 = MODE == M_S2, TCV = MODE == M_TC;
  constexpr int NACC = TCV ? 4 * NB : NB;  // accumulator blocks per 32-channel block: transposed = four phases per position block
  constexpr int NPL = S2 ? 4 : 1;          // patch planes per channel octet (stride 2: parity planes)
  // 32-channel tiles (40 KB LDS) live on occupancy instead (<= 128 VGPRs); the transposed mode has 128 accumulator registers
#ifndef VSP_X3_DEEP
#define VSP_X3_DEEP 0   // two-set prefetch under SPLIT measured equal (three MFMAs per product already cover the load latency)
#endif
  constexpr bool DEEP = PT <= 3 && MODE != M_TC && (SPLIT ? bool(VSP_X3_DEEP) : !(MB == 1 && MODE == M_CONV));
#ifndef VSP_BF16_COMMIT_FIRST
#define VSP_BF16_COMMIT_FIRST 0   // 1: convert at the top of the interval and issue the weight DMA there (measured equal: the
#endif                             // interval is bound by the load latency that the closing vmcnt(0) + barrier exposes)
  constexpr bool COMMIT_FIRST = VSP_BF16_COMMIT_FIRST;
  constexpr int CO_T = 32 * MB * WM, NPIX = 32 * NB * WN, T = 9;
  constexpr int WPART = T * 2 * CO_T;      // 16-byte units of one precision part: [tap][octet][co]
  constexpr int WSLAB = NPART * WPART;     // per weight buffer: [part][tap][octet][co]
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l32 = lane & 31, kh = lane >> 5;
  // XCD-aware work order.  The dispatcher deals workgroups round-robin over the 8 XCDs (each with its own L2), so neighbours
  // in launch order never share an L2.  Remap (bijective for any grid size) so that every XCD walks a CONTIGUOUS range of a
  // logical order in which the workgroups that read the same input are adjacent: fastest the output-channel tiles of one
  // pixel tile (same patch), then neighbouring tiles (shared halo), then images.  Dilation-group launches keep the launch
  // order (residue-major): there the column residues of a tile, which read the same 32-byte sectors, already follow each other
  // on one XCD two slots apart, and packing them into the same instant instead measured 25 % slower.
  // Shared-input dilation groups (wg_order 3, round 2): the four SMART branches read the SAME image region with different row
  // strides, and in launch order no two of them ever meet in one L2 -- measured 13-15x the input fetched (8.1 GB for a 0.54 GB
  // input on 64 -> 4x16 at 512^2), the launch fabric-bound at 4 TB/s.  Region-major order: a region = one image band of 8 TH
  // rows x 4 column tiles; for every dilation d | 8 exactly 8 workgroups per column tile cover it (d residues x 8/d row tiles),
  // so a region is G x co_tiles x 32 workgroups that one XCD runs back to back on ~1.3 MB of input.
  const int GX = gridDim.x, GY = gridDim.y, GN = GX * GY, GT = GN * gridDim.z;
  int b = blockIdx.z, xl = blockIdx.x, yy = blockIdx.y;
  int reg_ry = -1, reg_ty = 0, reg_tx = 0;
  if (p.G == 1 || (MODE == M_CONV && p.wg_order == 3)) {   // (wg_order 2: G == 1 only)
    const int wgid = blockIdx.x + GX * (blockIdx.y + GY * blockIdx.z);
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    if (p.G == 1 && p.wg_order == 2) {
      // weight-heavy layers (the 512 -> 5632 stem: 52 MB of bf16 weights, 88 channel tiles, 128 pixel tiles): with the channel tiles
      // fastest every pixel tile streams ALL the weights through its XCD's 4 MB L2 -- PMC FETCH_SIZE 7.2 GB for 0.56 GB of operands,
      // the launch moving 3.9 TB/s through the fabric.  Here a GROUP of p.tiles_x channel tiles (their weights fit the L2) sweeps every
      // pixel tile of every image before the next group starts: the weights are fetched once per group, the (smaller) input once per
      // group sweep.
      const int cgs = p.tiles_x, npt = GX * (int)gridDim.z;         // channel tiles per group, pixel tiles of the batch
      const int full = (GY / cgs) * cgs * npt;                      // workgroups of the whole groups
      int cgrp, rem, cw;
      if (lid < full) { cgrp = lid / (cgs * npt); rem = lid - cgrp * (cgs * npt); cw = cgs; }
      else { cgrp = GY / cgs; rem = lid - full; cw = GY - cgrp * cgs; }   // the last, narrower group
      const int pt = rem / cw;
      yy = cgrp * cgs + (rem - pt * cw);
      b = pt / GX;
      xl = pt - b * GX;
    } else if (p.G == 1) {
      b = lid / GN;
      const int lrem = lid - b * GN;
      xl = lrem / GY;
      yy = lrem - xl * GY;
    } else {
      constexpr int CGX = 4;
      const int nb = p.tiles_y, ncg = p.tiles_x;               // (host: bands per image, column groups per band)
      const int per_region = GY * 8 * CGX;
      const int region = lid / per_region, w = lid - region * per_region;
      b = region / (nb * ncg);
      const int rr = region - b * (nb * ncg);
      const int band = rr / ncg, cg = rr - band * ncg;
      const int slot = w / (GY * CGX), w2 = w - slot * (GY * CGX);
      const int cx = w2 / GY;
      yy = w2 - cx * GY;
      const int dg = p.dil[yy / p.co_tiles];
      reg_ry = slot % dg;
      reg_ty = band * (8 / dg) + slot / dg;
      reg_tx = cg * CGX + cx;
    }
  }
  const int g = yy / p.co_tiles, ct = yy - g * p.co_tiles;
  const int d = MODE == M_CONV ? p.dil[p.G > 4 ? 0 : g] : 1;
  // extent of the tile grid: the sub-image of one residue class / the stride-2 output / the (H+1) x (W+1) position grid
  const int SH = TCV ? p.H + 1 : S2 ? p.OH : (p.H + d - 1) / d, SW = TCV ? p.W + 1 : S2 ? p.OW : p.W;
  const int twl = p.tw_log2, TW = 1 << twl, TH = NPIX >> twl;
  const int tiles_x = (SW + TW - 1) >> twl, tiles_y = (SH + TH - 1) / TH;
  const int per_res = tiles_x * tiles_y;
  int ry, ty_i, tx_i;
  if (reg_ry >= 0) {
    if (reg_ty >= tiles_y || reg_tx >= tiles_x) return;          // (bands / column groups that the image does not fill)
    ry = reg_ry; ty_i = reg_ty; tx_i = reg_tx;
  } else {
    if (xl >= per_res * d) return;                               // (row counts that d does not divide leave a few spare blocks)
    ry = xl / per_res;
    const int tile_i = xl - ry * per_res;
    ty_i = tile_i / tiles_x;
    tx_i = tile_i - ty_i * tiles_x;
  }
  const int oy0 = ty_i * TH, ox0 = tx_i << twl;                // sub-image coordinates
  const int co0 = ct * CO_T;
  const int co_pad = (p.cout_g + 31) & ~31;
  const int chw = p.H * p.W;
  const int nchunk = (p.Cin + BCK - 1) / BCK;
  const int pitch = p.bf_pitch, PR = MODE == M_CONV ? TH + 2 : TH + 1, PC = MODE == M_CONV ? TW + 2 * d : TW + 1;
  const int PLANE = p.bf_plane;     // >= PR * pitch (stride 2: == 8 mod 16 so that the two planes a store hits do not collide)
  const int PPART = 2 * NPL * PLANE;   // 16-byte units of one precision part: [octet][plane][position]
  const int PBUF = NPART * PPART;      // per patch buffer

  u32x4* Wl = smem16;               // 2 x [T][2][CO_T]
  u32x4* Pl = smem16 + 2 * WSLAB;   // 2 x [2][NPL][PLANE]

  // ---- patch staging: wave w serves channel octet (w & 1), positions (w >> 1) * 64 + lane + 128 e
  const int oct = wave & 1;
  const int pbase = (wave >> 1) * 64 + lane;
  // task index -> (image offset, LDS slot).  Stride 1 / transposed: the task index IS the plane position.  Stride 2: tasks
  // run over the input patch in raster order (2 TH + 1 rows of 2 PC columns) and scatter into the four parity planes.
  const int RC = S2 ? 2 * PC : pitch;
  const int NTASK = S2 ? (2 * PR - 1) * RC : PR * pitch;
  unsigned poff[PT];  // BYTE offset inside a channel plane, unsigned 32-bit: the loads take the scalar-base + lane-offset form
  int pdst[PT];
  int pdst1[PAIR ? PT : 1];   // (PAIR) LDS slot of the task's second pixel
  unsigned pin = 0, pwr = 0;  // bit e: position inside the image / position exists in the plane
  unsigned pwr1 = 0;          // (PAIR) second pixel exists in the plane
  if constexpr (PAIR) {
    // tasks = pixel PAIRS at even image columns (W is even: a pair lies wholly inside or wholly outside a row, its load is 4-byte aligned).
    // Patch column 0 is image column x_first; when that is odd it is the SECOND pixel of its pair (s0 = 1).
    const int x_first = S2 ? 2 * ox0 - p.padx[0] : ox0 - d;
    const int s0 = x_first & 1;
    const int NCOL = S2 ? 2 * PC - 1 : PC, NROW = S2 ? 2 * PR - 1 : PR;
    const int NPC = (NCOL + s0 + 1) >> 1;
    const int NT2 = NROW * NPC;
#pragma unroll
    for (int e = 0; e < PT; ++e) {
      const int idx = pbase + 128 * e;
      const int r = idx / NPC, c0 = 2 * (idx - r * NPC) - s0;
      int iy;
      bool row_in;
      if constexpr (S2) {
        iy = 2 * oy0 - p.pady[0] + r;
        row_in = iy >= 0 && iy < p.H;
      } else {
        const int sy = oy0 - 1 + r;
        iy = sy * d + ry;
        row_in = sy >= 0 && iy < p.H;
      }
      const int ix = x_first + c0;
      const bool task = idx < NT2;
      const bool in = task && row_in && ix >= 0 && ix < p.W;
      auto slot = [&](int c) { return S2 ? ((r & 1) * 2 + (c & 1)) * PLANE + (r >> 1) * pitch + (c >> 1) : r * pitch + c; };
      pdst[e] = slot(c0);
      pdst1[e] = slot(c0 + 1);
      poff[e] = in ? (unsigned)(iy * p.W + ix) * ES : (NOSC ? 0x7ffffff0u : 0u);   // (NOSC: past the image's descriptor -> the load returns zeros)
      pin |= in ? (1u << e) : 0u;
      pwr |= (task && c0 >= 0) ? (1u << e) : 0u;
      pwr1 |= (task && c0 + 1 < NCOL) ? (1u << e) : 0u;
    }
  } else {
#pragma unroll
  for (int e = 0; e < PT; ++e) {
    const int idx = pbase + 128 * e;
    const int r = idx / RC, c = idx - r * RC;
    bool wr, in;
    int iy, ix;
    if constexpr (S2) {
      wr = idx < NTASK && c < 2 * PC - 1;
      iy = 2 * oy0 - p.pady[0] + r;
      ix = 2 * ox0 - p.padx[0] + c;
      in = wr && iy >= 0 && ix >= 0 && iy < p.H && ix < p.W;
      pdst[e] = ((r & 1) * 2 + (c & 1)) * PLANE + (r >> 1) * pitch + (c >> 1);
    } else {
      wr = idx < NTASK && c < PC;
      const int sy = oy0 - 1 + r, sx = ox0 - d + c;   // (transposed mode: d = 1)
      iy = sy * d + ry;
      ix = sx;
      in = wr && sy >= 0 && sx >= 0 && iy < p.H && ix < p.W;
      pdst[e] = idx;
    }
    poff[e] = in ? (unsigned)(iy * p.W + ix) * ES : 0u;
    pin |= in ? (1u << e) : 0u;
    pwr |= wr ? (1u << e) : 0u;
  }
  }
  const AT* xb = reinterpret_cast<const AT*>(p.x) + ((int64_t)b * p.x_ch + (int64_t)g * p.x_gs) * chw;
  // Two register sets: chunk k lives in set k & 1.  Its loads are issued TWO intervals before its MFMAs (top of interval
  // k - 2), its conversion + LDS write is spread over the tap loop of interval k - 1 (VALU work in the shadow of the MFMAs).
  float pregA[PT][8], pregB[PT][8];
  // (NOSC: the descriptor ends with this group's last channel plane -- the range check compares the lane offset with size - scalar offset)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<AT*>(xb), 0, NOSC ? p.Cin * chw * (int)ES : 0x7fffffff, 0x00020000);
  const float* iscp = p.in_scale + (int64_t)b * p.in_scale_bstride;
  const float* ishp = p.in_shift;
  // Cin is a multiple of 8 (host): a wave's channel octet is either wholly inside or wholly past Cin
  auto issue_p = [&](float (&pr)[PT][8], int c) {
#ifdef VSP_BF16_ABLATE  // 0x200000 no patch loads after the prologue
    if ((p.dbg & 0x200000) && c > 0) return;
#endif
    const int cib = c * BCK + 8 * oct;
    if (cib >= p.Cin) return;                                  // wave-uniform; the octet is committed as zeros
    // buffer loads: resource = this image, scalar offset = the channel plane (wave-uniform), vector offset = the lane's 32-bit
    // byte offset inside a plane.  (With flat pointers hipcc hoists 24 loop-invariant 64-bit per-lane addresses = 48 VGPRs.)
    int soff = cib * chw * (int)ES;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int e = 0; e < PT; ++e) {  // (tasks past the plane read element 0 and are never written)
        if constexpr (PAIR)  // two neighbouring pixels: one aligned 4-byte load
          pr[e][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)poff[e], soff, 0));
        else if constexpr (IOB)  // the raw 16 bits (sign-extended by the load); widened to fp32 when the value is committed
          pr[e][j] = __builtin_bit_cast(float, (int)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, (int)poff[e], soff, 0));
        else
          pr[e][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)poff[e], soff, 0));
      }
      soff += chw * (int)ES;
    }
  };
  auto load_scales = [&](int c, float (&sc)[8], float (&sh)[8]) -> bool {  // wave-uniform: scalar loads
    if constexpr (NOSC) return c * BCK + 8 * oct < p.Cin;
    const int cib = min(c * BCK + 8 * oct, p.Cin - 8);
    // constant address space: a uniform load from it goes through the scalar cache (a plain global pointer does not, the
    // compiler cannot know that nothing stores to it).  Absent operands: a device constant through stride 0 (host).
    // (Round 4: the same operands from an LDS table filled once per workgroup -- what the row-vector kernel does per tile -- measured
    //  2-7 % faster on the dilation-group launches and 7 % SLOWER on the plain 64- / 128-channel layers: one more barrier in the
    //  prologue, four LDS reads and sixteen v_readfirstlane per interval; not kept.  Writing that variant turned up a compiler
    //  trap: __builtin_bit_cast(int, t[k]) on the ELEMENT expression of an ext_vector read element 0 for every k.)
    typedef const float __attribute__((address_space(4))) * cfp4;
    cfp4 s0 = (cfp4)(uintptr_t)(iscp + cib * p.bf_isc_s);
    cfp4 h0 = (cfp4)(uintptr_t)(ishp + cib * p.bf_ish_s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = s0[j * p.bf_isc_s];
      sh[j] = h0[j * p.bf_ish_s];
    }
    return c * BCK + 8 * oct < p.Cin;
  };
  auto commit_one = [&](u32x4* Pdst, const float (&pr)[PT][8], const float (&sc)[8], const float (&sh)[8], bool oct_ok, int e) {
    if constexpr (NOSC) {
      constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;   // v_perm_b32 selectors: the low / the high halves of (S1, S0)
      u32x4 p0 = {0u, 0u, 0u, 0u}, p1 = p0;
      if (oct_ok) {   // (wave-uniform; an octet past Cin was never loaded: zeros)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned a = __builtin_bit_cast(unsigned, pr[e][2 * k]), bq_ = __builtin_bit_cast(unsigned, pr[e][2 * k + 1]);
          p0[k] = __builtin_amdgcn_perm(bq_, a, LO);
          p1[k] = __builtin_amdgcn_perm(bq_, a, HI);
        }
      }
      if ((pwr >> e) & 1u) Pdst[oct * NPL * PLANE + pdst[e]] = p0;
      if ((pwr1 >> e) & 1u) Pdst[oct * NPL * PLANE + pdst1[e]] = p1;
      return;
    }
    if constexpr (PAIR) {
      const bool in = ((pin >> e) & 1u) && oct_ok;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (!(((h ? pwr1 : pwr) >> e) & 1u)) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned w2 = __builtin_bit_cast(unsigned, pr[e][j]);
          const float xv = __builtin_bit_cast(float, h ? (w2 & 0xffff0000u) : (w2 << 16));
          v[j] = fmaf(xv, sc[j], sh[j]);
        }
        // (the padding select on the four packed words instead of the eight values: a pixel outside the image loaded element 0 of its plane)
        const u32x4 pk = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        Pdst[oct * NPL * PLANE + (h ? pdst1[e] : pdst[e])] = in ? pk : u32x4{0u, 0u, 0u, 0u};
      }
      return;
    }
    if (!((pwr >> e) & 1u)) return;
    const bool in = ((pin >> e) & 1u) && oct_ok;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xv = IOB ? __builtin_bit_cast(float, __builtin_bit_cast(unsigned, pr[e][j]) << 16) : pr[e][j];
      v[j] = in ? fmaf(xv, sc[j], sh[j]) : 0.f;
    }
    const u32x4 hi = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
    Pdst[oct * NPL * PLANE + pdst[e]] = hi;
    if constexpr (SPLIT) {  // lo = bf16(v - hi): the next eight significant bits
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned hb = (j & 1) ? (hi[j >> 1] & 0xffff0000u) : (hi[j >> 1] << 16);
        r[j] = v[j] - __builtin_bit_cast(float, hb);
      }
      Pdst[PPART + oct * NPL * PLANE + pdst[e]] =
          u32x4{pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3]), pack_bf16(r[4], r[5]), pack_bf16(r[6], r[7])};
    }
  };

  // ---- weight slab by LDS-DMA: the chunk's [tap][octet] rows of this co tile, 64 rows (1 KiB) per wave instruction
  constexpr int NDMA = WSLAB / 64;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w) + (int64_t)b * p.w_bs + (int64_t)g * nchunk * (NPART * T * 2) * co_pad;
  auto issue_w = [&](u32x4* Wdst, int c) {
#ifdef VSP_BF16_ABLATE  // tuning only (VSP_CONV_DBG): 0x100000 no weight DMA after the prologue, 0x400000 every chunk re-reads chunk 0's slab
    if ((p.dbg & 0x100000) && c > 0) return;
    if (p.dbg & 0x400000) c = 0;
#endif
    const u32x4* src = wsrc + (int64_t)c * (NPART * T * 2) * co_pad;
#pragma unroll
    for (int k = 0; k < (NDMA + 3) / 4; ++k) {
      const int i = wave + 4 * k;
      const int L = i * 64 + lane;
      const int row = L / CO_T, co = L - row * CO_T;
      if (i < NDMA && co0 + co < co_pad)
        __builtin_amdgcn_global_load_lds(src + row * co_pad + co0 + co, Wdst + i * 64, 16, 0, 0);
    }
  };

  // ---- fragments
  int pixpos[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (wn * NB + nb) * 32 + l32;
    pixpos[nb] = (n >> twl) * pitch + (n & (TW - 1)) + kh * NPL * PLANE;
  }
  const int a_lane = kh * CO_T + wm * MB * 32 + l32;

  f32x16 acc[MB][NACC];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NACC; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  // ---- pipeline: at the top of interval c, W[c&1] and P[c&1] hold chunk c (made visible by the closing barrier of c-1),
  //      the register set (c+1)&1 holds the raw patch of chunk c+1
  auto interval = [&](int c, float (&prLoad)[PT][8], float (&prCommit)[PT][8]) {
    const int cur = c & 1, nxt = cur ^ 1;
    // order matters: with an LDS-DMA in flight hipcc waits for vmcnt(0) at the first use of ANY loaded register, so the
    // weight DMA of chunk c+1 is issued only after the last commit of this interval (PT taps in); the patch loads of
    // chunk c+2 go first and have the whole interval to land
    float sc[8], sh[8];
    const bool oct_ok = load_scales(c + 1, sc, sh);
    u32x4* Pn = Pl + nxt * PBUF;
    if constexpr (DEEP && COMMIT_FIRST) {
      // everything the previous interval loaded is complete (its closing barrier waited for vmcnt(0)): convert it first, then
      // give the weight DMA and the next patch loads the WHOLE interval to land
#pragma unroll
      for (int e = 0; e < PT; ++e) commit_one(Pn, prCommit, sc, sh, oct_ok, e);
      if (c + 1 < nchunk) issue_w(Wl + nxt * WSLAB, c + 1);
      if (c + 2 < nchunk) issue_p(prLoad, c + 2);
    } else if constexpr (DEEP) {
      if (c + 2 < nchunk) issue_p(prLoad, c + 2);
    } else {
      if (c + 1 < nchunk) {
        issue_w(Wl + nxt * WSLAB, c + 1);
        issue_p(prCommit, c + 1);
      }
    }
    constexpr int DMA_TAP = PT - 1;
    const u32x4* Wc = Wl + cur * WSLAB + a_lane;
    const u32x4* Pc = Pl + cur * PBUF;
    if constexpr (TCV) {
      bf16x8 bq[NPART][NB][4];  // the position block shifted by (-(ky >> 1), -(kx >> 1)): four distinct fragments serve the nine taps
#pragma unroll
      for (int pt = 0; pt < NPART; ++pt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            bq[pt][nb][s4] = __builtin_bit_cast(bf16x8, Pc[pt * PPART + pixpos[nb] + (1 - (s4 >> 1)) * pitch + (1 - (s4 & 1))]);
#pragma unroll
      for (int tap = 0; tap < T; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const int ph = (ky & 1) * 2 + (kx & 1), s4 = (ky >> 1) * 2 + (kx >> 1);
        bf16x8 a[NPART][MB];
#pragma unroll
        for (int pt = 0; pt < NPART; ++pt)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) a[pt][mb] = __builtin_bit_cast(bf16x8, Wc[pt * WPART + tap * 2 * CO_T + mb * 32]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            f32x16& ac = acc[mb][nb * 4 + ph];
            if constexpr (SPLIT) {
              ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mb], bq[0][nb][s4], ac, 0, 0, 0);
              ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mb], bq[1][nb][s4], ac, 0, 0, 0);
            }
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mb], bq[0][nb][s4], ac, 0, 0, 0);
          }
        if constexpr (DEEP && !COMMIT_FIRST) {
          if (tap < PT) commit_one(Pn, prCommit, sc, sh, oct_ok, tap);
          if (tap == DMA_TAP && c + 1 < nchunk) issue_w(Wl + nxt * WSLAB, c + 1);
        }
      }
    } else {
      // hand-pipelined: the fragments of tap t+1 are requested before the MFMAs of tap t, and a scheduling barrier per tap
      // keeps the compiler from hoisting all 36 fragment reads to the top (256 VGPRs and spills otherwise)
      bf16x8 a[2][NPART][MB], bq[2][NPART][NB];
      auto load_tap = [&](int tap, bf16x8 (&af)[NPART][MB], bf16x8 (&bf)[NPART][NB]) {
        const int ky = tap / 3, kx = tap % 3;
        const int toff = S2 ? ((ky & 1) * 2 + (kx & 1)) * PLANE + (ky >> 1) * pitch + (kx >> 1) : ky * pitch + kx * d;
#pragma unroll
        for (int pt = 0; pt < NPART; ++pt) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[pt][mb] = __builtin_bit_cast(bf16x8, Wc[pt * WPART + tap * 2 * CO_T + mb * 32]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bf[pt][nb] = __builtin_bit_cast(bf16x8, Pc[pt * PPART + pixpos[nb] + toff]);
        }
      };
      load_tap(0, a[0], bq[0]);
#pragma unroll
      for (int tap = 0; tap < T; ++tap) {
        const int cs = tap & 1;
        if (tap + 1 < T) load_tap(tap + 1, a[cs ^ 1], bq[cs ^ 1]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if constexpr (SPLIT) {  // small terms first
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][1][mb], bq[cs][0][nb], acc[mb][nb], 0, 0, 0);
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0][mb], bq[cs][1][nb], acc[mb][nb], 0, 0, 0);
            }
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cs][0][mb], bq[cs][0][nb], acc[mb][nb], 0, 0, 0);
          }
        if constexpr (DEEP && !COMMIT_FIRST) {
          if (tap < PT) commit_one(Pn, prCommit, sc, sh, oct_ok, tap);
          if (tap == DMA_TAP && c + 1 < nchunk) issue_w(Wl + nxt * WSLAB, c + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (!DEEP) {
      if (c + 1 < nchunk) {
#pragma unroll
        for (int e = 0; e < PT; ++e) commit_one(Pn, prCommit, sc, sh, oct_ok, e);
      }
    }
    __syncthreads();
  };
  // Epilogue operands of the tile's channels: ONE table in LDS (scale = out_scale x ch_scale, bias = ch_bias + bias1, bias2, slope2),
  // filled while the first loads are in flight.  (Per row / per lane loads in the epilogue each cost a round trip through the scalar
  // or vector cache: six dependent operands per output row -- the row-vector kernel's tuning build showed what that costs.)
  float4* Ept = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(smem16) + p.bf_tab);
  {
    issue_w(Wl, 0);
    issue_p(pregA, 0);
    if (DEEP && nchunk > 1) issue_p(pregB, 1);
    if (tid < CO_T) {
      const int cgl = min(co0 + tid, p.cout_g - 1);   // (channels past the group: clamped, never stored)
      const int co = g * p.cout_g + cgl;
      const float* ospb = p.osp + (int64_t)b * (p.G * p.cout_g) * p.oss;
      Ept[tid] = float4{ospb[co * p.oss] * p.csp[co * p.css], p.cbp[co * p.cbs] + p.b1p[co * p.b1s], p.b2p[co * p.b2s], p.s2p[co * p.s2s]};
    }
    float sc[8], sh[8];
    const bool oct_ok = load_scales(0, sc, sh);
#pragma unroll
    for (int e = 0; e < PT; ++e) commit_one(Pl, pregA, sc, sh, oct_ok, e);
    __syncthreads();
  }
  if constexpr (DEEP) {
    for (int c = 0; c < nchunk; c += 2) {
      interval(c, pregA, pregB);
      if (c + 1 < nchunk) interval(c + 1, pregB, pregA);
    }
  } else {
    for (int c = 0; c < nchunk; ++c) interval(c, pregA, pregA);
  }

  // ---- epilogue.  The accumulators hold one pixel x 16 channels per lane; stored like that every store instruction moves
  // 4 bytes per lane and every channel operand is a per-lane load (measured: half of a 64-channel 512^2 layer).  Instead the
  // tile is transposed through LDS, one 32-channel block row (mb) at a time: E[channel][pixel] fp32, then one lane owns 4
  // consecutive pixels of one channel -> 16-byte noise / residual loads and stores, and (256-pixel tiles) the channel is
  // wave-uniform, so its six operands come through the scalar cache.
  const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW;
  const float nw = p.nwp[0];
  const float s1 = p.s1, g1 = p.g1, g2 = p.g2;
  AT* yb = reinterpret_cast<AT*>(p.y) + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const AT* r1b = reinterpret_cast<const AT*>(p.r1p) + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w;
  const AT* r2b = reinterpret_cast<const AT*>(p.r2p) + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w;
  const bool has_nz = p.nzs != 0, has_r1 = p.r1s != 0, has_r2 = p.r2s != 0;
  const int y_plane = p.y_h * p.y_w;
  using vsp::f32x4u;
  constexpr int Q = NPIX / 4;                 // pixel quads per channel row
  constexpr int EROWS = 32 * WM;              // channel rows per pass
  constexpr int EIT = EROWS * Q / BNT;
  float* El = reinterpret_cast<float*>(smem16);  // [EROWS][NPIX]
  if constexpr (TCV) {
    // position (m, n) feeds outputs (2m + py, 2n + px); no noise / residual here (they follow the blur in the reference).
    // The two px phases of a lane are neighbours in memory: one 8-byte store (rows of 2W+1 floats are only 4-byte aligned)
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = (wn * NB + nb) * 32 + l32;
      const int m = oy0 + (n >> twl), c = ox0 + (n & (TW - 1));
      // (no `continue` in front of the operand loads: a divergent branch there makes the compiler wait for every load in flight at the
      //  join, and the 6 x 16 x MB per-channel operands of a lane then leave one round trip after the other; only the stores are predicated)
      const bool pos_ok = m <= p.H && c <= p.W;
      const bool pair = c < p.W;  // column 2c + 1 exists
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int cg_ = co0 + (wm * MB + mb) * 32 + 8 * (i >> 2) + 4 * kh + (i & 3);
          const bool ch_ok = cg_ < p.cout_g;
          const int cg = ch_ok ? cg_ : p.cout_g - 1;
          const int co = g * p.cout_g + cg;
          const float4 e4 = Ept[cg - co0];
          const float os = e4.x, cb = e4.y, b2 = e4.z, sl2 = e4.w;
          AT* yc = yb + (int64_t)co * y_plane;
          auto fin = [&](float v) {
            v = v * os + cb;
            v = (v > 0.f ? v : v * s1) * g1 + b2;
            return (v > 0.f ? v : v * sl2) * g2;
          };
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            if (!pos_ok || !ch_ok || m + py > p.H) continue;  // row 2m + 1 exists only for m < H
            const int yo = (2 * m + py) * p.y_w + 2 * c;
            const float v0 = fin(acc[mb][nb * 4 + py * 2][i]), v1 = fin(acc[mb][nb * 4 + py * 2 + 1][i]);
            if constexpr (IOB) {
              typedef unsigned u32h __attribute__((aligned(2)));
              if (pair)
                *reinterpret_cast<u32h*>(yc + yo) = vsp::bf16_pack(v0, v1);
              else
                vsp::Elem<AT>::store1(yc + yo, v0);
            } else {
              if (pair)
                *reinterpret_cast<f32x2u*>(yc + yo) = f32x2u{v0, v1};
              else
                yc[yo] = v0;
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (mb > 0) __syncthreads();  // (the chunk loop ended on a barrier)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        El[(wm * 32 + 8 * (i >> 2) + 4 * kh + (i & 3)) * NPIX + (wn * NB + nb) * 32 + l32] = acc[mb][TCV ? 0 : nb][i];
    __syncthreads();
    auto operands = [&](int row, int& co, float& os, float& cb, float& b2, float& sl2) -> bool {
      const int cg = co0 + ((row >> 5) * MB + mb) * 32 + (row & 31);
      const bool ok = cg < p.cout_g;
      co = g * p.cout_g + (ok ? cg : p.cout_g - 1);   // (clamped: the loads of the residuals are unconditional, the store is not)
      const float4 e4 = Ept[cg - co0];                // (rows past the group hold the clamped channel's operands)
      os = e4.x; cb = e4.y; b2 = e4.z; sl2 = e4.w;
      return ok;
    };
    const bool quads = (p.OW & 3) == 0 && p.OW >= 4;   // (uniform) a lane's four pixels are inside or outside the row together
    {
#pragma unroll 2
      for (int it = 0; it < EIT; ++it) {
        int row, q;
        if constexpr (Q == 64) {
          row = it * 4 + wave;  // wave-uniform
          q = lane;
        } else {
          const int L = it * BNT + tid;
          row = L / Q;
          q = L - row * Q;
        }
        int co;
        float os, cb, b2, sl2;
        const bool ch_ok = operands(row, co, os, cb, b2, sl2);
        const f32x4 av = *reinterpret_cast<const f32x4*>(El + row * NPIX + 4 * q);
        const int n = 4 * q;
        const int oy = (oy0 + (n >> twl)) * d + ry, ox = ox0 + (n & (TW - 1));   // polyphase rows, dense columns
        auto fin = [&](float v, float nzv, float r1v, float r2v) {
          v = v * os + cb;
          v = (v > 0.f ? v : v * s1) * g1;
          v += nzv * nw + b2;
          v = (v > 0.f ? v : v * sl2) * g2;
          return v + r1v + r2v;
        };
        if (quads) {
          // every load on clamped coordinates, behind uniform tests only; the store alone is predicated (see the transposed branch)
          const bool inside = ch_ok && oy < p.OH && ox < p.OW;
          const int oyc = min(oy, p.OH - 1), oxc = min(ox, p.OW - 4);
          const int ro = co * y_plane + oyc * p.y_w + oxc;
          f32x4u nzv = {0.f, 0.f, 0.f, 0.f}, r1v = nzv, r2v = nzv;
          if (has_nz) nzv = *reinterpret_cast<const f32x4u*>(nzp + oyc * p.OW + oxc);
          if (has_r1) r1v = vsp::Elem<AT>::load4(r1b + ro);
          if (has_r2) r2v = vsp::Elem<AT>::load4(r2b + ro);
          const f32x4u o4 = {fin(av[0], nzv[0], r1v[0], r2v[0]), fin(av[1], nzv[1], r1v[1], r2v[1]),
                             fin(av[2], nzv[2], r1v[2], r2v[2]), fin(av[3], nzv[3], r1v[3], r2v[3])};
          if (inside) vsp::Elem<AT>::store4(yb + ro, o4);
          continue;
        }
        if (!ch_ok || oy >= p.OH || ox >= p.OW) continue;
        const int ro = co * y_plane + oy * p.y_w + ox;
        if (ox + 3 < p.OW) {
          f32x4u nzv = {0.f, 0.f, 0.f, 0.f}, r1v = nzv, r2v = nzv;
          if (has_nz) nzv = *reinterpret_cast<const f32x4u*>(nzp + oy * p.OW + ox);
          if (has_r1) r1v = vsp::Elem<AT>::load4(r1b + ro);
          if (has_r2) r2v = vsp::Elem<AT>::load4(r2b + ro);
          const f32x4u o4 = {fin(av[0], nzv[0], r1v[0], r2v[0]), fin(av[1], nzv[1], r1v[1], r2v[1]),
                             fin(av[2], nzv[2], r1v[2], r2v[2]), fin(av[3], nzv[3], r1v[3], r2v[3])};
          vsp::Elem<AT>::store4(yb + ro, o4);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (ox + j >= p.OW) continue;
            vsp::Elem<AT>::store1(yb + ro + j, fin(av[j], has_nz ? nzp[oy * p.OW + ox + j] : 0.f, has_r1 ? vsp::Elem<AT>::load1(r1b + ro + j) : 0.f,
                                                   has_r2 ? vsp::Elem<AT>::load1(r2b + ro + j) : 0.f));
          }
        }
      }
    }
  }
}

struct BfGeom {
  int twl, pitch, plane, pt, tab;
  int pt2;      // staging tasks per thread when a task is a pixel pair (PAIR)
  size_t lds;
};

static int bf_pitch(int twl, int pc) {
  // 32-pixel B blocks must land on 16 distinct positions (mod 16) per LDS access group: a row of 32 needs nothing,
  // two rows of 16 need pitch == 0 (mod 16), four rows of 8 need pitch == 8 (mod 16)
  if (twl == 5) return pc;
  if (twl == 4) return (pc + 15) & ~15;
  int pch = (pc & ~15) + 8;
  return pch >= pc ? pch : pch + 16;
}

static BfGeom bf_geom(const ConvK& q, int mode, int co_t, int npix, int erows, int npart) {
  int sw;  // width of the tile grid (the narrowest sub-image of a dilation-group launch decides the tile width)
  if (mode == M_TC) sw = q.W + 1;
  else if (mode == M_S2) sw = q.OW;
  else {
    sw = q.W;
  }
  int dmax = 1;
  if (mode == M_CONV)
    for (int g = 0; g < (q.G > 4 ? 1 : q.G); ++g) dmax = q.dil[g] > dmax ? q.dil[g] : dmax;
  BfGeom r;
  r.twl = sw >= 32 ? 5 : (sw > 8 ? 4 : 3);
  const int TW = 1 << r.twl, TH = npix >> r.twl;
  const int PR = mode == M_CONV ? TH + 2 : TH + 1, PC = mode == M_CONV ? TW + 2 * dmax : TW + 1;
  r.pitch = bf_pitch(r.twl, PC);
  r.plane = PR * r.pitch;
  int ntask = r.plane, npl = 1;
  if (mode == M_S2) {
    r.plane = (r.plane & ~15) + 8 >= r.plane ? (r.plane & ~15) + 8 : (r.plane & ~15) + 24;  // == 8 (mod 16)
    ntask = (2 * PR - 1) * 2 * PC;
    npl = 4;
  }
  r.pt = (ntask + 127) / 128;
  r.pt2 = ((mode == M_S2 ? (2 * PR - 1) * PC : PR * ((PC + 2) / 2)) + 127) / 128;   // (rows x pairs per row, the odd first column included)
  r.lds = ((size_t)2 * 9 * 2 * co_t + (size_t)2 * 2 * npl * r.plane) * 16 * npart;
  const size_t epi = (size_t)erows * npix * sizeof(float);  // epilogue transpose buffer
  if (epi > r.lds) r.lds = epi;
  r.tab = (int)r.lds;                                        // the per-channel epilogue operand table behind both (16 B per output channel of the tile)
  r.lds += (size_t)co_t * 16;
  return r;
}

template <int MB, int NB, int WM, int WN, int PT, int MODE, bool SPLIT = false, bool IOB = false, bool PAIR = false, bool NOSC = false>
int launch_bf(ConvK q, const BfGeom& gm, hipStream_t stream) {
  if constexpr (!SPLIT && !IOB) {
    // copy-only pair staging (NOSC): per-image weights (the entry checked even rows and the alignment of x), and every launch WITHOUT an
    // input scale / shift (plain convolutions: the SMART fusion layers, the encoder's head stages) -- bf16(x * 1 + 0) is x
    const bool plain = q.bf_isc_s == 0 && q.bf_ish_s == 0 && (q.W & 1) == 0 && (reinterpret_cast<uintptr_t>(q.x) & 3) == 0;
    if (q.io_bf16 && (q.w_bs != 0 || plain)) {
      if (gm.pt2 <= 1) return launch_bf<MB, NB, WM, WN, 1, MODE, false, true, true, true>(q, gm, stream);
      if (gm.pt2 <= 2) return launch_bf<MB, NB, WM, WN, 2, MODE, false, true, true, true>(q, gm, stream);
      if (gm.pt2 <= 3) return launch_bf<MB, NB, WM, WN, 3, MODE, false, true, true, true>(q, gm, stream);
      if (gm.pt2 <= 5) return launch_bf<MB, NB, WM, WN, 5, MODE, false, true, true, true>(q, gm, stream);
      if (q.w_bs != 0)
        return vsp::fail(VSP_ENOTSUP, "conv2d_bf16: per-image weights: the patch plane of this tile needs more than five pair tasks per thread");
    }
    if (q.io_bf16) {
      // bf16 activations: pixel-pair staging wherever rows are 4-byte multiples (env VSP_BF16_PAIR=0: the one-pixel tasks, for A/B runs)
      static const bool pair_env = !(vsp::tune_env("VSP_BF16_PAIR") && atoi(vsp::tune_env("VSP_BF16_PAIR")) == 0);
      if (pair_env && (q.W & 1) == 0 && (reinterpret_cast<uintptr_t>(q.x) & 3) == 0) {
        if (gm.pt2 <= 1) return launch_bf<MB, NB, WM, WN, 1, MODE, false, true, true>(q, gm, stream);
        if (gm.pt2 <= 2) return launch_bf<MB, NB, WM, WN, 2, MODE, false, true, true>(q, gm, stream);
        if (gm.pt2 <= 3) return launch_bf<MB, NB, WM, WN, 3, MODE, false, true, true>(q, gm, stream);
        if (gm.pt2 <= 5 && PT >= 5) return launch_bf<MB, NB, WM, WN, 5, MODE, false, true, true>(q, gm, stream);
      }
      return launch_bf<MB, NB, WM, WN, PT, MODE, false, true>(q, gm, stream);
    }
  }
  constexpr int CO_T = 32 * MB * WM, NPIX = 32 * NB * WN;
  static vsp::LdsAttrOnce attr;   // per device
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_bf16_kernel<MB, NB, WM, WN, PT, MODE, SPLIT, IOB, PAIR, NOSC>), 150 * 1024, "conv2d_bf16")) return rc;
  q.tw_log2 = gm.twl;
  q.bf_pitch = gm.pitch;
  q.bf_plane = gm.plane;
  q.bf_tab = gm.tab;
  q.co_tiles = (q.cout_g + CO_T - 1) / CO_T;
  const int TW = 1 << gm.twl, TH = NPIX >> gm.twl;
  int blocks = 0;
  if (MODE == M_CONV) {
    for (int g = 0; g < (q.G > 4 ? 1 : q.G); ++g) {
      const int d = q.dil[g];
      const int SH = (q.H + d - 1) / d, SW = q.W;
      const int n = ((SW + TW - 1) / TW) * ((SH + TH - 1) / TH) * d;
      blocks = n > blocks ? n : blocks;
    }
  } else {
    const int EH = MODE == M_TC ? q.H + 1 : q.OH, EW = MODE == M_TC ? q.W + 1 : q.OW;
    blocks = ((EW + TW - 1) / TW) * ((EH + TH - 1) / TH);
  }
  q.wg_order = 0;
  if (MODE == M_CONV && q.G >= 2 && q.G <= 4 && q.x_gs == 0 && !(q.dbg & 0x800000)) {
    bool ok = true;
    for (int g = 0; g < q.G; ++g) ok = ok && (q.dil[g] == 1 || q.dil[g] == 2 || q.dil[g] == 4 || q.dil[g] == 8);
    if (ok) {  // region-major order over bands of 8 TH rows x 4 column tiles (see the kernel)
      q.wg_order = 3;
      q.tiles_y = (q.H + 8 * TH - 1) / (8 * TH);
      q.tiles_x = ((q.W + TW - 1) / TW + 3) / 4;
      blocks = q.tiles_y * q.tiles_x * 32;
    }
  }
  if (q.G == 1 && !(q.dbg & 0x20000)) {   // weight-heavy layers: channel-tile groups sweep the pixel tiles (see the kernel)
    const int64_t wtile = (int64_t)q.Cin * 9 * CO_T * 2 * (SPLIT ? 2 : 1), wall = wtile * q.co_tiles;   // bf16 weight bytes of one channel tile / of the layer
    const int64_t xall = (int64_t)q.B * q.Cin * q.H * q.W * (q.io_bf16 ? 2 : 4);
    if (wall > 8 * 1024 * 1024 && q.co_tiles >= 4 && wall * 2 > xall / 8) {
      int cgs = (int)((int64_t)(3 * 1024 * 1024 / 2) / wtile);     // half of an XCD's L2 for the group's weights
      cgs = cgs < 1 ? 1 : (cgs > q.co_tiles ? q.co_tiles : cgs);
      q.wg_order = 2;
      q.tiles_x = cgs;
    }
  }
  dim3 grid((unsigned)blocks, (unsigned)(q.co_tiles * q.G), (unsigned)q.B);
  conv_bf16_kernel<MB, NB, WM, WN, PT, MODE, SPLIT, IOB, PAIR, NOSC><<<grid, BNT, gm.lds, stream>>>(q);
  return VSP_OK;
}

template <int MB, int NB, int WM, int WN, int MODE>
int launch_split(const ConvK& q, hipStream_t stream) {
  const BfGeom gm = bf_geom(q, MODE, 32 * MB * WM, 32 * NB * WN, 32 * WM, 2);
  if (gm.lds > 150 * 1024) return vsp::fail(VSP_ENOTSUP, "conv2d_bf16x3: tile does not fit LDS");
  if (gm.pt <= 3) return launch_bf<MB, NB, WM, WN, 3, MODE, true>(q, gm, stream);
  if (gm.pt <= 5) return launch_bf<MB, NB, WM, WN, 5, MODE, true>(q, gm, stream);
  return vsp::fail(VSP_ENOTSUP, "conv2d_bf16x3: patch plane of %d positions is too large", gm.plane);
}

template <int MB, int NB, int WM, int WN, int MODE>
int launch_shape(const ConvK& q, hipStream_t stream) {
  const BfGeom gm = bf_geom(q, MODE, 32 * MB * WM, 32 * NB * WN, 32 * WM, 1);
  if (gm.lds > 150 * 1024) return vsp::fail(VSP_ENOTSUP, "conv2d_bf16: tile does not fit LDS");
  if (gm.pt <= 3) return launch_bf<MB, NB, WM, WN, 3, MODE>(q, gm, stream);
  if (MODE != M_TC && gm.pt <= 5) return launch_bf<MB, NB, WM, WN, 5, MODE>(q, gm, stream);
  if (MODE != M_S2 && gm.pt <= 7) return launch_bf<MB, NB, WM, WN, 7, MODE>(q, gm, stream);
  return vsp::fail(VSP_ENOTSUP, "conv2d_bf16: patch plane of %d positions is too large", gm.plane);
}

}  // namespace

// mode: 0 = stride-1 conv, 1 = stride-2 conv, 2 = stride-2 transposed conv.
// variant: 0 = automatic; stride 1: 1 = 32 ch x 256 px, 2 = 64 x 256, 3 = 128 x 128, 4 = 64 x 128;
//          6 = 128 ch x 64 px, 7 = 32 ch x 128 px;  stride 2: 3 = 128 ch x 128 px, 4 = 64 ch x 128 px, 6 = 128 ch x 64 px;  transposed: 4 = 64 ch x 128 positions, 5 = 32 ch x 128 positions,
//          8 = 32 ch x 256 positions
// split = 1: the bf16x3 form (stride-1 / stride-2 modes): variants 1 = 32 ch x 256 px, 4 = 64 ch x 128 px, 7 = 32 ch x 128 px
int bf16_launch_split(const ConvK& q, int mode, int variant, hipStream_t stream) {
  if (mode == M_S2) {
    switch (variant) {
      case 7: return launch_split<1, 1, 1, 4, M_S2>(q, stream);  // 32 ch x 128 px
      case 0:   // 64 ch x 64 px: half the patch staging per MFMA (the doubled parity planes leave room for one workgroup per CU
      case 6: return launch_split<1, 1, 2, 2, M_S2>(q, stream);  // either way); stem 512 -> 5632: 3.2 -> 2.7 ms
      default: return vsp::fail(VSP_EINVAL, "conv2d_bf16x3: unknown stride-2 variant %d", variant);
    }
  }
  if (mode == M_TC) {
    switch (variant) {
      case 0:
      case 5: return launch_split<1, 1, 1, 4, M_TC>(q, stream);
      default: return vsp::fail(VSP_EINVAL, "conv2d_bf16x3: unknown transposed variant %d", variant);
    }
  }
  if (variant == 0)  // tools/bench_bf16.py with X3=1: the 32-channel tiles keep two workgroups per CU with the doubled LDS images
    variant = (q.G == 1 && (int64_t)q.H * q.W >= 64 * 64) ? 1 : 7;
  switch (variant) {
    case 1: return launch_split<1, 2, 1, 4, M_CONV>(q, stream);
    case 4: return launch_split<2, 1, 1, 4, M_CONV>(q, stream);
    case 7: return launch_split<1, 1, 1, 4, M_CONV>(q, stream);
    default: return vsp::fail(VSP_EINVAL, "conv2d_bf16x3: unknown variant %d", variant);
  }
}

int bf16_launch(const ConvK& q, int mode, int variant, hipStream_t stream) {
  if (mode == M_TC) {
    if (variant == 0) variant = 5;  // 32-channel tiles: 122 VGPRs, more workgroups in flight (tools/bench_bf16_s2t.py)
    switch (variant) {
      case 4: return launch_shape<2, 1, 1, 4, M_TC>(q, stream);
      case 5: return launch_shape<1, 1, 1, 4, M_TC>(q, stream);
      case 8: return launch_shape<1, 2, 1, 4, M_TC>(q, stream);
      default: return vsp::fail(VSP_EINVAL, "conv2d_bf16: unknown transposed variant %d", variant);
    }
  }
  if (mode == M_S2) {
    // 128 ch x 128 px for the wide heads (512 -> 5632 at 32^2: 939 -> 865 us, 512 -> 2048 at 16^2: 116 -> 90 us at B = 8; tools/bench_bf16_s2t.py)
    if (variant == 0) variant = q.cout_g >= 2048 ? 3 : (q.cout_g >= 128 && (int64_t)q.OH * q.OW <= 16 * 16) ? 6 : 4;
    switch (variant) {
      case 3: return launch_shape<2, 2, 2, 2, M_S2>(q, stream);
      case 4: return launch_shape<2, 1, 1, 4, M_S2>(q, stream);
      case 6: return launch_shape<2, 1, 2, 2, M_S2>(q, stream);
      default: return vsp::fail(VSP_EINVAL, "conv2d_bf16: unknown stride-2 variant %d", variant);
    }
  }
  if (variant == 0) {
    int dmax = q.dil[0];
    for (int g = 1; g < (q.G > 4 ? 1 : q.G); ++g) dmax = q.dil[g] > dmax ? q.dil[g] : dmax;
    const int64_t px = (int64_t)((q.H + dmax - 1) / dmax) * q.W;  // the smallest row-polyphase sub-image
    if (q.cout_g <= 32) variant = 1;
    else if (px <= 64 && q.cout_g >= 128) variant = 6;
    else if (q.G > 1) variant = px >= 512 ? 2 : 4;
    else if (q.cout_g <= 64) variant = px >= 128 * 128 ? 2 : 4;
    else variant = px >= 64 * 64 ? 2 : 4;
  }
  switch (variant) {
    case 1: return launch_shape<1, 2, 1, 4, M_CONV>(q, stream);
    case 2: return launch_shape<2, 2, 1, 4, M_CONV>(q, stream);
    case 3: return launch_shape<2, 2, 2, 2, M_CONV>(q, stream);
    case 4: return launch_shape<2, 1, 1, 4, M_CONV>(q, stream);
    case 6: return launch_shape<2, 1, 2, 2, M_CONV>(q, stream);
    case 7: return launch_shape<1, 1, 1, 4, M_CONV>(q, stream);
    default: return vsp::fail(VSP_EINVAL, "conv2d_bf16: unknown variant %d", variant);
  }
}

}  // namespace vspconv

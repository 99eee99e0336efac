// 3x3 stride-1 convolution on the bf16 matrix pipe for the LOW-CHANNEL, LARGE-MAP layers of the bf16-activation configuration
// (32 -> 32 at 1024^2, 64 -> 64 at 512^2, 128 -> 128 at 256^2 ...: BASELINE configs[2]).  Those layers sit near the HBM roofline
// on paper (64 -> 64 at 512^2: 1 byte per 36 bf16 MACs) and the general kernel (conv_bf16.hip) spends its time on the way the
// operands get into fragment order: a lane's MFMA fragment there is 8 CHANNELS of one pixel, so an NCHW image is transposed
// element by element (one 16-bit load per element, pack, ds_write) and the output goes back through an LDS transpose.
//
// "Row-vector K": the reduction index of v_mfma_f32_32x32x16_bf16 is (channel, tap) in any order both operands agree on.  Here a
// lane's eight k-values are TWO channels x FOUR consecutive pixels of one input row, [x-1, x, x+1, x+2] -- the three horizontal
// taps of that row plus one zero weight -- i.e. two 8-byte reads of the image AS IT LIES in HBM.  So
//   * the patch in LDS is a straight copy of the NCHW bf16 rows (16-byte loads, 16-byte ds_writes, the per-sample style scale
//     applied on the way: bf16(x * s) exactly as conv_bf16.hip rounds it); nothing is transposed;
//   * a B fragment is two ds_read_b64 at 2-byte granularity (gfx950 LDS takes unaligned b64), lanes 4 bytes apart;
//   * the 32 pixels of an N-block are the EVEN or the ODD pixels of a 64-pixel row segment, so a lane ends up with both
//     pixels of a pair (2j, 2j+1) for 16 channels: the epilogue packs the pair into one dword and stores it to the NCHW bf16
//     output directly from the accumulators -- a wave store covers two whole 128-byte lines, no LDS transpose, no barrier;
//   * a B fragment (input row r) feeds the three output rows r-1, r, r+1 of the wave, the A fragment of (dy, channel quad) every
//     output row and both parities: 6-8 LDS fragment reads per 48 MFMAs and k-chunk instead of 1 per MFMA.
// The price is 4/3 of the MFMAs (the zero tap), which these layers have to spare.
// Workgroup = 4 waves, wave = MB 32-channel blocks x RWV output rows x 64 pixels: 64 channels x 8 rows (MB = 2, RWV = 2), 32 x 8, 32 x 16.
// K runs in chunks of 8 channels (two MFMA k-steps per vertical tap), patch and weights double-buffered, one barrier per chunk;
// weights [chunk][dy][quad][half][co][8] bf16 (hip_ops.bf16rv_weight) come in by LDS-DMA.
// Numerics: as conv_bf16.hip (bf16 operands, exact products, fp32 accumulation, fp32 epilogue chain) in another summation order.
#include "conv_kernel.h"
#include "vsp_bf16.h"

// This file must be compiled with -fno-slp-vectorize (Makefile; tools/build_abl.sh copies the flags of the Makefile rule): with the SLP
// vectoriser on, hipcc turns the commit / epilogue arithmetic into v_pk_fma_f32 / v_pk_add_f32 with op_sel broadcasts of operands that
// have just come back from LDS, and the kernel then miscomputes 0.01-0.05 % of its outputs when two of its workgroups share a CU
// ("packed fp32"; tests/test_hip_ops.py::test_conv2d_bf16rv_repeat is the fence).  There is no pragma for the SLP
// pass, so the build rule defines this macro next to the flag and any other build path stops here instead of dropping it silently.
#ifndef VSP_BUILT_WITHOUT_SLP
#error "conv_bf16_rv.hip: build with -fno-slp-vectorize -DVSP_BUILT_WITHOUT_SLP (see vspbfr_amd/csrc/Makefile)"
#endif

namespace vspconv {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2h __attribute__((ext_vector_type(2), aligned(2)));

constexpr int RV_TW = 64;           // pixels per tile row
constexpr int RV_SEG = 10;          // 16-byte segments per patch row: columns x0 - 8 ... x0 + 71
constexpr int RV_ROWB = RV_SEG * 16;
constexpr int RV_CK = 8;            // channels per chunk
constexpr int RV_MAXC = 256;        // input channels the LDS scale table holds

// DIL = 1, 2, 4, 8: one DILATION GROUP of a SMART branch launch (the host launches the groups one after the other).  Rows are polyphase
// as in conv_bf16.hip (a tile = TH rows ry, ry + d, ... of the image).  Columns: the patch row is staged DE-INTERLEAVED into d residue
// sub-rows (pixel c of the patch row -> sub-row c % d, entry c / d), so the three taps x - d, x, x + d of a pixel are again neighbours
// in LDS and a lane's window [m-1 .. m+2] of its sub-row is read exactly as for d = 1 (three aligned dwords + a funnel shift by
// the lane's own 0 / 16 bits); a lane still owns the ADJACENT pixels (2j, 2j+1) of the image row (its two N-blocks read different
// sub-rows), so the epilogue and its packed stores do not change.
template <int MB, int RWV, int DIL = 1>   // 32-channel blocks and output rows per wave, dilation of the group
__global__ __launch_bounds__(256, 2) void conv_bf16_rv_kernel(const ConvK p, const int ntiles) {
  constexpr int TH = 4 * RWV, PR = TH + 2, CO_T = 32 * MB;
  constexpr int SEGS = DIL == 8 ? 11 : RV_SEG;      // 16-byte segments per patch row: the window reaches 2 d pixels past the tile
  constexpr int LROW = DIL == 8 ? 192 : RV_ROWB;    // bytes per patch row in LDS
  constexpr int SP = LROW / DIL;                    // ... per residue sub-row (a multiple of 4)
  constexpr int SPC = (PR * SEGS + 63) / 64;        // staging slots per channel; a wave stages two channels of every chunk
  constexpr int PT = 2 * SPC;
  constexpr int PCH = PR * LROW;                    // bytes per channel of the patch image
  constexpr int PBUF = RV_CK * PCH;
  constexpr int WROWS = 12;                         // [dy 3][quad 2][half 2] rows of CO_T fragments per chunk
  constexpr int WBUF = WROWS * CO_T * 16;
  constexpr int NGRP = 2 * (RWV + 2);               // (quad, input row) groups per chunk
  static_assert(PT <= NGRP, "one commit per fragment group");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wl = smem;                  // 2 x WBUF
  unsigned char* Pl = smem + 2 * WBUF;       // 2 x PBUF
  float4* Ep = reinterpret_cast<float4*>(smem + 2 * WBUF + 2 * PBUF);   // [tile parity][channel of the tile]: scale, bias, bias2, slope2
  float2* Sc = reinterpret_cast<float2*>(smem + 2 * WBUF + 2 * PBUF + 2 * CO_T * 16);   // [tile parity][input channel]: style scale, shift

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, kh = lane >> 5;

  // PERSISTENT workgroups: the grid is what the chip holds (two per CU); the logical tile order is co tile (same patch) -> tile
  // column -> tile row -> image, and every XCD (own L2; the dispatcher deals workgroups round-robin over the eight) gets a
  // contiguous range of workgroups.  A tile lives ~40 us, of which launch, first loads and
  // the epilogue were a third (tuning build: 237 of 790 us with everything else switched off): the first loads of tile t + 1 are
  // issued BEFORE the epilogue of tile t.
  const int tiles_x = p.W / RV_TW, tiles_y = ((p.H + DIL - 1) / DIL + TH - 1) / TH, co_tiles = p.co_tiles;
  // Tile walk: INTERLEAVED -- in step k the resident workgroups take the tiles k G ... k G + G - 1 (G = grid size), workgroup g the
  // g-th of them, every XCD a contiguous eighth.  Neighbouring tiles are then in flight at the SAME time and their shared lines (a
  // 64-pixel row segment is exactly one 128-byte line: the one-pixel left and two-pixel right halo cost two more lines per row; the
  // two halo rows) are L2 hits.  With a contiguous range of tiles per workgroup nothing that ran together shared anything: PMC
  // FETCH_SIZE 1.87 GB for the 0.54 GB input of 64 -> 64 at 512^2 (B = 16), the launch moving 4.8 TB/s through the fabric -- bound by
  // its own over-fetch (profiles/r04_conv_traffic_pmc_c3_bf16act.json).
  int t_begin, t_end, t_step;
  {
    const int GT = gridDim.x, wgid = blockIdx.x;
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int g = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    if (p.dbg & 0x10000) {   // (tuning: the contiguous ranges)
      t_begin = (int)((int64_t)g * ntiles / GT);
      t_end = (int)((int64_t)(g + 1) * ntiles / GT);
      t_step = 1;
    } else {
      t_begin = g;
      t_end = ntiles;
      t_step = GT;
    }
  }
  if (t_begin >= t_end) return;
  struct Tile { int b, oy0, x0, co0, ry; };   // oy0: first row of the tile in the row-residue sub-image ry
  auto decode = [&](int lid) {
    Tile t;
    const int ct = lid % co_tiles; lid /= co_tiles;
    const int tx = lid % tiles_x; lid /= tiles_x;
    const int ty = lid % tiles_y; lid /= tiles_y;
    t.ry = lid % DIL;
    t.b = lid / DIL; t.oy0 = ty * TH; t.x0 = tx * RV_TW; t.co0 = ct * CO_T;
    return t;
  };
  const int chw = p.H * p.W;
  const int nchunk = p.Cin / RV_CK;
  const int Cout = p.cout_g;     // channels of the group (a multiple of 8); weights are padded to rv_copad rows
  const int cbase = p.rv_cbase;  // first channel of the group in the layer's per-channel operands and in y

  // ---- patch staging: slot e of wave w = channel 2w + e / SPC of the chunk, 16-byte segment (e % SPC) * 64 + lane of its PR x 10.
  // State of the tile being STAGED (from the moment its first loads are issued, i.e. before the previous tile's epilogue):
  const bool affine = p.bf_isc_s != 0 || p.bf_ish_s != 0;   // (uniform) absent: the patch is a plain copy
  const bool shifted = p.bf_ish_s != 0;
  typedef const float __attribute__((address_space(4))) * cfp4;   // uniform loads through the scalar cache
  const float slope_c = ((cfp4)(uintptr_t)p.s2p)[0];
  const bool fast = p.s1 == 1.f && p.g1 == 1.f && p.s2s == 0 && slope_c >= 0.f && slope_c <= 1.f;   // (uniform) the lean epilogue below
  int poff[PT];
  unsigned pwr = 0, pin = 0;
  int st_xoff = 0, st_co0 = 0, st_b = 0;   // byte offset of the image in x, first channel of the tile, image
  const float* iscp = p.in_scale;
  const int pdst0 = 2 * wave * PCH + lane * 16;   // d = 1: slot e lands at + (e / SPC) PCH + (e % SPC) 1024: a patch row is exactly its ten segments
  int pdstv[DIL > 1 ? PT : 1];                    // d > 1: first byte of the segment's entries in residue sub-row 0
#pragma unroll
  for (int e = 0; e < PT; ++e) {
    const int tt = lane + 64 * (e % SPC);
    pwr |= tt < PR * SEGS ? (1u << e) : 0u;
    if constexpr (DIL > 1) {
      const int row = tt / SEGS, seg = tt - row * SEGS;
      pdstv[e] = ((2 * wave + e / SPC) * PR + row) * LROW + seg * (16 / DIL);
    }
  }
  auto stage_tile = [&](const Tile& t) {
    pin = 0;
#pragma unroll
    for (int e = 0; e < PT; ++e) {
      const int chl = e / SPC, tt = lane + 64 * (e % SPC);
      const int row = tt / SEGS, seg = tt - row * SEGS;
      const int sy = t.oy0 - 1 + row, iy = sy * DIL + t.ry, ix = t.x0 - 8 + 8 * seg;
      const bool in = tt < PR * SEGS && sy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      poff[e] = in ? (chl * chw + iy * p.W + ix) * 2 : 0x7fffffff;   // (past the buffer: the load returns zeros)
      pin |= in ? (1u << e) : 0u;
    }
    st_xoff = t.b * p.x_ch * chw * 2;
    st_co0 = t.co0;
    iscp = p.in_scale + (int64_t)t.b * p.in_scale_bstride;
    st_b = t.b;
  };
  float4 epv;
  float2 scv[RV_MAXC / 256];
  auto ep_load = [&]() {   // epilogue operands of the staged tile's channels (threads 0 .. CO_T - 1), written to LDS by first_commit
    if (affine) {
#pragma unroll
      for (int k = 0; k < RV_MAXC / 256; ++k) {
        const int ci = tid + 256 * k;
        scv[k] = ci < p.Cin ? float2{iscp[ci * p.bf_isc_s], p.in_shift[ci * p.bf_ish_s]} : float2{0.f, 0.f};
      }
    }
    if (tid < CO_T) {
      const int co = cbase + min(st_co0 + tid, Cout - 1);   // (channels past the group: clamped, never stored)
      const float os = p.osp[((int64_t)st_b * p.rv_ctot + co) * p.oss] * p.csp[co * p.css];
      const float cb = p.cbp[co * p.cbs] + p.b1p[co * p.b1s];
      const float b2 = p.b2p[co * p.b2s];
      epv = fast ? float4{os * p.g2, (cb + b2) * p.g2, 0.f, 0.f} : float4{os, cb, b2, p.s2p[co * p.s2s]};
    }
  };
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7ffffff0, 0x00020000);
  u32x4 pregA[PT], pregB[PT];
  auto issue_p = [&](u32x4 (&pr)[PT], int c) {
#ifdef VSP_BF16_ABLATE  // tuning build only (VSP_CONV_DBG): 1 no patch loads after the first chunk, 2 no weight DMA, 4 no stores, 8 no MFMAs, 16 no commits, 32 no B fragment reads, 64 no epilogue, 128 no A fragment reads
    if ((p.dbg & 1) && c > 0) return;
#endif
    const int soff = st_xoff + (c * RV_CK + 2 * wave) * chw * 2;
#pragma unroll
    for (int e = 0; e < PT; ++e) pr[e] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, poff[e], soff, 0);
  };
  // the per-channel input scale / shift of the staged image: fetched ONCE per tile with the first loads and kept in LDS (a scalar
  // load per chunk missed the scalar cache every other chunk -- a memory round trip in front of every commit)
  int sc_par = 0;
  auto load_scales = [&](int c, float (&sc)[2], float (&sh)[2]) {
    const float2* src = Sc + sc_par * RV_MAXC + c * RV_CK + 2 * wave;
    const float2 v0 = src[0], v1 = src[1];
    sc[0] = v0.x; sh[0] = v0.y; sc[1] = v1.x; sh[1] = v1.y;
  };
  auto commit_one = [&](unsigned char* Pdst, const u32x4 (&pr)[PT], const float (&sc)[2], const float (&sh)[2], int e) {
    if (!((pwr >> e) & 1u)) return;
#ifdef VSP_BF16_ABLATE
    if (p.dbg & 16) return;
#endif
    u32x4 v = pr[e];
    if (affine) {
      const float s = sc[e / SPC], h = sh[e / SPC];
      const bool in = (pin >> e) & 1u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = fmaf(vsp::bf16_lo(v[k]), s, h), hi = fmaf(vsp::bf16_hi(v[k]), s, h);
        const unsigned w = vsp::bf16_pack(lo, hi);
        v[k] = (!shifted || in) ? w : 0u;   // (a shift must not leak into the zero padding)
      }
    }
    if constexpr (DIL == 1) {
      *reinterpret_cast<u32x4*>(Pdst + pdst0 + (e / SPC) * PCH + (e % SPC) * 1024) = v;
    } else {   // de-interleave the eight pixels into the d residue sub-rows
      unsigned char* dst = Pdst + pdstv[e];
      constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;   // v_perm_b32 selectors: the low / the high halves of (S1, S0)
      if constexpr (DIL == 2) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_amdgcn_perm(v[1], v[0], LO), __builtin_amdgcn_perm(v[3], v[2], LO)};
        *reinterpret_cast<u32x2*>(dst + SP) = u32x2{__builtin_amdgcn_perm(v[1], v[0], HI), __builtin_amdgcn_perm(v[3], v[2], HI)};
      } else if constexpr (DIL == 4) {
        *reinterpret_cast<unsigned*>(dst) = __builtin_amdgcn_perm(v[2], v[0], LO);
        *reinterpret_cast<unsigned*>(dst + SP) = __builtin_amdgcn_perm(v[2], v[0], HI);
        *reinterpret_cast<unsigned*>(dst + 2 * SP) = __builtin_amdgcn_perm(v[3], v[1], LO);
        *reinterpret_cast<unsigned*>(dst + 3 * SP) = __builtin_amdgcn_perm(v[3], v[1], HI);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<unsigned short*>(dst + (2 * k) * SP) = (unsigned short)(v[k] & 0xffffu);
          *reinterpret_cast<unsigned short*>(dst + (2 * k + 1) * SP) = (unsigned short)(v[k] >> 16);
        }
      }
    }
  };

  // ---- weight slab by LDS-DMA: 12 rows of CO_T 16-byte fragments, 64 per wave instruction
  constexpr int NDMA = WROWS * CO_T / 64;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w);
  auto issue_w = [&](unsigned char* Wdst, int c) {
#ifdef VSP_BF16_ABLATE
    if ((p.dbg & 2) && c > 0) return;
#endif
    const u32x4* src = wsrc + (int64_t)c * WROWS * p.rv_copad;
#pragma unroll
    for (int k = 0; k < (NDMA + 3) / 4; ++k) {
      const int i = wave + 4 * k;
      const int L = i * 64 + lane;
      const int row = L / CO_T, co = L - row * CO_T;
      if (i < NDMA) __builtin_amdgcn_global_load_lds(src + row * p.rv_copad + st_co0 + co, reinterpret_cast<u32x4*>(Wdst) + i * 64, 16, 0, 0);
    }
  };

  // ---- fragments
  const int b_lane = kh * 2 * PCH + 12 + 4 * l32;       // byte of pixel pair l32 - 1 (pixels x0 + 2 l32 - 2, - 1) in row 0 of the half's first channel
  const int a_lane = (kh * CO_T + l32) * 16;
  const int wr0 = wave * RWV;                            // first output row of the wave inside the tile
  int boffd[DIL > 1 ? 2 : 1], shd[DIL > 1 ? 2 : 1];      // d > 1: the lane's window in its residue sub-row, per pixel parity: first aligned dword, funnel shift
  if constexpr (DIL > 1) {
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int c = 8 + 2 * l32 + pp;                    // column of the pixel in the patch row
      const int bo = (c % DIL) * SP + 2 * (c / DIL - 1);
      boffd[pp] = kh * 2 * PCH + (bo & ~3);
      shd[pp] = (bo & 2) ? 16 : 0;
    }
  }

  f32x16 acc[MB][RWV][2];

  auto interval = [&](int c, u32x4 (&prLoad)[PT], u32x4 (&prCommit)[PT]) {
    const int cur = c & 1, nxt = cur ^ 1;
    float sc[2], sh[2];
    if (affine) load_scales(c + 1 < nchunk ? c + 1 : c, sc, sh);
    if (c + 2 < nchunk) issue_p(prLoad, c + 2);
    const unsigned char* Wc = Wl + cur * WBUF + a_lane;
    const unsigned char* Pc = Pl + cur * PBUF + (DIL == 1 ? b_lane : 0) + wr0 * LROW;
    unsigned char* Pn = Pl + nxt * PBUF;
    // B fragments of one (quad, input row) group: per channel THREE aligned dwords E0 E1 E2 = pixel pairs j - 1, j, j + 1 of the row
    // (ds_read_b32, lanes 4 bytes apart: 256 bytes per instruction and half); the odd pixels' window [2j .. 2j+3] is (E1, E2), the
    // even pixels' window [2j-1 .. 2j+2] two funnel shifts -- 12 bytes of LDS per lane and channel for both parities (two
    // overlapping ds_read_b64 per parity measured 10 clk each: with eight waves per CU the LDS, not the matrix pipe, was the bound)
    auto load_b = [&](int grp, bf16x8 (&bf)[2]) {
#ifdef VSP_BF16_ABLATE
      if (p.dbg & 32) return;
#endif
      const int q = grp / (RWV + 2), ir = grp - q * (RWV + 2);
      if constexpr (DIL == 1) {
        unsigned e[2][3];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int k = 0; k < 3; ++k) e[ch][k] = *reinterpret_cast<const unsigned*>(Pc + (4 * q + ch) * PCH + ir * LROW + 4 * k);
        bf[0] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(e[0][1], e[0][0], 16), __builtin_amdgcn_alignbit(e[0][2], e[0][1], 16),
                                                 __builtin_amdgcn_alignbit(e[1][1], e[1][0], 16), __builtin_amdgcn_alignbit(e[1][2], e[1][1], 16)});
        bf[1] = __builtin_bit_cast(bf16x8, u32x4{e[0][1], e[0][2], e[1][1], e[1][2]});
      } else {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          unsigned e[2][3];
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int k = 0; k < 3; ++k) e[ch][k] = *reinterpret_cast<const unsigned*>(Pc + boffd[pp] + (4 * q + ch) * PCH + ir * LROW + 4 * k);
          bf[pp] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(e[0][1], e[0][0], shd[pp]), __builtin_amdgcn_alignbit(e[0][2], e[0][1], shd[pp]),
                                                    __builtin_amdgcn_alignbit(e[1][1], e[1][0], shd[pp]), __builtin_amdgcn_alignbit(e[1][2], e[1][1], shd[pp])});
        }
      }
    };
    bf16x8 a[3][MB], bq[2][2];
    load_b(0, bq[0]);
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp) {
      const int q = grp / (RWV + 2), ir = grp - q * (RWV + 2);
#ifdef VSP_BF16_ABLATE
      if (!(p.dbg & 128))
#endif
      if (ir == 0) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            a[dy][mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wc + ((dy * 2 + q) * 2 * CO_T + mb * 32) * 16));
      }
      const int cs = grp & 1;
      if (grp + 1 < NGRP) load_b(grp + 1, bq[cs ^ 1]);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int r = ir - dy;
          if (r < 0 || r >= RWV) continue;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
#ifdef VSP_BF16_ABLATE
            if (p.dbg & 8) continue;
#endif
            acc[mb][r][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy][mb], bq[cs][pp], acc[mb][r][pp], 0, 0, 0);
          }
        }
      // (Measured alternative: all commits + the weight DMA at the TOP of the interval, the patch loads of chunk c + 2 behind them and
      //  `s_waitcnt vmcnt(PT)` + a bare s_barrier at the end, so that the prefetch flies across the barrier instead of being drained by
      //  the vmcnt(0) the DMA needs: correct, and 520 against 514 us -- the drain is not what the 150 us of the patch loads are.)
      if (c + 1 < nchunk) {
        if (grp < PT) commit_one(Pn, prCommit, sc, sh, grp);
        if (grp == PT - 1) issue_w(Wl + nxt * WBUF, c + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  auto first_loads = [&]() {
    issue_p(pregA, 0);
    if (nchunk > 1) issue_p(pregB, 1);
    ep_load();
    issue_w(Wl, 0);
  };
  auto first_commit = [&](int par) {
    sc_par = par;
    if (affine) {
#pragma unroll
      for (int k = 0; k < RV_MAXC / 256; ++k)
        if (tid + 256 * k < p.Cin) Sc[par * RV_MAXC + tid + 256 * k] = scv[k];
    }
    if (tid < CO_T) Ep[par * CO_T + tid] = epv;
    if (affine) __syncthreads();   // (the first chunk's own scales)
    float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
    if (affine) load_scales(0, sc, sh);
#pragma unroll
    for (int e = 0; e < PT; ++e) commit_one(Pl, pregA, sc, sh, e);
    __syncthreads();
  };

  // ---- epilogue operands that do not depend on the tile
  const int y_plane = p.y_h * p.y_w;
  const bool has_nz = p.nzs != 0, has_r1 = p.r1s != 0, has_r2 = p.r2s != 0;
  const bool act1 = !(p.s1 == 1.f && p.g1 == 1.f);
  const float s1 = p.s1, g1 = p.g1, g2 = p.g2, nw = ((cfp4)(uintptr_t)p.nwp)[0];
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.r1p), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.r2p), 0, 0x7ffffff0, 0x00020000);

  Tile cur = decode(t_begin);
  int par = 0;
  stage_tile(cur);
  first_loads();
  first_commit(par);
  for (int t = t_begin; t < t_end; t += t_step) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < RWV; ++r)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mb][r][pp][i] = 0.f;
    // the noise of the wave's rows: requested before the chunk loop, long landed when the epilogue reads it
    f32x2 nzv[RWV];
    int voff[RWV];
#pragma unroll
    for (int r = 0; r < RWV; ++r) {
      const int oy = (cur.oy0 + wr0 + r) * DIL + cur.ry;
      const bool ok = oy < p.OH;
      nzv[r] = f32x2{0.f, 0.f};
      if (has_nz && ok) nzv[r] = *reinterpret_cast<const f32x2*>(p.nzp + (int64_t)cur.b * p.OH * p.OW + oy * p.OW + cur.x0 + 2 * l32);
      voff[r] = ok ? ((kh * 4 * y_plane) + oy * p.y_w + cur.x0 + 2 * l32) * 2 : 0x7fffffff;   // (rows past the image: dropped by the range check)
    }
    for (int c = 0; c < nchunk; c += 2) {
      interval(c, pregA, pregB);
      if (c + 1 < nchunk) interval(c + 1, pregB, pregA);
    }
    const bool has_next = t + t_step < t_end;
    const Tile done = cur;
    if (has_next) {   // (every wave is past the last barrier of the chunk loop: both buffer pairs are free)
      cur = decode(t + t_step);
      stage_tile(cur);
      first_loads();
    }
    // ---- epilogue: lane = pixel pair (x0 + 2 l32, +1) of the wave's rows, channels 8 (i >> 2) + 4 kh + (i & 3) of every 32-block
    const int y_img = (done.b * p.y_ch + p.y_coff + cbase) * y_plane * 2, r_img = (done.b * p.res_ch + p.res_coff + cbase) * y_plane * 2;
#ifdef VSP_BF16_ABLATE
    if (!(p.dbg & 64))
#endif
    if (fast) {
      // StyledConv / plain-conv flavour (no first activation, constant slope in [0, 1]): gain folded into the per-channel pair
      // (scale, bias) and the noise row, leaky ReLU = max(v, slope v): ~9 vector instructions per pixel pair and one 8-byte LDS
      // read per channel, four channels requested at a time.  Plain floats on purpose: written on float2 vectors (v_pk_fma_f32
      // with op_sel broadcasts of the freshly read pair) the even pixel of the last 16 lanes lost its bias term on a few row
      // segments per launch whenever two workgroups shared a CU (never with one) -- see DESIGN.md, round-4 findings.
      f32x2 nzg[RWV];
#pragma unroll
      for (int r = 0; r < RWV; ++r) nzg[r] = nzv[r] * (nw * g2);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          f32x2 ab[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {   // (read through the type it was written with: a float4 store read back as a float2 is undefined under TBAA)
            const float4 t4 = Ep[par * CO_T + mb * 32 + 8 * ib + k + 4 * kh];
            ab[k] = f32x2{t4.x, t4.y};
          }
          if (done.co0 + mb * 32 + 8 * ib >= Cout) continue;   // (uniform: channel octets past the group; Cout is a multiple of 8)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = 4 * ib + k;
            const int cplane = (done.co0 + mb * 32 + 8 * ib + k) * y_plane * 2;
#pragma unroll
            for (int r = 0; r < RWV; ++r) {
              float v[2] = {fmaf(acc[mb][r][0][i], ab[k][0], ab[k][1]), fmaf(acc[mb][r][1][i], ab[k][0], ab[k][1])};
              if (has_nz) { v[0] += nzg[r][0]; v[1] += nzg[r][1]; }
              v[0] = fmaxf(v[0], v[0] * slope_c);
              v[1] = fmaxf(v[1], v[1] * slope_c);
              if (has_r1) {
                const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r1rsrc, voff[r], r_img + cplane, 0);
                v[0] += vsp::bf16_lo(w); v[1] += vsp::bf16_hi(w);
              }
              if (has_r2) {
                const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r2rsrc, voff[r], r_img + cplane, 0);
                v[0] += vsp::bf16_lo(w); v[1] += vsp::bf16_hi(w);
              }
#ifdef VSP_BF16_ABLATE
              if ((p.dbg & 4) && v[0] != 12345.f) continue;
#endif
              __builtin_amdgcn_raw_buffer_store_b32(vsp::bf16_pack(v[0], v[1]), yrsrc, voff[r], y_img + cplane, 0);
            }
          }
        }
      }
    } else
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int col = mb * 32 + 8 * (i >> 2) + (i & 3);        // + 4 kh: the lane half
        if (done.co0 + mb * 32 + 8 * (i >> 2) >= Cout) continue;  // (uniform)
        const float4 e4 = Ep[par * CO_T + col + 4 * kh];
        const float os = e4.x, cb = e4.y, b2 = e4.z, pg = g2, ng = e4.w * g2;
        const int cplane = (done.co0 + col) * y_plane * 2;
#pragma unroll
        for (int r = 0; r < RWV; ++r) {
          float v[2] = {fmaf(acc[mb][r][0][i], os, cb), fmaf(acc[mb][r][1][i], os, cb)};
          if (act1) {
            v[0] = (v[0] > 0.f ? v[0] : v[0] * s1) * g1;
            v[1] = (v[1] > 0.f ? v[1] : v[1] * s1) * g1;
          }
          v[0] += nzv[r][0] * nw + b2;
          v[1] += nzv[r][1] * nw + b2;
          v[0] *= v[0] > 0.f ? pg : ng;
          v[1] *= v[1] > 0.f ? pg : ng;
          if (has_r1) {
            const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r1rsrc, voff[r], r_img + cplane, 0);
            v[0] += vsp::bf16_lo(w); v[1] += vsp::bf16_hi(w);
          }
          if (has_r2) {
            const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r2rsrc, voff[r], r_img + cplane, 0);
            v[0] += vsp::bf16_lo(w); v[1] += vsp::bf16_hi(w);
          }
#ifdef VSP_BF16_ABLATE
          if ((p.dbg & 4) && v[0] != 12345.f) continue;
#endif
          __builtin_amdgcn_raw_buffer_store_b32(vsp::bf16_pack(v[0], v[1]), yrsrc, voff[r], y_img + cplane, 0);
        }
      }
    }
    if (has_next) {
      par ^= 1;
      first_commit(par);
    }
  }
}

template <int MB, int RWV, int DIL = 1>
int launch_rv(ConvK q, hipStream_t stream) {
  constexpr int TH = 4 * RWV, PR = TH + 2, CO_T = 32 * MB;
  constexpr size_t lds = 2 * (size_t)(12 * CO_T * 16) + 2 * (size_t)(RV_CK * PR * (DIL == 8 ? 192 : RV_ROWB)) + 2 * CO_T * 16 + 2 * RV_MAXC * 8;
  static vsp::LdsAttrOnce attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_bf16_rv_kernel<MB, RWV, DIL>), 150 * 1024, "conv2d_bf16rv")) return rc;
  q.co_tiles = (q.cout_g + CO_T - 1) / CO_T;
  const int64_t ntiles = (int64_t)(q.W / RV_TW) * (((q.H + DIL - 1) / DIL + TH - 1) / TH) * DIL * q.co_tiles * q.B;
  if (ntiles > 0x7fffffff) return vsp::fail(VSP_EINVAL, "conv2d_bf16rv: grid too large");
  static const int per_cu = vsp::tune_env("VSP_BF16RV_WGS") ? atoi(vsp::tune_env("VSP_BF16RV_WGS")) : 2;   // resident workgroups per CU (tuning)
  const int64_t grid = ntiles < (int64_t)vsp::kNumCU * per_cu ? ntiles : (int64_t)vsp::kNumCU * per_cu;
  conv_bf16_rv_kernel<MB, RWV, DIL><<<dim3((unsigned)grid), 256, lds, stream>>>(q, (int)ntiles);
  return VSP_OK;
}

}  // namespace

// What the row-vector kernel serves: bf16 activations, 3x3, stride 1, G = 1 or 2-4 DILATION GROUPS over one input (dilations from
// {1, 2, 4, 8}, padding = dilation), Cin % 8 == 0 (<= 256), channels per group % 32 == 0 (plain layers) / % 8 == 0 (groups), W % 64 == 0,
// 16-byte aligned image rows, tensors below 2 GiB (32-bit byte offsets from the tensor base).
bool bf16rv_eligible(const ConvK& q) {
  if (!q.io_bf16 || q.KH != 3 || q.KW != 3 || q.G < 1 || q.G > 4 || (q.G > 1 && q.x_gs != 0)) return false;
  for (int g = 0; g < q.G; ++g)
    if (q.dil[g] != 1 && (q.G == 1 || (q.dil[g] != 2 && q.dil[g] != 4 && q.dil[g] != 8))) return false;
  if (q.Cin % RV_CK || q.Cin > RV_MAXC || q.cout_g % (q.G == 1 ? 32 : 8) || q.W % RV_TW || q.OH != q.H || q.OW != q.W) return false;
  if ((reinterpret_cast<uintptr_t>(q.x) & 15) || (reinterpret_cast<uintptr_t>(q.y) & 3)) return false;
  if ((q.r1s && (reinterpret_cast<uintptr_t>(q.r1p) & 3)) || (q.r2s && (reinterpret_cast<uintptr_t>(q.r2p) & 3))) return false;
  if (q.nzs && (reinterpret_cast<uintptr_t>(q.nzp) & 7)) return false;
  const int64_t lim = ((int64_t)1 << 31) - 64;   // 32-bit byte offsets from the tensor base; the padding marker 0x7fffffff lies above them
  const int64_t plane = (int64_t)q.y_h * q.y_w;
  if ((q.y_w & 1) || (int64_t)q.B * q.x_ch * q.H * q.W * 2 >= lim || (int64_t)q.B * q.y_ch * plane * 2 >= lim || (int64_t)q.B * q.res_ch * plane * 2 >= lim) return false;
  return true;
}

// variant (plain layers): 0 = automatic (64-channel tiles when Cout allows), 1 = 32 channels x 8 rows, 2 = 64 channels x 8 rows,
// 3 = 32 channels x 16 rows (x 64 pixels).  Dilation groups: one launch per group on the 32-channel x 8-row tile.
int bf16rv_launch(const ConvK& q0, int variant, hipStream_t stream) {
  ConvK q = q0;
  q.rv_copad = (q.cout_g + 31) & ~31;
  q.rv_ctot = q.G * q.cout_g;
  q.rv_cbase = 0;
  if (q.G == 1) {
    if (variant == 0) variant = q.cout_g % 64 == 0 ? 2 : 1;
    if (variant == 2 && q.cout_g % 64) return vsp::fail(VSP_EINVAL, "conv2d_bf16rv: 64-channel tiles need Cout %% 64 == 0");
    switch (variant) {
      case 1: return launch_rv<1, 2>(q, stream);
      case 2: return launch_rv<2, 2>(q, stream);
      case 3: return launch_rv<1, 4>(q, stream);
      default: return vsp::fail(VSP_EINVAL, "conv2d_bf16rv: unknown variant %d", variant);
    }
  }
  const int64_t wgroup = (int64_t)(q.Cin / RV_CK) * 12 * q.rv_copad * 8;   // bf16 elements of one group's weights
  const vsp::bf16_t* w0 = reinterpret_cast<const vsp::bf16_t*>(q0.w);
  const int G = q.G;
  q.G = 1;
  for (int g = 0; g < G; ++g) {
    q.rv_cbase = g * q.cout_g;
    q.w = reinterpret_cast<const float*>(w0 + g * wgroup);
    int rc;
    switch (q0.dil[g]) {
      case 1: rc = launch_rv<1, 2, 1>(q, stream); break;
      case 2: rc = launch_rv<1, 2, 2>(q, stream); break;
      case 4: rc = launch_rv<1, 2, 4>(q, stream); break;
      default: rc = launch_rv<1, 2, 8>(q, stream); break;
    }
    if (rc) return rc;
  }
  return VSP_OK;
}

}  // namespace vspconv

// Direct 3x3 convolution on fp32 MFMA as ONE software pipeline per workgroup (round 3): the layers that have no Winograd
// form -- stride-2 convolutions (the e4e style-head stem 512 -> 5632, StyledConv_down, IR-SE down-convs), the stride-2
// transposed up-convs and the four dilated SMART branches of narrow layers.
//
// conv_kernel.h stages a chunk, crosses a barrier, multiplies, crosses a barrier: the staging of a workgroup is exposed
// unless a second resident workgroup happens to be in its MFMA phase (measured: MFMA phase alone 131 TFLOP/s, whole
// kernel 88-104 on these layers).  Here the slab / patch LDS buffers are DOUBLE, the loads of chunk i+2 and the LDS
// commit of chunk i+1 are pieces of the instruction stream BETWEEN the MFMAs of chunk i (an in-order wave overlaps with
// its own MFMAs only what sits between them in program order: DESIGN 4, Winograd findings), and one barrier per chunk
// is all the synchronisation left:
//
//     top of interval i:   LDS[i & 1] = chunk i   |  registers = chunk i+1 (loads in flight)
//     units 0 .. NUC-1 :   MFMAs of chunk i  +  slices of  registers -> LDS[(i+1) & 1]
//     units NUC .. NU-1:   MFMAs of chunk i  +  slices of  loads(chunk i+2) -> registers
//     barrier
//
// (unit = one tap x one 4-channel k-step = MB*NB MFMAs; the A / B fragments of unit u+1 are read while unit u multiplies).
// Operand contract, tile geometry, XCD-aware work order and the epilogue chain are those of conv_igemm_kernel; the
// per-channel input scale comes as pointer + strides (absent -> a device constant with stride 0).  Not served (the plan
// falls back to conv_igemm_kernel): an input shift, Cin not a multiple of the chunk, scalar weight rows, kernels other than 3x3.
#include "conv_kernel.h"
#include <type_traits>

#ifndef VSP_PIPE_SBU
#define VSP_PIPE_SBU 1
#endif
#ifndef VSP_PIPE_SBU_T
#define VSP_PIPE_SBU_T 1
#endif

namespace vspconv {

namespace {

__device__ __forceinline__ float uload(const float* base, int idx) {  // wave-uniform operand through the scalar cache
  typedef const float __attribute__((address_space(4))) * cfp4;
  return ((cfp4)(uintptr_t)base)[__builtin_amdgcn_readfirstlane(idx)];
}

// MODE 0: plain / grouped conv (stride 1 or 2, per-group dilation); 1: stride-2 transposed 3x3 (four sub-pixel phases);
// 2: the four dilation groups of a SMART layer from one shared patch (M-block = group)
//
// FG = true (round 3, "fixed geometry", dilation-group mode): the tile is 16 x 16 pixels and the dilations are 1, 2, 4, 8, so the patch
// row pitch, the plane pitch and the size of an LDS buffer are constants and the interval is instantiated per buffer PARITY: every
// fragment read is ONE base register + an immediate.  The generic kernel pays a v_add per B read (12 VALU + 12 LDS instructions per 8
// MFMAs on the dilation groups: PMC 65 % MFMA busy): 64 -> 4 x 16 at 512^2 1550 -> 1406 us (110 TFLOP/s).  The same treatment of the
// transposed mode (main tiles only, strips as a second launch; unit = k-step with all nine taps sharing 2 x (NP + 1) shifted B
// fragments: 15 reads per 18 MFMAs instead of 27) was built and measured: 678 vs 669 us at 512 -> 256 / 64^2 -- no gain, removed.
template <int MB, int NB, int WM, int WN, int CK, int PROWS, int OCC, int MODE, bool FG = false>
__global__ __launch_bounds__(64 * WM * WN, OCC) void conv_pipe_kernel(const ConvK p) {
  // KG (MODE 3): the DATA GRADIENT of the four dilation groups in one pass -- the staging, weight image and channel mapping of the
  // dilation-group mode (four blocks of 16 output channels over one shared 32-halo patch), but the dilation belongs to the INPUT
  // channel quarter a chunk lies in and all four M-blocks multiply the same B fragments: out = sum_q conv(x[q], W[q], dil[q]).
  constexpr bool TC = MODE == 1, DG = MODE == 2, KG = MODE == 3, DGS = DG || KG;
  static_assert(!FG || DG, "fixed geometry serves the dilation-group mode");
  static_assert(!TC || NB % 4 == 0, "transposed mode: N-blocks come in groups of four sub-pixel phases");
  static_assert(!DGS || (MB == 4 && WM == 1), "dilation-group modes: M-block = group / block of 16 output channels");
  constexpr int NP = TC ? NB / 4 : NB;  // 16-wide groups of patch positions per wave
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int CO_T = 16 * MB * WM;
  constexpr int WS = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;
  constexpr int T = 9, KS = CK / 4, NU = T * KS;
  // fixed geometry (FG): 16-wide tiles
  constexpr int F_NPOS = 16 * WN * NP;                     // positions / pixels per workgroup
  constexpr int F_TH = F_NPOS / 16;
  constexpr int F_PW = TC ? 17 : 32;                       // patch row: 16 + 1 (transposed) / 16 + 2 * 8 (dilation groups)
  constexpr int F_WPC = CK >= WM * WN ? 1 : WM * WN / CK;
  constexpr int F_PS = ((PROWS * F_WPC * 64) & ~31) + 16 >= PROWS * F_WPC * 64 ? ((PROWS * F_WPC * 64) & ~31) + 16
                                                                               : ((PROWS * F_WPC * 64) & ~31) + 48;   // = round_pitch(.., 0)
  constexpr int F_BUF = T * CK * WS + CK * F_PS;
  static_assert(!FG || !DG || F_TH == 16, "dilation-group fixed geometry: 16 x 16 pixels");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, kq = lane >> 4;

  // ---- work order and tile geometry (as conv_igemm_kernel)
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
  } else if (p.wg_order == 2) {
    // weight-heavy layers (the 512 -> 5632 stem: 104 MB of weights): a GROUP of p.wg_cgs channel tiles sweeps every pixel tile of the
    // batch before the next group starts -- the group's weights stay in the XCD's L2 and the (smaller) input is streamed once per
    // sweep, instead of every pixel tile streaming all the weights (round 3: FETCH_SIZE 6.8 GB for this one launch).
    const int GX = gridDim.x, GY = gridDim.y, GT = GX * GY * gridDim.z;
    const int wgid = blockIdx.x + GX * (blockIdx.y + GY * blockIdx.z);
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    const int cgs = p.wg_cgs, npt = GX * (int)gridDim.z;
    const int full = (GY / cgs) * cgs * npt;
    int cgrp, rem, cw;
    if (lid < full) { cgrp = lid / (cgs * npt); rem = lid - cgrp * (cgs * npt); cw = cgs; }
    else { cgrp = GY / cgs; rem = lid - full; cw = GY - cgrp * cgs; }
    const int pt = rem / cw;
    by = cgrp * cgs + (rem - pt * cw);
    bz = pt / GX;
    tile = pt - bz * GX;
  }
  const int tx_i = tile % p.tiles_x, ty_i = tile / p.tiles_x;
  const int g = DGS ? 0 : by / p.co_tiles;
  const int co0 = DGS ? by * 16 : (by % p.co_tiles) * CO_T;  // within the group
  const int b = bz;
  int twl = FG ? 4 : p.tw_log2, TH = FG ? F_TH : p.th, oy0 = ty_i * (FG ? F_TH : p.th), ox0 = tx_i << (FG ? 4 : p.tw_log2);
  int mlim = p.H + 1;  // transposed: first invalid position row of this block
  if (TC && !FG && p.strip_col >= 0) {  // edge strips of the (H+1) x (W+1) position grid (see conv_igemm_kernel)
    constexpr int NPIXB = 16 * WN * NP;
    mlim = p.H;
    int j = tile - p.tiles_x * p.tiles_y;
    if (j >= p.strip_col) {
      j -= p.strip_col;
      twl = __builtin_ctz(NPIXB); TH = 1; oy0 = p.H; ox0 = j * NPIXB; mlim = p.H + 1;
    } else if (j >= 0) {
      twl = 0; TH = NPIXB; oy0 = j * NPIXB; ox0 = p.W;
    }
  }
  const int TW = 1 << twl;
  const int gi = p.G > 4 ? 0 : g;
  const int D = DGS ? max(max(p.dil[0], p.dil[1]), max(p.dil[2], p.dil[3])) : p.dil[gi];
  const int PH = TC ? TH + 1 : (TH - 1) * p.sy + 2 * D + 1;
  const int PW = FG ? F_PW : TC ? TW + 1 : (TW - 1) * p.sx + 2 * D + 1;
  const int plane = PH * PW;
  const int PS = FG ? F_PS : p.bf_plane;            // plane pitch (host: >= the 64-word rows the waves stage, == 16 mod 32 or odd)
  const int BUF = FG ? F_BUF : T * CK * WS + CK * PS;   // floats per LDS buffer
  const int iy0 = TC ? oy0 - 1 : DGS ? oy0 - D : oy0 * p.sy - p.pady[gi];
  const int ix0 = TC ? ox0 - 1 : DGS ? ox0 - D : ox0 * p.sx - p.padx[gi];

  // per-lane fragment offsets
  int pixoff[NP];
#pragma unroll
  for (int nb = 0; nb < NP; ++nb) {
    const int n = (wn * NP + nb) * 16 + lr;
    const int py = n >> twl, px = n & (TW - 1);
    pixoff[nb] = T * CK * WS + py * p.sy * PW + px * p.sx + kq * PS;
  }
  const int a_lane = kq * WS + wm * MB * 16 + lr;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int chw = p.H * p.W;
  const float* xb = p.x + ((int64_t)b * p.x_ch + (int64_t)g * p.x_gs) * chw;
  const float* wg = p.w + (int64_t)g * T * p.Cin * p.cout_g;
  // Buffer resources with the REAL extents: a lane offset beyond them reads as zero (hardware range check), which is how words
  // outside the image and weight columns beyond cout_g become zeros -- no select, no branch in the commit.  (The scalar offset
  // selects the channel plane / chunk; kOOB exceeds every extent by itself, so the zero does not depend on whether the range check
  // includes the scalar part.)
  constexpr int kOOB = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (p.x_ch - g * p.x_gs) * chw * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, (DGS ? 4 : 1) * T * p.Cin * p.cout_g * 4, 0x00020000);
  const unsigned pw_magic = (unsigned)(((1ull << 32) + PW - 1) / PW);  // exact floor(i / PW) for i < 2^16, PW <= 2^8

  // ---- staging plan (chunk-invariant).  Patch: a wave owns whole 64-word rows of a channel plane -- PCH channels per wave
  //      (CK >= NW) or WPC waves per channel -- so the channel is wave-uniform and the LDS address is base + immediate.
  constexpr int PCH = CK >= NW ? CK / NW : 1;
  constexpr int WPC = CK >= NW ? 1 : NW / CK;
  const int pcl0 = CK >= NW ? wave : wave / WPC;
  const int j0 = CK >= NW ? 0 : wave % WPC;
  int poff[PROWS];   // byte offset of plane word (j0 + WPC e) * 64 + lane inside the channel image (kOOB: outside -> reads 0)
#pragma unroll
  for (int e = 0; e < PROWS; ++e) {
    const int i = (j0 + WPC * e) * 64 + lane;
    const int r = (int)__umulhi((unsigned)i, pw_magic);
    const int c = i - r * PW;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool in = i < plane && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    poff[e] = in ? (iy * p.W + ix) * 4 : kOOB;
  }
  const int pdst = T * CK * WS + pcl0 * PS + j0 * 64 + lane;   // + pc * NW * PS + e * WPC * 64
  // Weights: float4 items (tap, channel, 4 output channels), item i of the chunk = thread i % NT, pass i / NT; what is left after
  // the full passes goes out as RV floats per thread (9 taps make the item count odd: no ragged pass, no divergent store)
  constexpr int V = CO_T / 4;
  constexpr int WITEMS = T * CK * V;
  constexpr int WF = WITEMS / NT;                    // full passes
  constexpr int RV = ((WITEMS % NT) * 4 + NT - 1) / NT;   // floats per thread of the tail (0, 1 or 2); threads beyond it load zeros
  static_assert(RV <= 2, "weight tail must split into at most 2 floats per thread");   // (out of range) into a slab pad column
  static_assert(RV == 0 || (WS > CO_T && NT / 16 <= T * CK), "tail threads without a piece need the slab's pad columns");
  constexpr int WMAX = WF + (RV ? 1 : 0);
  int woff[WMAX];    // byte offset relative to the chunk's first input channel (kOOB: zero fill)
  int wdst[WMAX];    // LDS float index
#pragma unroll
  for (int w = 0; w < WMAX; ++w) {
    const int f = w < WF ? (tid + w * NT) * 4 : WF * NT * 4 + tid * RV;   // first float of this thread's piece
    const int i = f >> 2, sub = f & 3;
    const int row = i / V, c4 = i - row * V;
    const int tap = row / CK, cl = row - tap * CK;
    const bool piece = i < WITEMS;                   // (only the tail pass can run out of pieces)
    wdst[w] = piece ? row * WS + c4 * 4 + sub
                    : (RV == 2 ? ((tid >> 3) % (T * CK)) * WS + CO_T + (tid & 7) * 2 : (tid >> 4) * WS + CO_T + (tid & 15));
    if constexpr (DGS) {
      const int cc = co0 + (c4 & 3) * 4 + sub;
      woff[w] = (piece && cc < p.cout_g) ? (((c4 >> 2) * T * p.Cin + tap * p.Cin + cl) * p.cout_g + cc) * 4 : kOOB;
    } else {
      const int cc = co0 + c4 * 4 + sub;
      woff[w] = (piece && cc < p.cout_g) ? ((tap * p.Cin + cl) * p.cout_g + cc) * 4 : kOOB;
    }
  }
  f32x4 wreg[WMAX];
  float preg[PCH][PROWS];
  float psc[PCH];    // input scale of the channels being committed (scalar loads at the top of the interval)

  // one staging item = one weight piece or one patch word: k < WMAX weights, then the patch words (pc major)
  constexpr int NITEM = WMAX + PCH * PROWS;
  auto issue_item = [&](int k, int ci0) {  // load item k of the chunk starting at channel ci0 (raw value; consumed by commit_item)
    if (k < WF) {
      wreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[k], ci0 * p.cout_g * 4, 0));
    } else if (k < WMAX) {
      if constexpr (RV == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs, woff[k], ci0 * p.cout_g * 4, 0));
        wreg[k][0] = v[0]; wreg[k][1] = v[1];
      } else {
        wreg[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, woff[k], ci0 * p.cout_g * 4, 0));
      }
    } else {
      const int pc = (k - WMAX) / PROWS, e = (k - WMAX) % PROWS;
      const int ci = ci0 + pcl0 + pc * NW;
      preg[pc][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, poff[e], ci * chw * 4, 0));
    }
  };
  auto load_scales = [&](int ci0) {
#pragma unroll
    for (int pc = 0; pc < PCH; ++pc)
      psc[pc] = uload(p.wcp, b * p.wc_bs + (g * p.x_gs + ci0 + pcl0 + pc * NW) * p.wc_cs);
  };
  auto commit_item = [&](int k, float* buf) {
    if (k < WF) {
      *reinterpret_cast<f32x4*>(buf + wdst[k]) = wreg[k];
    } else if (k < WMAX) {
      if constexpr (RV == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(buf + wdst[k]) = f32x2{wreg[k][0], wreg[k][1]};
      } else {
        buf[wdst[k]] = wreg[k][0];
      }
    } else {
      const int pc = (k - WMAX) / PROWS, e = (k - WMAX) % PROWS;
      buf[pdst + pc * NW * PS + e * (WPC * 64)] = preg[pc][e] * psc[pc];
    }
  };

  // ---- one unit of MFMAs: tap = u / KS, k-step = u % KS.  Fragments are read one unit ahead into the other register set.
  float af[2][MB], bfr[2][NP];
  int kg_d = D, kg_b = 0;   // KG: dilation of the input-channel quarter of the current chunk, offset of its taps inside the halo-D patch
  auto load_frag = [&](int u, const float* buf, float (&a)[MB], float (&bq)[NP]) {
    const int tap = u / KS, c4 = u % KS;
    const int ky = tap / 3, kx = tap % 3;
    const float* wt = buf + (tap * CK + c4 * 4) * WS + a_lane;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a[mb] = wt[mb * 16];
    if constexpr (!DG) {
      const int boff = TC ? (1 - (ky >> 1)) * PW + (1 - (kx >> 1)) : KG ? kg_b + (ky * PW + kx) * kg_d : (ky * PW + kx) * D;
#pragma unroll
      for (int np = 0; np < NP; ++np) bq[np] = buf[c4 * 4 * PS + pixoff[np] + boff];
    }
  };
  auto mfma_unit = [&](int u, const float (&a)[MB], const float (&bq)[NP]) {
    const int tap = u / KS;
    if constexpr (TC) {
      const int ph = ((tap / 3) & 1) * 2 + ((tap % 3) & 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int np = 0; np < NP; ++np)
          acc[mb][np * 4 + ph] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], bq[np], acc[mb][np * 4 + ph], 0, 0, 0);
    } else {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], bq[nb], acc[mb][nb], 0, 0, 0);
    }
  };
  // dilation-group mode: every group (= M-block) has its own B fragments (tap offsets scale with the group's dilation)
  int gbase[4];
  if constexpr (DG) {
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) gbase[gg] = (D - p.dil[gg]) * (PW + 1);
  }
  float bdg[2][4][NB];
  auto load_frag_dg = [&](int u, const float* buf, float (&a)[MB], float (&bq)[4][NB]) {
    const int tap = u / KS, c4 = u % KS;
    const int ky = tap / 3, kx = tap % 3;
    const float* wt = buf + (tap * CK + c4 * 4) * WS + a_lane;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) a[gg] = wt[gg * 16];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int boff = gbase[gg] + (ky * PW + kx) * p.dil[gg];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[gg][nb] = buf[c4 * 4 * PS + pixoff[nb] + boff];
    }
  };
  auto mfma_unit_dg = [&](const float (&a)[MB], const float (&bq)[4][NB]) {
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[gg][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gg], bq[gg][nb], acc[gg][nb], 0, 0, 0);
  };

  // ---- pipeline
  const int nchunk = (p.Cin + CK - 1) / CK;
  // units that carry commit slices / issue slices
  constexpr int NUI = NU >= 6 ? NU / 3 : 1;       // the last NUI units issue the loads of chunk i+2
  constexpr int NUC = NU - NUI;                   // the first NUC units commit chunk i+1
  constexpr int SBU = TC ? VSP_PIPE_SBU_T : VSP_PIPE_SBU;   // units between scheduling fences
  auto interval = [&](int i, auto commit_tag, auto issue_tag) {
    constexpr bool COMMIT = decltype(commit_tag)::value, ISSUE = decltype(issue_tag)::value;
    const float* cur = smem + (i & 1) * BUF;
    float* nxt = smem + ((i + 1) & 1) * BUF;
    if constexpr (COMMIT) load_scales((i + 1) * CK);
    if constexpr (KG) {
      const int q4 = p.Cin >> 2, ci = i * CK;
      kg_d = ci < q4 ? p.dil[0] : ci < 2 * q4 ? p.dil[1] : ci < 3 * q4 ? p.dil[2] : p.dil[3];
      kg_b = (D - kg_d) * (PW + 1);
    }
    if constexpr (DG) load_frag_dg(0, cur, af[0], bdg[0]); else load_frag(0, cur, af[0], bfr[0]);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int s = u & 1;
      if (u + 1 < NU) {
        if constexpr (DG) load_frag_dg(u + 1, cur, af[s ^ 1], bdg[s ^ 1]); else load_frag(u + 1, cur, af[s ^ 1], bfr[s ^ 1]);
      }
      if constexpr (DG) mfma_unit_dg(af[s], bdg[s]); else mfma_unit(u, af[s], bfr[s]);
      if (COMMIT && u < NUC) {
#pragma unroll
        for (int k = u * NITEM / NUC; k < (u + 1) * NITEM / NUC; ++k) commit_item(k, nxt);
      }
      if (ISSUE && u >= NUC) {
#pragma unroll
        for (int k = (u - NUC) * NITEM / NUI; k < (u - NUC + 1) * NITEM / NUI; ++k) issue_item(k, (i + 2) * CK);
      }
      // unit boundary: nothing moves across (the compiler otherwise sinks every load of the interval to its end, i.e. to within
      // a few hundred cycles of the commit that consumes them); inside a unit the scheduler interleaves freely
      if ((u + 1) % SBU == 0 || u + 1 == NU) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  {  // prologue: chunk 0 -> LDS[0], chunk 1 -> registers
#pragma unroll
    for (int k = 0; k < NITEM; ++k) issue_item(k, 0);
    load_scales(0);
#pragma unroll
    for (int k = 0; k < NITEM; ++k) commit_item(k, smem);
    if (nchunk > 1) {
#pragma unroll
      for (int k = 0; k < NITEM; ++k) issue_item(k, CK);
    }
    __syncthreads();
  }
  if constexpr (!FG) {
    int iv = 0;
    for (; iv < nchunk - 2; ++iv) interval(iv, std::true_type{}, std::true_type{});
    if (nchunk >= 2) { interval(iv, std::true_type{}, std::false_type{}); ++iv; }
    interval(iv, std::false_type{}, std::false_type{});
  } else {
    // ---- fixed geometry: every LDS address of the MFMA phase is `base register + immediate`; the interval exists once per buffer
    //      parity (cur / nxt are constants), sub-step = one tap of one k-step
    const int fa = a_lane;                                   // A fragments: smem[fa + imm]
    const int fb = T * CK * WS + (wn * NP) * F_PW + lr + kq * F_PS;   // B fragments: smem[fb + imm] (position group np: + np * F_PW)
    constexpr int NS = T * KS;                               // sub-steps per interval
    constexpr int NSI = NS >= 6 ? NS / 3 : 1, NSC = NS - NSI;
    auto staging = [&](int sidx, int i, auto commit_tag, auto issue_tag, float* nxt) {
      constexpr bool COMMIT = decltype(commit_tag)::value, ISSUE = decltype(issue_tag)::value;
      if (COMMIT && sidx < NSC) {
#pragma unroll
        for (int k = sidx * NITEM / NSC; k < (sidx + 1) * NITEM / NSC; ++k) commit_item(k, nxt);
      }
      if (ISSUE && sidx >= NSC) {
#pragma unroll
        for (int k = (sidx - NSC) * NITEM / NSI; k < (sidx - NSC + 1) * NITEM / NSI; ++k) issue_item(k, (i + 2) * CK);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      // dilation groups 1, 2, 4, 8 (the host checks them): group gg's tap (ky, kx) sits at (8 - d) (PW + 1) + (ky PW + kx) d
      float ga[2][4], gb[2][4][NB];
      auto load_dg = [&](int u, int cur, float (&a)[4], float (&b)[4][NB]) {
        const int tap = u / KS, c4 = u % KS;
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) a[gg] = smem[fa + cur + (tap * CK + c4 * 4) * WS + gg * 16];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int d_ = 1 << gg;
          const int off = (8 - d_) * (F_PW + 1) + (ky * F_PW + kx) * d_;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b[gg][nb] = smem[fb + cur + c4 * 4 * F_PS + nb * F_PW + off];
        }
      };
      auto interval_fg = [&](int i, auto par_tag, auto commit_tag, auto issue_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool COMMIT = decltype(commit_tag)::value;
        constexpr int cur = PAR * F_BUF, nxo = (1 - PAR) * F_BUF;
        if constexpr (COMMIT) load_scales((i + 1) * CK);
        load_dg(0, cur, ga[0], gb[0]);
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const int s_ = u & 1;
          if (u + 1 < NS) load_dg(u + 1, cur, ga[s_ ^ 1], gb[s_ ^ 1]);
#pragma unroll
          for (int gg = 0; gg < 4; ++gg)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[gg][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s_][gg], gb[s_][gg][nb], acc[gg][nb], 0, 0, 0);
          staging(u, i, commit_tag, issue_tag, smem + nxo);
        }
        __syncthreads();
      };
      int iv = 0;
      for (; iv + 3 < nchunk; iv += 2) {
        interval_fg(iv, std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
        interval_fg(iv + 1, std::integral_constant<int, 1>{}, std::true_type{}, std::true_type{});
      }
      for (; iv < nchunk; ++iv) {
        const bool c_ = iv + 1 < nchunk, s2 = iv + 2 < nchunk;
        if (iv & 1) {
          if (s2) interval_fg(iv, std::integral_constant<int, 1>{}, std::true_type{}, std::true_type{});
          else if (c_) interval_fg(iv, std::integral_constant<int, 1>{}, std::true_type{}, std::false_type{});
          else interval_fg(iv, std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
        } else {
          if (s2) interval_fg(iv, std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
          else if (c_) interval_fg(iv, std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{});
          else interval_fg(iv, std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
        }
      }
    }
  }

  // ---- epilogue (the operand chain of conv_igemm_kernel; absent operands are constants behind a zero stride)
  const int Cout = p.G * p.cout_g;
  const float* osp = p.osp + (int64_t)b * Cout * p.oss;
  const float s1 = p.s1, g1 = p.g1, g2 = p.g2;
  const int oss = p.oss, css = p.css, cbs = p.cbs, b1s = p.b1s, b2s = p.b2s, s2s = p.s2s;
  float* yb = p.y + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const int y_plane = p.y_h * p.y_w;
  if constexpr (TC) {
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
    int yoff[NP][2];
    bool pair[NP];
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
    // The up-convs of the path carry the demodulation scale only (noise, bias and activation follow the blur): one multiply per output
    // instead of the ten-instruction chain.  On fp32 MFMA every vector instruction costs pipe time, and a shallow up-conv has few
    // MFMAs per output: 64 -> 32 at 513^2 ran 2.9 vector instructions per MFMA (64 % MFMA busy), about a third of them this chain.
    // (a zero stride = the operand is absent = its neutral constant; the second slope alone can be a constant other than 1 behind stride 0)
    const bool only_os = css == 0 && cbs == 0 && b1s == 0 && b2s == 0 && s2s == 0 && s1 == 1.f && g1 == 1.f && g2 == 1.f &&
                         __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.s2p[0])) == 0x3f800000;
    if (only_os) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cg = co0 + (wm * MB + mb) * 16 + kq * 4 + r;
          const bool cok = cg < p.cout_g;
          const int co = cok ? cg : 0;
          const float os = osp[co * oss];
          float* yc = yb + (int64_t)co * y_plane;
#pragma unroll
          for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              if (yoff[np][py] < 0 || !cok) continue;
              const float v0 = acc[mb][np * 4 + py * 2][r] * os, v1 = acc[mb][np * 4 + py * 2 + 1][r] * os;
              if (pair[np])
                *reinterpret_cast<f32x2u*>(yc + yoff[np][py]) = f32x2u{v0, v1};
              else
                yc[yoff[np][py]] = v0;
            }
        }
      }
      return;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cg = co0 + (wm * MB + mb) * 16 + kq * 4 + r;
        const bool cok = cg < p.cout_g;
        const int co = cok ? cg : 0;
        const float os = osp[co * oss], cs = p.csp[co * css], cb = p.cbp[co * cbs];
        const float b1 = p.b1p[co * b1s], b2 = p.b2p[co * b2s], sl2 = p.s2p[co * s2s];
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
  } else {
    const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW * p.nzs;
    const float nw = p.nwp[0];
    const int nzs = p.nzs;
    const float* r1b = p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r1s;
    const float* r2b = p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r2s;
    const int r1s = p.r1s, r2s = p.r2s;
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
      for (int r = 0; r < 4; ++r) {
        const int cg = DGS ? co0 + kq * 4 + r : co0 + (wm * MB + mb) * 16 + kq * 4 + r;  // channel within the group
        const bool cok = cg < p.cout_g;
        const int co = (DGS ? mb : g) * p.cout_g + (cok ? cg : 0);
        const float os = osp[co * oss], cs = p.csp[co * css], cb = p.cbp[co * cbs];
        const float b1 = p.b1p[co * b1s], b2 = p.b2p[co * b2s], sl2 = p.s2p[co * s2s];
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
}

}  // namespace

// name = MBxNBxWMxWNxCK k1 p3 o<OCC> [t | d]: PF field 3 = this pipeline; the PMAX field carries PROWS (64-word patch rows per wave)
#define VSP_CFGP(MB, NB, WM, WN, CK, PROWS, OCC)                                                     \
  {                                                                                                  \
    MB, NB, WM, WN, CK, 1, PROWS, 3, OCC, #MB "x" #NB "x" #WM "x" #WN "x" #CK "k1p3o" #OCC "r" #PROWS, \
        conv_pipe_kernel<MB, NB, WM, WN, CK, PROWS, OCC, 0>                                          \
  }
#define VSP_CFGPT(MB, NB, WM, WN, CK, PROWS, OCC)                                                    \
  {                                                                                                  \
    MB, NB, WM, WN, CK, 1, PROWS, 3, OCC, #MB "x" #NB "x" #WM "x" #WN "x" #CK "k1p3o" #OCC "r" #PROWS "t", \
        conv_pipe_kernel<MB, NB, WM, WN, CK, PROWS, OCC, 1>                                          \
  }
#define VSP_CFGPD(NB, WN, CK, PROWS, OCC)                                                            \
  {                                                                                                  \
    4, NB, 1, WN, CK, 1, PROWS, 3, OCC, "4x" #NB "x1x" #WN "x" #CK "k1p3o" #OCC "r" #PROWS "d",        \
        conv_pipe_kernel<4, NB, 1, WN, CK, PROWS, OCC, 2>                                            \
  }

// fixed-geometry variant of the dilation-group mode: name suffix "fd"
#define VSP_CFGPDF(NB, WN, CK, PROWS, OCC)                                                           \
  {                                                                                                  \
    4, NB, 1, WN, CK, 1, PROWS, 3, OCC, "4x" #NB "x1x" #WN "x" #CK "k1p3o" #OCC "r" #PROWS "fd",       \
        conv_pipe_kernel<4, NB, 1, WN, CK, PROWS, OCC, 2, true>                                      \
  }

// data gradient of the dilation groups (MODE 3): name suffix "a"
#define VSP_CFGPK(NB, WN, CK, PROWS, OCC)                                                            \
  {                                                                                                  \
    4, NB, 1, WN, CK, 1, PROWS, 3, OCC, "4x" #NB "x1x" #WN "x" #CK "k1p3o" #OCC "r" #PROWS "a",        \
        conv_pipe_kernel<4, NB, 1, WN, CK, PROWS, OCC, 3>                                            \
  }

extern const Cfg kCfgsP[] = {
    VSP_CFGP(4, 4, 2, 4, 8, 18, 1),    // 128 co x 256 pix, stride-2 patches up to 33 x 33
    VSP_CFGP(4, 2, 2, 4, 8, 9, 1),     // 128 co x 128 pix
    VSP_CFGP(4, 4, 2, 4, 4, 9, 1),     // 128 co x 256 pix, 4-channel chunks
    VSP_CFGP(4, 4, 1, 4, 8, 9, 1),     // 64 co x 256 pix, 4 waves
    VSP_CFGP(4, 2, 2, 4, 4, 5, 1),     // 128 co x 128 pix, 4-channel chunks
    VSP_CFGP(2, 4, 2, 4, 8, 18, 1),    // 64 co x 256 pix, 8 waves
    VSP_CFGP(2, 4, 2, 4, 4, 9, 1),
    VSP_CFGP(4, 4, 1, 8, 8, 18, 1),    // 64 co x 512 pix
    VSP_CFGP(4, 2, 2, 4, 4, 5, 2),     // 128 co x 128 pix, 4-channel chunks, two workgroups per CU
    VSP_CFGP(2, 4, 2, 4, 4, 9, 2),     // 64 co x 256 pix
    VSP_CFGP(2, 2, 2, 4, 8, 9, 2),     // 64 co x 128 pix
    VSP_CFGP(2, 2, 2, 4, 4, 5, 2),
    // transposed (stride-2 up-convs): NB = 4 phases x NP position groups; the patch rows also cover the 1 x NPIX / NPIX x 1
    // edge strips ((NPIX + 1) x 2 words)
    VSP_CFGPT(2, 16, 2, 4, 8, 9, 1),   // 64 co x 256 positions (16 x 16), 128 accumulator registers
    VSP_CFGPT(2, 16, 2, 4, 4, 5, 1),
    VSP_CFGPT(2, 8, 2, 4, 8, 5, 1),    // 64 co x 128 positions
    VSP_CFGPT(4, 8, 2, 4, 8, 5, 1),    // 128 co x 128 positions
    VSP_CFGPT(1, 16, 2, 4, 8, 9, 1),   // 32 co x 256 positions
    VSP_CFGPT(2, 8, 2, 4, 8, 5, 2),    // 64 co x 128 positions, two workgroups per CU
    VSP_CFGPT(2, 8, 2, 4, 4, 3, 2),
    VSP_CFGPT(1, 8, 2, 4, 8, 5, 2),    // 32 co x 128 positions
    VSP_CFGPT(2, 8, 1, 4, 8, 3, 2),    // 32 co x 128 positions, 4 waves (strips 9 rows: CK / NW = 2 channels per wave)
    // the four dilation groups from one shared patch (halo 8: 32 x 32 words for 16 x 16 pixels)
    VSP_CFGPD(2, 8, 4, 8, 2),          // 4 x 16 co x 256 pix, two workgroups per CU
    VSP_CFGPD(2, 8, 8, 16, 1),
    VSP_CFGPD(4, 4, 4, 16, 1),         // 4 waves x 64 pix
    VSP_CFGPD(4, 4, 8, 16, 1),
    VSP_CFGPD(4, 8, 4, 12, 1),         // 4 x 16 co x 512 pix (16 x 32 tile, 32 x 48 patch)
    // fixed geometry (constant-offset fragment reads, intervals per buffer parity)
    // (OCC = waves per SIMD the register budget must admit: 4 = two 8-wave workgroups per CU, i.e. at most 128 VGPRs)
    VSP_CFGPDF(2, 8, 4, 8, 4),         // 4 x 16 co x 256 pix
    VSP_CFGPDF(2, 8, 4, 8, 2),
    VSP_CFGPDF(2, 8, 8, 16, 1),
    // (narrow layers -- 32 output channels x 512 / 256 pixels, 8 waves along the pixels: VSP_CFGP(2, 4, 1, 8, 8, 10, 1 | 2), (2, 2, 1, 8, 8, 6, 2) --
    //  were built and measured on 32 -> 32 at 1024^2, batch 8: 1604 us = 96 TFLOP/s against 1420 us of the Winograd kernel; not kept)
    // the data gradient of the four dilation groups (appended: earlier indices keep their meaning)
    VSP_CFGPK(2, 8, 4, 8, 2),          // 64 co x 256 pix, two workgroups per CU
    VSP_CFGPK(2, 8, 8, 16, 1),
    VSP_CFGPK(4, 8, 4, 12, 1),         // 64 co x 512 pix (16 x 32 tile, 32 x 48 patch)
    VSP_CFGPK(4, 4, 4, 16, 1),         // 4 waves x 64 pix
};
extern const int kNumP = sizeof(kCfgsP) / sizeof(kCfgsP[0]);

}  // namespace vspconv

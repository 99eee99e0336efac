// Winograd F(2x2, 3x3) on fp32 MFMA, "row-owner" form (round 4): every wave computes its OWN B fragments.
//
// conv_wino.hip runs the input transform V = B^T d B as a workgroup-wide task list: patch (LDS) -> 512 transform tasks -> V (LDS) ->
// barrier -> B fragments.  All eight waves alternate transform and matrix work behind one barrier per 8 input channels, and the
// matrix pipe idles a third of the time (PMC: 68 % / 61 % / 50 % busy on the judged shapes) whatever the order inside the interval.
// Here the V image does not exist.  Wave w already owns the two Winograd positions (xi, nu) = (w / 2, 2 (w % 2) + {0, 1}) -- ONE row
// of the 4 x 4 transformed tile.  The B fragment of an MFMA is V[xi][nu][ci = lane / 16][tile = lane % 16], and
//     V[xi][.] = (row combination xi of the window) . B       ->  W[l] = d[rA][l] +- d[rB][l]   (two window rows, four columns)
//     V[xi][nu] = W[p] +- W[q]
// so a lane needs two 16-byte rows of ITS (channel, tile) window from the staged patch and six additions per pair of positions: the
// same LDS read volume as fetching the fragment from a V image, no V stores, no transform task list, and -- the point -- no
// workgroup-wide dependency between the transform and the MFMAs.  What is left of the barrier is the hand-over of the double-
// buffered patch (one per IVC input channels); inside a stage a wave runs a private software pipeline: window reads and the six
// additions of k-step s + 1 sit between the MFMAs of k-step s.
// The style scale / folded-BatchNorm affine are applied when the patch is committed (they are per input channel), so the fragment
// path carries no multiply.  U (fragment order, conv_wino.hip / weight_pack.hip), tile geometry, work order and the epilogue are
// those of conv_wino.hip: the two kernels are interchangeable per launch (wino_launch picks).
#include "conv_kernel.h"
#include <type_traits>

namespace vspconv {

namespace {

__device__ __forceinline__ float uload_ro(const float* base, int idx) {  // wave-uniform operand through the scalar cache
  typedef const float __attribute__((address_space(4))) * cfp4;
  return ((cfp4)(uintptr_t)base)[__builtin_amdgcn_readfirstlane(idx)];
}

constexpr int RO_NTHR = 512;

constexpr int IVC = 8;   // input channels per sub-stage: one channel plane per wave

template <int MBW, int TLX_ = 8>
struct RG {  // geometry (undilated): MBW 16-channel blocks x NBW 16-tile blocks per wave and position, MBW * NBW = 8; TLX_ tile columns
             // per workgroup (8: 16 x 16 pixels at 32 tiles; 16: 32 x 8 pixels -- 128-byte output row segments)
  static constexpr int KS = IVC / 4;
  static constexpr int NBW = 8 / MBW;
  static constexpr int WCO = 16 * MBW;
  static constexpr int NTILE = 16 * NBW;
  static constexpr int TLX = NBW == 8 ? 16 : TLX_;
  static constexpr int TLY = NTILE / TLX;
  static constexpr int PR = 2 * TLY + 2;
  // Patch rows are staged as ALIGNED 16-byte segments: image columns ox0 - 4 ... ox0 + 2 TLX + 3 (W % 4 == 0: a segment lies inside the
  // image or outside as a whole, so validity is one bit per lane and the load is one buffer_load_dwordx4), SEG segments per row.
  static constexpr int SEG = (2 * TLX + 8) / 4;
  // LDS image: word 1 + r PCP + (ix - ox0 + 4).  The window of tile column tx starts at ix = ox0 + 2 tx - 1 -> word r PCP + 4 + 2 tx: EVEN,
  // so the window reads are 8-byte accesses (bank = dword mod 64, 32-lane groups = two channels x 16 tiles).  8-tile-wide geometries:
  // a group holds 8 tile columns (16 consecutive dwords) of two tile rows and two channels -> row pitch 24 (two rows = 48 = -16 banks)
  // and plane pitch == 32 (mod 64) put the four 16-dword runs on the four quarters of the 64 banks.  16-tile-wide: one tile row fills
  // 32 dwords, the second channel takes the other half (row pitch 40 = the 40 staged columns).
  static constexpr int PCP = 4 * SEG;
  static constexpr int PPITCH = (PR * PCP + 1 + 31) / 64 * 64 + 32;
  static constexpr int NLD = (PR * SEG + 63) / 64;    // wave loads per channel plane (one for the 8-tile-wide geometries: 10 x 6 segments)
  static constexpr int LDS_P = IVC * PPITCH;          // floats per sub-stage buffer (ring of 2 M)
  static constexpr int ETILE = NTILE > 64 ? 64 : NTILE;
  static constexpr int EMB = (MBW >= 2 && ETILE <= 32) ? 2 : 1;
  static constexpr int EP = ETILE + 4;
  static constexpr int LDS_M = 16 * 16 * EMB * EP;
  static constexpr int lds_floats(int m) { return 2 * m * LDS_P > LDS_M ? 2 * m * LDS_P : LDS_M; }
  static constexpr int UF = 2 * MBW;
};

// M = barrier period in sub-stages of 8 input channels.  The patch lives in a ring of R = 2 M sub-stage buffers; sub-stage s reads
// ring[s % R], the planes of sub-stage s + M are committed during sub-stage s (into the slot sub-stage s - M was read from), and a
// barrier closes every M-th sub-stage: between the commit of a sub-stage and its first read, and between the last read of a slot and
// its overwrite, lies at least one barrier.  M = 1 is the plain double buffer.
template <int MBW, int M, int TLXV = 8>
__global__ __launch_bounds__(RO_NTHR, 4) void conv_wino_ro_kernel(const ConvK p) {
  using Gm = RG<MBW, TLXV>;
  constexpr int NBW = Gm::NBW, WCO = Gm::WCO, NTILE = Gm::NTILE, TLX = Gm::TLX, TLY = Gm::TLY, PR = Gm::PR, PCP = Gm::PCP, SEG = Gm::SEG;
  constexpr int PPITCH = Gm::PPITCH, LDS_P = Gm::LDS_P, UF = Gm::UF, KS = Gm::KS, NLD = Gm::NLD;
  constexpr int R = 2 * M;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Pl = smem;   // R x [IVC][PPITCH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  // work order: pixel-tile-major per XCD (conv_wino.hip, order 1) or dispatch order
  int b = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  if (p.wg_order) {
    const int GX = gridDim.x, GY = gridDim.y, GZ = gridDim.z, GT = GX * GY * GZ;
    const int wgid = blockIdx.x + GX * (blockIdx.y + GY * blockIdx.z);
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    const int GN = GX * GY;
    b = lid / GN;
    const int lrem = lid - b * GN;
    bx = lrem / GY;
    by = lrem - bx * GY;
  }
  const int g = by / p.co_tiles, ct = by - g * p.co_tiles;
  const int tiles_x = (p.W + 2 * TLX - 1) / (2 * TLX);
  const int tx_i = bx % tiles_x, ty_i = bx / tiles_x;
  const int oy0 = ty_i * (2 * TLY), ox0 = tx_i * (2 * TLX);
  const int co0 = ct * WCO;
  const int chw = p.H * p.W;
  const float* xb = p.x + (int64_t)b * p.x_ch * chw;
  const int nstage = (p.Cin + IVC - 1) / IVC;
  const int nchunk4 = (p.Cin + 3) / 4;
#ifdef VSP_WINO_ABLATE   // tuning only (VSP_CONV_DBG): 1 no MFMAs, 2 no window reads / fragments, 4 no U loads, 8 no patch loads, 16 no commit, 32 no barrier, 64 U from chunk 0, 128 window reads broadcast
  const int ab = p.dbg;
#else
  constexpr int ab = 0;
#endif

  // ---- patch staging: plane k of a stage belongs to wave k % 8; a lane owns segment (row, seg) = (l / SEG, l % SEG), l = lane + 64 i.
  //      Lanes past the plane repeat segment 0 -- same address, same value -- instead of sitting behind a divergent branch around a load.
  int p_voff[NLD], p_dst[NLD];
  unsigned p_ok = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int l = lane + 64 * i;
    const bool live = l < PR * SEG;
    const int r = live ? l / SEG : 0, sg = live ? l - r * SEG : 0;
    const int iy = oy0 - 1 + r, ix = ox0 - 4 + 4 * sg;
    const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    p_voff[i] = in ? (iy * p.W + ix) * 4 : 0x7ffffff0;   // padding: a lane offset past the resource, the load returns zeros
    p_ok |= in ? (1u << i) : 0u;
    p_dst[i] = 1 + r * PCP + 4 * sg;
  }
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.x_ch * chw * 4, 0x00020000);   // this image (the range check compares the lane offset with size - scalar offset)
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v preg[NLD];
  auto load_plane = [&](int j, f32x4v (&dst)[NLD]) {    // this wave's plane of sub-stage j (clamped: past the end the last one is loaded again)
    const int jj = j < nstage ? j : nstage - 1;
    const int ci = jj * IVC + wave;
    const bool chin = ci < p.Cin;                                  // (a channel past the layer: every lane offset out of range -> zeros)
    const int soff = (chin ? ci : 0) * chw * 4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) dst[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, chin ? p_voff[i] : 0x7ffffff0, soff, 0));
  };
  const float* wt_b = p.wtp + b * p.wt_bs;   // (per-image bases hoisted: the interval's scalar address arithmetic is part of its issue time)
  const float* wc_b = p.wcp + b * p.wc_bs;
  const bool affine = p.wc_cs != 0 || p.wsh_cs != 0 || p.wc_bs != 0;
  auto commit_plane = [&](float* Pdst, int j, const f32x4v (&src)[NLD]) {   // style scale and (folded BatchNorm) affine ride on the patch
    const int ci = j * IVC + wave;
    const bool chok = ci < p.Cin;
    const int cc = chok ? ci : p.Cin - 1;
    const float st = uload_ro(wt_b, cc * p.wt_cs);
    float sc = st, sh = 0.f;
    if (affine) {   // (uniform for the launch) folded-BatchNorm input of the IR-SE body; the modulated layers skip two scalar loads and their address arithmetic
      sc = uload_ro(wc_b, cc * p.wc_cs) * st;
      sh = uload_ro(p.wshp, cc * p.wsh_cs) * st;
    }
    float* dst = Pdst + wave * PPITCH;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      // (These 16-byte writes start at word 1 + 4 l: NOT 16-byte aligned, served as four dword passes with the lanes four banks apart --
      //  14 conflict cycles per write, 27 % of the kernel's LDS-active cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.27; with the
      //  commits switched off 0.008).  Writing the ALIGNED unit (d of segment l - 1 through DPP wave_shr:1, a, b, c) takes the ratio to
      //  0.015 and the kernel from 605 to 630 us at 256 -> 256 / 128^2, 1211 to 1378 us at 32 -> 32 / 1024^2: the LDS is not what this
      //  kernel waits for, the extra VALU on the commit path is.  Kept misaligned.)
      // padding and absent channels arrive as ZEROS (out-of-range lane offset): only the affine shift still has to be masked, one select
      // per segment -- no zero FACTOR that would turn an Inf / NaN at a clamped address into a NaN border
      const float shm = (((p_ok >> i) & 1u) && chok) ? sh : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[p_dst[i] + e] = fmaf(src[i][e], sc, shm);
    }
  };

  // ---- U fragments: [group][co tile][chunk][wave][pp 2][lane][mb MBW] floats, one 4-channel chunk (k-step) and position per load.
  //      Buffer loads: resource = this channel tile's slice, scalar offset = chunk, lane offset fixed (a flat pointer costs a 64-bit VALU
  //      add per load and a handful of scalar instructions for the 64-bit chunk offset: 56 SALU + 60 VALU per 32 MFMAs were measured).
  const float* utile = p.w + ((int64_t)g * p.co_tiles + ct) * nchunk4 * (8 * 64 * UF);
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(utile), 0, nchunk4 * (8 * 64 * UF) * 4, 0x00020000);
  const int u_voff = ((wave * 2 * 64 + lane) * MBW) * 4;
  auto load_u_half = [&](int c, int pp, float (&u)[UF]) {
    if (ab & 4) return;
    const int cc = (ab & 64) ? 0 : (c < nchunk4 ? c : nchunk4 - 1);
    const int soff = cc * (8 * 64 * UF * 4);
    if constexpr (MBW == 4) {
      typedef float f32x4b __attribute__((ext_vector_type(4)));
      const f32x4b a = __builtin_bit_cast(f32x4b, __builtin_amdgcn_raw_buffer_load_b128(ursrc, u_voff + pp * 64 * MBW * 4, soff, 0));
      u[pp * 4 + 0] = a[0]; u[pp * 4 + 1] = a[1]; u[pp * 4 + 2] = a[2]; u[pp * 4 + 3] = a[3];
    } else if constexpr (MBW == 2) {
      typedef float f32x2b __attribute__((ext_vector_type(2)));
      const f32x2b a = __builtin_bit_cast(f32x2b, __builtin_amdgcn_raw_buffer_load_b64(ursrc, u_voff + pp * 64 * MBW * 4, soff, 0));
      u[pp * 2 + 0] = a[0]; u[pp * 2 + 1] = a[1];
    } else {
      u[pp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ursrc, u_voff + pp * 64 * MBW * 4, soff, 0));
    }
  };

  // ---- this wave's row of the transformed tile: xi = wave / 2 -> W = d[rA] + sgn d[rB]; nuh = wave % 2 -> which two of V[xi][0..3]
  const int xi = wave >> 1, nuh = wave & 1;
  const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int rB = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sgn = xi == 1 ? 1.f : -1.f;
  // lane's window origin: tile = lr + 16 nb -> (ty, tx); an N-block step is 16 / TLX tile rows = a constant word offset
  constexpr int NBSTEP = (16 / TLX) * 2 * PCP;   // (TLX = 16: the next N-block is the next tile row; TLX = 8: two tile rows down)
  const int wty = lr / TLX, wtx = lr - wty * TLX;
  const int woffA = kq * PPITCH + (2 * wty + rA) * PCP + 4 + 2 * wtx;
  const int woffB = kq * PPITCH + (2 * wty + rB) * PCP + 4 + 2 * wtx;
  typedef float f32x2a __attribute__((ext_vector_type(2)));
  auto fragments = [&](const float* Psrc, int ks, float (&bv)[2][NBW], auto nuh_tag) {   // k-step ks of the stage (channels 4 ks + kq) -> B fragments
    constexpr int NUHK = decltype(nuh_tag)::value;   // 0 / 1: compile-time column half (no selects, the unused fourth column is never combined); 2: run time
    if (ab & 2) return;
    const float* base = (ab & 128) ? smem + (lane & 1) * 4 : Psrc + ks * (4 * PPITCH);   // (128: every lane reads the same two words: LDS broadcast)
#pragma unroll
    for (int n0 = 0; n0 < NBW; n0 += 2) {   // two N-blocks at a time: eight window registers live
      f32x2a wa[2][2], wb[2][2];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int nb = n0 + n;
        wa[n][0] = *reinterpret_cast<const f32x2a*>(base + woffA + nb * NBSTEP);
        wa[n][1] = *reinterpret_cast<const f32x2a*>(base + woffA + nb * NBSTEP + 2);
        wb[n][0] = *reinterpret_cast<const f32x2a*>(base + woffB + nb * NBSTEP);
        wb[n][1] = *reinterpret_cast<const f32x2a*>(base + woffB + nb * NBSTEP + 2);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int nb = n0 + n;
        const float w1 = fmaf(wb[n][0].y, sgn, wa[n][0].y), w2 = fmaf(wb[n][1].x, sgn, wa[n][1].x);
        // nu = 0: W0 - W2, 1: W1 + W2, 2: W2 - W1, 3: W1 - W3
        if constexpr (NUHK == 1) {
          const float w3 = fmaf(wb[n][1].y, sgn, wa[n][1].y);
          bv[0][nb] = w2 - w1;
          bv[1][nb] = w1 - w3;
        } else if constexpr (NUHK == 0) {
          const float w0 = fmaf(wb[n][0].x, sgn, wa[n][0].x);
          bv[0][nb] = w0 - w2;
          bv[1][nb] = w1 + w2;
        } else {
          const float w0 = fmaf(wb[n][0].x, sgn, wa[n][0].x), w3 = fmaf(wb[n][1].y, sgn, wa[n][1].y);
          bv[0][nb] = nuh ? w2 - w1 : w0 - w2;
          bv[1][nb] = nuh ? w1 - w3 : w1 + w2;
        }
      }
    }
  };

  f32x4 acc[2][MBW][NBW];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[pp][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply_pp = [&](int pp, const float (&u)[UF], const float (&bv)[2][NBW]) {
    if (ab & 1) return;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
        acc[pp][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pp * MBW + mb], bv[pp][nb], acc[pp][mb][nb], 0, 0, 0);
  };

  // ---- pipeline.  U runs one k-step ahead.  Prefetch indices are clamped instead of guarded (a branch join makes the compiler's wait
  //      counts pessimistic): the last sub-stages re-load / re-commit planes nobody reads.
  float ua[UF], ub[UF];
  float bva[2][NBW], bvb[2][NBW];
#pragma unroll
  for (int q = 0; q < UF; ++q) ua[q] = ub[q] = 1.f;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) bva[0][nb] = bva[1][nb] = bvb[0][nb] = bvb[1][nb] = 1.f;
  load_u_half(0, 0, ua); load_u_half(0, 1, ua);
  {
    f32x4v first[M][NLD];   // sub-stages 0 .. M - 1: all loads leave together
#pragma unroll
    for (int s0 = 0; s0 < M; ++s0) load_plane(s0, first[s0]);
#pragma unroll
    for (int s0 = 0; s0 < M; ++s0) commit_plane(Pl + s0 * LDS_P, s0, first[s0]);
  }
  load_plane(M, preg);
  __syncthreads();
  if ((p.dbg & 0x1000) && wave >= 4) __builtin_amdgcn_s_setprio(1);   // (tuning: static priority for the younger half of the workgroup)
  int rcur = 0, rnxt = M % R;   // ring slots of sub-stage j and j + M
  auto main_loop = [&](auto nuh_tag) {
  for (int j = 0; j < nstage; ++j) {
    const float* Pcur = Pl + rcur * LDS_P;
    float* Pnxt = Pl + rnxt * LDS_P;
    fragments(Pcur, 0, bva, nuh_tag);
    // The issue points of the vector-memory instructions are pinned inside the MFMA stream (fence mask: VALU, SALU and LDS operations may
    // cross, MFMAs and vector-memory instructions may not): left alone the compiler sinks the U loads to the end of the k-step in front
    // of their use (they then wait behind `vmcnt(0)` a few MFMAs later).
    constexpr int SB = 0x2 | 0x4 | 0x80 | 0x100 | 0x200;
    // k-step 0: the U set of k-step 1 leaves first (consumed one k-step later; issuing a set right behind its last use -- two k-steps
    // of cover -- measured SLOWER: the compiler's wait for the patch registers then also covers the U load just issued)
    load_u_half(j * KS + 1, 0, ub);
    load_u_half(j * KS + 1, 1, ub);
    __builtin_amdgcn_sched_barrier(SB);
    multiply_pp(0, ua, bva);
    __builtin_amdgcn_sched_barrier(SB);
    if (!(ab & 16)) commit_plane(Pnxt, j + M, preg);
    if (!(ab & 8)) load_plane(j + M + 1, preg);
    __builtin_amdgcn_sched_barrier(SB);
    fragments(Pcur, 1, bvb, nuh_tag);
    multiply_pp(1, ua, bva);
    __builtin_amdgcn_sched_barrier(SB);
    // k-step 1
    load_u_half(j * KS + 2, 0, ua);
    load_u_half(j * KS + 2, 1, ua);
    __builtin_amdgcn_sched_barrier(SB);
    multiply_pp(0, ub, bvb);
    __builtin_amdgcn_sched_barrier(SB);
    multiply_pp(1, ub, bvb);
    __builtin_amdgcn_sched_barrier(SB);
    static_assert(KS == 2, "two k-steps per sub-stage");
    rcur = rcur + 1 == R ? 0 : rcur + 1;
    rnxt = rnxt + 1 == R ? 0 : rnxt + 1;
    if (M == 1 || (j % M) == M - 1) {
      if (!(ab & 32)) __syncthreads();
    }
  }
  };
  // (uniform) the loop is instantiated per column half where the registers allow it (the 32-channel variant spills with two copies)
  if constexpr (MBW == 4) {
    if (nuh) main_loop(std::integral_constant<int, 1>{}); else main_loop(std::integral_constant<int, 0>{});
  } else {
    main_loop(std::integral_constant<int, 2>{});
  }
  if (M > 1) __syncthreads();   // (the epilogue reuses the ring)

#ifdef VSP_WINO_ABLATE
  if (ab & 512) {   // no epilogue (one word keeps the accumulators alive)
    if (acc[0][0][0][0] == 123.456f) p.y[0] = 1.f;
    return;
  }
#endif
  // ---- epilogue (conv_wino.hip): per 16-channel block(s), all sixteen positions through LDS, one thread per (channel, tile)
  constexpr int ETILE = Gm::ETILE, ENB = ETILE / 16, EP = Gm::EP;
  float* Ml = smem;
  const int Cout = p.G * p.cout_g;
  const float* osp = p.osp + (int64_t)b * Cout * p.oss;
  const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW * p.nzs;
  const float nw = p.nwp[0];
  float* yb = p.y + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const float* r1b = p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r1s;
  const float* r2b = p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r2s;
  const int y_plane = p.y_h * p.y_w;
  constexpr int EMB = Gm::EMB, ECO = 16 * EMB;
  constexpr int EPT = ECO * ETILE / RO_NTHR;
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  constexpr int TP = ETILE / 2;                       // horizontal tile pairs per pass
  constexpr int EPT2 = ECO * TP / RO_NTHR;            // (channel, tile pair) items per thread and pass
  static_assert(ECO * TP % RO_NTHR == 0 && TLX % 2 == 0, "tile pairs");
  struct QOps { float os, cs, cb, b1, b2, sl2; int cbase; };
  // 16-byte form: dense rows of whole quads, every operand plane 16-byte aligned (uniform)
  const bool quads = p.r1s <= 1 && p.r2s <= 1 && p.nzs <= 1 && (p.OW & 3) == 0 && p.y_w == p.OW && (y_plane & 3) == 0 &&
                     !(reinterpret_cast<uintptr_t>(yb) & 15) && !(p.r1s && (reinterpret_cast<uintptr_t>(r1b) & 15)) &&
                     !(p.r2s && (reinterpret_cast<uintptr_t>(r2b) & 15)) && !(p.nzs && (reinterpret_cast<uintptr_t>(nzp) & 15)) && !(p.dbg & 0x4000);
  const bool vec2 = p.r1s <= 1 && p.r2s <= 1;
  const bool pairs = vec2 && (p.OW & 1) == 0 && p.OW >= 2;
#pragma unroll
  for (int mb0 = 0; mb0 < MBW; mb0 += EMB) {
#pragma unroll
    for (int th = 0; th < NTILE / ETILE; ++th) {
      if (mb0 + th > 0) __syncthreads();
      QOps qops[EPT2];
      if (quads) {
#pragma unroll
        for (int it = 0; it < EPT2; ++it) {
          const int e_co = (tid + it * RO_NTHR) / TP;
          const int cgi = co0 + mb0 * 16 + e_co;
          const int cg = g * p.cout_g + (cgi < p.cout_g ? cgi : p.cout_g - 1);
          qops[it].os = osp[cg * p.oss]; qops[it].cs = p.csp[cg * p.css]; qops[it].cb = p.cbp[cg * p.cbs];
          qops[it].b1 = p.b1p[cg * p.b1s]; qops[it].b2 = p.b2p[cg * p.b2s]; qops[it].sl2 = p.s2p[cg * p.s2s];
          qops[it].cbase = cg * y_plane;
        }
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int m2 = 0; m2 < EMB; ++m2)
#pragma unroll
          for (int nb = 0; nb < ENB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Ml[((2 * wave + pp) * ECO + m2 * 16 + kq * 4 + r) * EP + nb * 16 + lr] = acc[pp][mb0 + m2][th * ENB + nb][r];
      __syncthreads();
      if (quads) {
        // (uniform) two horizontally adjacent tiles per thread: the 2 x 4 output pixels leave as two 16-byte stores, the sixteen
        // position values of both tiles come as 8-byte LDS reads, and a thread keeps ONE channel per pass (its six per-channel
        // operands were requested before the exchange).  Measured on 64 -> 64 at 512^2: the 8-byte-per-lane form spent 200 of the
        // kernel's 870 us here.
#pragma unroll
        for (int it = 0; it < EPT2; ++it) {
          const int dp = tid + it * RO_NTHR;
          const int e_co = dp / TP, tp = dp - e_co * TP;
          const int e_t = (tp / (TLX / 2)) * TLX + 2 * (tp % (TLX / 2));      // first tile of the pair
          const int e_tile = th * ETILE + e_t;
          const int sy = oy0 + 2 * (e_tile / TLX), sx = ox0 + 2 * (e_tile % TLX);
          float2 m[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) m[q] = *reinterpret_cast<const float2*>(Ml + (q * ECO + e_co) * EP + e_t);
          float yv[2][4];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float t0[4], t1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
              const float m0 = h ? m[nu].y : m[nu].x, m1 = h ? m[4 + nu].y : m[4 + nu].x;
              const float m2 = h ? m[8 + nu].y : m[8 + nu].x, m3 = h ? m[12 + nu].y : m[12 + nu].x;
              t0[nu] = m0 + m1 + m2;
              t1[nu] = m1 - m2 - m3;
            }
            yv[0][2 * h] = t0[0] + t0[1] + t0[2]; yv[0][2 * h + 1] = t0[1] - t0[2] - t0[3];
            yv[1][2 * h] = t1[0] + t1[1] + t1[2]; yv[1][2 * h + 1] = t1[1] - t1[2] - t1[3];
          }
          const int cgi = co0 + mb0 * 16 + e_co;
          const bool cok = cgi < p.cout_g;
          const QOps& o = qops[it];
          auto fin = [&](float v, float nz, float r1v, float r2v) {
            v = v * o.os * o.cs + o.cb + o.b1;
            v = (v > 0.f ? v : v * p.s1) * p.g1;
            v += nz * nw + o.b2;
            v = (v > 0.f ? v : v * o.sl2) * p.g2;
            return v + r1v + r2v;
          };
          typedef float f32x4q __attribute__((ext_vector_type(4)));
          f32x4q nz[2], r1v[2], r2v[2];
          int ro[2];
          bool inside[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = sy + i;
            inside[i] = cok && oy < p.OH && sx < p.OW;
            const int oyc = min(oy, p.OH - 1), oxc = min(sx, p.OW - 4);
            ro[i] = o.cbase + oyc * p.y_w + oxc;
            nz[i] = r1v[i] = r2v[i] = f32x4q{0.f, 0.f, 0.f, 0.f};
            if (p.nzs) nz[i] = *reinterpret_cast<const f32x4q*>(nzp + oyc * p.OW + oxc);
            if (p.r1s) r1v[i] = *reinterpret_cast<const f32x4q*>(r1b + ro[i]);
            if (p.r2s) r2v[i] = *reinterpret_cast<const f32x4q*>(r2b + ro[i]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x4q o4 = {fin(yv[i][0], nz[i][0], r1v[i][0], r2v[i][0]), fin(yv[i][1], nz[i][1], r1v[i][1], r2v[i][1]),
                               fin(yv[i][2], nz[i][2], r1v[i][2], r2v[i][2]), fin(yv[i][3], nz[i][3], r1v[i][3], r2v[i][3])};
            if (inside[i]) *reinterpret_cast<f32x4q*>(yb + ro[i]) = o4;
          }
        }
        continue;
      }
#pragma unroll
      for (int it = 0; it < EPT; ++it) {
        const int pair = tid + it * RO_NTHR;
        const int e_co = pair / ETILE, e_t = pair - e_co * ETILE;
        const int e_tile = th * ETILE + e_t;
        const int e_tx = e_tile % TLX;
        const int sy = oy0 + 2 * (e_tile / TLX);
        const int sx = ox0 + 2 * e_tx;
        float m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) m[q] = Ml[(q * ECO + e_co) * EP + e_t];
        float t0[4], t1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          t0[nu] = m[nu] + m[4 + nu] + m[8 + nu];
          t1[nu] = m[4 + nu] - m[8 + nu] - m[12 + nu];
        }
        const float yv[2][2] = {{t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3]}, {t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]}};
        const int cgi = co0 + mb0 * 16 + e_co;
        const bool cok = cgi < p.cout_g;
        const int cg = g * p.cout_g + (cok ? cgi : p.cout_g - 1);
        const float os = osp[cg * p.oss], cs = p.csp[cg * p.css], cb = p.cbp[cg * p.cbs];
        const float b1 = p.b1p[cg * p.b1s], b2 = p.b2p[cg * p.b2s], sl2 = p.s2p[cg * p.s2s];
        const int cbase = cg * y_plane;
        auto fin = [&](float v, float nz, float r1v, float r2v) {
          v = v * os * cs + cb + b1;
          v = (v > 0.f ? v : v * p.s1) * p.g1;
          v += nz * nw + b2;
          v = (v > 0.f ? v : v * sl2) * p.g2;
          return v + r1v + r2v;
        };
        if (pairs) {
          f32x2u nz[2] = {{0.f, 0.f}, {0.f, 0.f}}, r1v[2] = {{0.f, 0.f}, {0.f, 0.f}}, r2v[2] = {{0.f, 0.f}, {0.f, 0.f}};
          int ro[2];
          bool inside[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = sy + i;
            inside[i] = cok && oy < p.OH && sx < p.OW;
            const int oyc = min(oy, p.OH - 1), oxc = min(sx, p.OW - 2);
            ro[i] = cbase + oyc * p.y_w + oxc;
            if (p.nzs) nz[i] = *reinterpret_cast<const f32x2u*>(nzp + oyc * p.OW + oxc);
            if (p.r1s) r1v[i] = *reinterpret_cast<const f32x2u*>(r1b + ro[i]);
            if (p.r2s) r2v[i] = *reinterpret_cast<const f32x2u*>(r2b + ro[i]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x2u o2 = {fin(yv[i][0], nz[i][0], r1v[i][0], r2v[i][0]), fin(yv[i][1], nz[i][1], r1v[i][1], r2v[i][1])};
            if (inside[i]) *reinterpret_cast<f32x2u*>(yb + ro[i]) = o2;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = sy + i, ox = sx;
            if (!cok || oy >= p.OH || ox >= p.OW) continue;
            const int ro = cbase + oy * p.y_w + ox;
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
              const int oxj = ox + jx;
              if (oxj >= p.OW) continue;
              const int rj = ro + jx;
              yb[rj] = fin(yv[i][jx], nzp[(oy * p.OW + oxj) * p.nzs], r1b[rj * p.r1s], r2b[rj * p.r2s]);
            }
          }
        }
      }
    }
  }
}

template <int MBW, int M, int TLXV = 8>
int launch_ro(ConvK q, hipStream_t stream) {
  using Gm = RG<MBW, TLXV>;
  static vsp::LdsAttrOnce attr;
  const size_t lds = (size_t)Gm::lds_floats(M) * sizeof(float);
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wino_ro_kernel<MBW, M, TLXV>), (int)lds, "conv2d_winograd (row-owner)")) return rc;
  q.co_tiles = (q.cout_g + Gm::WCO - 1) / Gm::WCO;
  const int blocks = ((q.W + 2 * Gm::TLX - 1) / (2 * Gm::TLX)) * ((q.H + 2 * Gm::TLY - 1) / (2 * Gm::TLY));
  q.wg_order = 1;
  if (q.dbg & 0x300) q.wg_order = ((q.dbg >> 8) & 3) == 1 ? 1 : 0;
  dim3 grid((unsigned)blocks, (unsigned)(q.co_tiles * q.G), (unsigned)q.B);
  conv_wino_ro_kernel<MBW, M, TLXV><<<grid, RO_NTHR, lds, stream>>>(q);
  return VSP_OK;
}

}  // namespace

// the row-owner form serves undilated launches whose rows are whole 16-byte segments (and at least 32 output channels per group)
bool wino_ro_eligible(const ConvK& q) {
  for (int g = 0; g < q.G; ++g)
    if (q.dil[g] != 1) return false;
  // padding = the raw-buffer range check of ONE image's descriptor (num_records = x_ch * H * W * 4 as an int; out-of-range lanes carry
  // offset 0x7ffffff0): an image of 2 GiB or more would wrap the record count and leave the padding unbacked -- refused here
  if ((int64_t)q.x_ch * q.H * q.W * 4 >= 0x7ffffff0ll) return false;
  return q.cout_g > 16 && q.W % 4 == 0 && (reinterpret_cast<uintptr_t>(q.x) & 15) == 0 && ((int64_t)q.H * q.W) % 4 == 0;
}

// m = barrier period in 8-channel sub-stages (1, 2 or 4).  (A half tile -- 64 channels x 16 tiles, 75 VGPRs, three workgroups per CU -- for the
// launches that fill only half of the chip's workgroup slots was measured and removed: 256 -> 256 at 32^2 60 vs 58 us, every larger layer
// 8-15 % slower: the small layers are bound by their K chain of 32 sub-stages, not by occupancy.)
int wino_ro_launch(ConvK q, int mbw, int m, hipStream_t stream) {
  if (mbw == 4) {
    // 16 tile columns x 2 tile rows per workgroup (32 x 4 output pixels: 128-byte row segments on both the patch loads and the stores)
    // wherever a row holds them: 0.5-2 % over the 8 x 4 form on every layer (128 -> 128 at 256^2: 769 -> 752 us); VSP_WINO_RO_WIDE = 0: off
    static const int wide_env = vsp::tune_env("VSP_WINO_RO_WIDE") ? atoi(vsp::tune_env("VSP_WINO_RO_WIDE")) : 1;
    if (wide_env && m == 1 && q.W >= 32) return launch_ro<4, 1, 16>(q, stream);
    return m == 4 ? launch_ro<4, 4>(q, stream) : (m == 2 ? launch_ro<4, 2>(q, stream) : launch_ro<4, 1>(q, stream));
  }
  return m == 4 ? launch_ro<2, 4>(q, stream) : (m == 2 ? launch_ro<2, 2>(q, stream) : launch_ro<2, 1>(q, stream));
}

}  // namespace vspconv

// Winograd F(2x2, 3x3) convolution on fp32 MFMA for gfx950: the stride-1 3x3 layers of the path (StyledConv, SMART fusion
// and SMART dilated branches, IR-SE convolutions) with 16 multiplies per 2x2 output tile instead of 36.
//
// The direct kernel (conv_kernel.h) is bound by the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 issues at 1/4 of the bf16
// rate): its big layers sit at 118 TFLOP/s of a 131 TFLOP/s MFMA-only ceiling.  The only way past that at fp32 is fewer
// multiplies.  With V = B^T d B (input tile 4x4), U = G g G^T (weights, precomputed once per layer), M = sum_ci U . V per
// Winograd position and Y = A^T M A, a 2x2 output tile costs 16 MACs per (co, ci) -- sixteen independent (Cout x Cin) x
// (Cin x tiles) GEMMs on MFMA -- plus transforms that are additions only.
//
// One workgroup (8 waves) owns 16 MBW output channels x 16 NBW tiles of one image; wave w owns the Winograd positions 2w
// and 2w+1.  What shapes the kernel (ablations on the first version: the global loads were 31 % of the time, LDS writes 5 %,
// the transform 5 %, and nothing of it overlapped the MFMAs because three barriers per chunk kept the eight waves in lock
// step):
//   * U never touches LDS.  No two waves share a position, so the host stores U in FRAGMENT order
//     [group][co tile][chunk][wave][pp][lane][mb] and a wave fetches its A fragments of a chunk as one contiguous run per
//     position, one chunk ahead, straight into the registers the MFMAs read.
//   * the input patch is prefetched two chunks ahead (it streams from MALL/HBM) and double-buffered in LDS, V is
//     double-buffered too: the transform of chunk i+1, the MFMAs of chunk i and the patch write of chunk i+2 share ONE
//     barrier interval (4 input channels per chunk keep all of it inside 128 VGPRs: two workgroups per CU).
//   * dilation d (the SMART branches) is polyphase in ROWS only: a workgroup owns rows ry, ry + d, ... (consecutive patch rows)
//     and a DENSE run of 2 TLX columns with a halo of d columns; its TLX tile columns are the d column residues x TLX / d tile
//     positions (tile tx: residue tx % d, first column residue + 2 d (tx / d)), and the transform reads its four window columns d
//     apart.  Global loads and stores stay whole row segments for every dilation (a column-polyphase form touches a 32-byte
//     sector per 4-byte element at d = 8).  The four dilation groups of a launch differ only in d and in their weight / channel
//     base.  Template DMAX = the largest dilation the patch buffer is sized for (1: ordinary layers, 8: SMART branches).
// Epilogue, per 16-channel block: the sixteen position accumulators meet in LDS, one thread per (channel, tile) applies
// A^T . A, the same fused operand chain as the direct kernel (demod, bias, two activations, noise, two residuals) and stores
// the 2x2 pixels.  Numerics: F(2x2,3x3) in fp32 adds ~1e-6 relative error (transform constants are 1 and 1/2).
#include "conv_kernel.h"
#include <type_traits>

namespace vspconv {

namespace {

// Wave-uniform operand through the scalar cache, whatever the compiler can prove about the index: the constant address space
// makes the load an s_load (lgkmcnt).  As a per-lane global load it joins vmcnt and drags the latency of every prefetch in
// flight into the interval (measured: 674 -> 802 us on 512 -> 512 at 64^2 when an unrelated edit flipped the compiler's choice).
__device__ __forceinline__ float uload(const float* base, int idx) {
  typedef const float __attribute__((address_space(4))) * cfp4;
  return ((cfp4)(uintptr_t)base)[__builtin_amdgcn_readfirstlane(idx)];
}

// (round 3, tried: delaying the second half of the first generation of workgroups by 4-15 us so that the two resident workgroups of a
//  CU run out of phase -- prologue / epilogue of one under the MFMA intervals of the other: no effect on 64 / 128 / 512 channels
//  (1029 / 868 / 771 us with and without, tools/wino_ablate.py).  Scheduler strategies of the compiler (-mllvm -amdgpu-sched-strategy=
//  max-ilp | max-memory-clause | iterative-ilp | iterative-maxocc) and -O2: within 2 % of -O3 either way on the five judged shapes.)
#ifndef VSP_WINO_PIN
#define VSP_WINO_PIN 1   // pinned steady-state schedule for the undilated kernel (measured +1.5..3.5 %); dilated: slower, off
#endif
constexpr int WCK = 4;      // input channels per chunk = one MFMA k-step
constexpr int NTHR = 512;

template <int MBW, int DMAX>
struct WG {  // workgroup geometry: MBW 16-channel blocks x NBW 16-tile blocks (MBW * NBW = 8: 64 accumulator registers)
  // input channels per barrier interval: two MFMA k-steps where the LDS budget allows it (32-tile geometry: 61 KB, two
  // workgroups per CU) -- half the barriers and a transform task for every thread; one k-step otherwise
  static constexpr int IVC = MBW == 4 ? 8 : 4;
  static constexpr int KS = IVC / 4;
  static constexpr int NBW = 8 / MBW;
  static constexpr int WCO = 16 * MBW;
  static constexpr int NTILE = 16 * NBW;
  static constexpr int TLX = NBW == 8 ? 16 : 8;
  static constexpr int TLY = NTILE / TLX;
  static constexpr int PR = 2 * TLY + 2, PC = 2 * TLX + 2 * DMAX;  // PC: the widest patch row (dilation DMAX)
  // Patch ROW pitch (round 3, bank conflicts).  The transform reads a task's window as 16-byte rows (ds_read2_b64: 16-lane groups,
  // bank = dword mod 32); a group holds the 8 tile columns of TWO tile rows (window rows 2 apart), so with 8-tile-wide geometries
  // the row pitch must put two rows 16 banks apart: pitch == 8 (mod 16) -> 24 for the 18-word rows (the pitch-18 image made every
  // transform read a 2-way conflict: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.39).  16-tile-wide geometries fill all 32 banks
  // from one row and keep the dense image; so do the dilated variants (their row width depends on the group's dilation).
  static constexpr int PCP = (DMAX == 1 && TLX == 8) ? 24 : PC;
  static constexpr int PPITCH = (PR * PCP + 15) / 16 * 16;
  static constexpr int VPITCH = NTILE + 16;           // k-slot rows 16 banks apart
  static constexpr int LDS_V = 16 * IVC * VPITCH;     // floats, two buffers
  static constexpr int LDS_P = IVC * PPITCH;          // floats, two buffers
  static constexpr int ETILE = NTILE > 64 ? 64 : NTILE;  // tiles per epilogue pass
  static constexpr int EMB = (MBW >= 2 && ETILE <= 32) ? 2 : 1;  // 16-channel blocks per epilogue pass
  static constexpr int EP = ETILE + 4;                // epilogue row pitch: 4 rows (one k-slot group) = 16 banks
  static constexpr int LDS_M = 16 * 16 * EMB * EP;
  static constexpr int LDS_STAGE = 2 * LDS_V + 2 * LDS_P;
  static constexpr int LDS_FLOATS = LDS_STAGE > LDS_M ? LDS_STAGE : LDS_M;
  static constexpr int UF = 2 * MBW;                  // U floats per lane and chunk: [pp 2][mb MBW]
};

template <int MBW, int DMAX>
__global__ __launch_bounds__(NTHR, 4) void conv_wino_kernel(const ConvK p) {
  using Gm = WG<MBW, DMAX>;
  constexpr int NBW = Gm::NBW, WCO = Gm::WCO, NTILE = Gm::NTILE, TLX = Gm::TLX, TLY = Gm::TLY, PR = Gm::PR;
  constexpr int PPITCH = Gm::PPITCH, VPITCH = Gm::VPITCH, LDS_V = Gm::LDS_V, LDS_P = Gm::LDS_P, UF = Gm::UF;
  constexpr int IVC = Gm::IVC, KS = Gm::KS;
  constexpr bool PADROW = Gm::PCP != Gm::PC;        // padded patch rows (undilated 8-tile-wide geometries)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vl = smem;               // 2 x [16][IVC][VPITCH]
  float* Pl = smem + 2 * LDS_V;   // 2 x [IVC][PPITCH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  // XCD-aware work order.  The dispatcher deals workgroups round-robin over the 8 XCDs (each with its own 4 MB L2), so in
  // dispatch order every L2 sees every pixel tile and every channel tile.  Bijective remap: XCD x walks a CONTIGUOUS range of
  // the work list, ordered either
  //   1 = pixel-tile-major (image, pixel tile, channel tile): neighbouring tiles share their halo (40 % of a 10 x 18 patch)
  //       and the channel tiles of one pixel tile read the same patch -- right when U is small (<= 64 channels), or
  //   2 = channel-tile-major (channel tile, image, pixel tile): a workgroup streams its whole U slice (64 co x Cin x 16
  //       positions x 4 B = 2 MB at 512 channels) and NO two waves share any of it, so U is the kernel's dominant fetch
  //       (32 KB per 8-channel interval against 5.8 KB of patch: 4.3 GB per 512 -> 512 launch at 64^2, ~5 TB/s).  In
  //       pixel-major order all 8 channel tiles (16.8 MB) compete for one 4 MB L2 and U streams from MALL / HBM; in
  //       channel-major order an XCD works on one or two channel tiles at a time and U stays L2-resident.
  //   4 = region-major (shared-input dilation groups, as in conv_bf16.hip): a region = one image band of 8 x 2 TLY rows x 4 column
  //       tiles; for every dilation d | 8 exactly 8 workgroups per column tile cover it (d residues x 8/d row tiles), and the four
  //       groups of a region run back to back on one XCD instead of never meeting in an L2.
  int b = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  int reg_ry = -1, reg_ty = 0, reg_tx = 0;
  if (DMAX > 1 && p.wg_order == 4) {
    constexpr int CGX = 4;
    const int GX = gridDim.x, GY = gridDim.y, GZ = gridDim.z, GT = GX * GY * GZ;
    const int wgid = blockIdx.x + GX * (blockIdx.y + GY * blockIdx.z);
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    const int nb = p.tiles_y, ncg = p.tiles_x;                 // (host: bands per image, column groups per band)
    const int per_region = GY * 8 * CGX;
    const int region = lid / per_region, w = lid - region * per_region;
    b = region / (nb * ncg);
    const int rr = region - b * (nb * ncg);
    const int band = rr / ncg, cg = rr - band * ncg;
    const int slot = w / (GY * CGX), w2 = w - slot * (GY * CGX);
    const int cx = w2 / GY;
    by = w2 - cx * GY;
    const int dg = p.dil[by / p.co_tiles];
    reg_ry = slot % dg;
    reg_ty = band * (8 / dg) + slot / dg;
    reg_tx = cg * CGX + cx;
  } else if (p.wg_order) {
    const int GX = gridDim.x, GY = gridDim.y, GZ = gridDim.z, GT = GX * GY * GZ;
    const int wgid = blockIdx.x + GX * (blockIdx.y + GY * blockIdx.z);
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    if (p.wg_order == 1) {
      const int GN = GX * GY;
      b = lid / GN;
      const int lrem = lid - b * GN;
      bx = lrem / GY;
      by = lrem - bx * GY;
    } else {
      const int GN = GX * GZ;
      by = lid / GN;
      const int lrem = lid - by * GN;
      b = lrem / GX;
      bx = lrem - b * GX;
    }
  }
  const int g = by / p.co_tiles, ct = by - g * p.co_tiles;
  const int d = DMAX == 1 ? 1 : p.dil[g];                     // row-polyphase stride and column tap spacing (1, 2, 4 or 8)
  const int SH = (p.H + d - 1) / d;                           // rows of one residue class
  const int tiles_x = (p.W + 2 * TLX - 1) / (2 * TLX), tiles_y = (SH + 2 * TLY - 1) / (2 * TLY);
  const int per_res = tiles_x * tiles_y;
  int ry, tx_i, ty_i;
  if (reg_ry >= 0) {
    if (reg_ty >= tiles_y || reg_tx >= tiles_x) return;       // (bands / column groups that the image does not fill)
    ry = reg_ry; ty_i = reg_ty; tx_i = reg_tx;
  } else {
    if (bx >= per_res * d) return;                            // (row counts that d does not divide leave a few spare blocks)
    ry = bx / per_res;
    const int tile_i = bx - ry * per_res;
    tx_i = tile_i % tiles_x;
    ty_i = tile_i / tiles_x;
  }
  const int oy0 = ty_i * (2 * TLY), ox0 = tx_i * (2 * TLX);   // sub-image rows, image columns
  const int PC = 2 * TLX + 2 * d;                             // patch row of this group
  const int PCP = PADROW ? Gm::PCP : PC;                      // its pitch in LDS
  const int co0 = ct * WCO;                                   // within the group
  const int chw = p.H * p.W;
  const float* xb = p.x + (int64_t)b * p.x_ch * chw;
  const int nchunk = (p.Cin + IVC - 1) / IVC;        // barrier intervals
  const int nchunk4 = (p.Cin + WCK - 1) / WCK;       // 4-channel U chunks (the weight layout's unit)

  // ---- input patch: chunk-invariant geometry, values prefetched two intervals ahead.  NTHR / IVC threads walk one
  //      channel's plane: the channel is wave-uniform and no index needs a division per step.
  const int PLANE = PR * PC;
  constexpr int PTH = NTHR / IVC;                  // threads per channel plane
  constexpr int PLD = (PR * Gm::PC + PTH - 1) / PTH;
  const int p_ch = __builtin_amdgcn_readfirstlane(tid / PTH), p_t = tid % PTH;  // (PTH is a multiple of 64: wave-uniform)
  // per word e of this thread: byte offset inside the channel image (0 when outside), a validity bit, and -- with padded rows --
  // its LDS word (row r of the plane lives at r * PCP)
  int p_voff[PLD];
  int p_dst[PADROW ? PLD : 1];
  unsigned p_in = 0;
#pragma unroll
  for (int e = 0; e < PLD; ++e) {
    const int rem = p_t + PTH * e;
    const int r = rem / PC, c = rem - r * PC;
    const int sy = oy0 - 1 + r, ix = ox0 - d + c;
    const int iy = sy * d + ry;
    const bool in = rem < PLANE && sy >= 0 && ix >= 0 && iy < p.H && ix < p.W;
    p_voff[e] = in ? (iy * p.W + ix) * 4 : 0;
    p_in |= in ? (1u << e) : 0u;
    if constexpr (PADROW) p_dst[e] = p_ch * PPITCH + r * Gm::PCP + c;
  }
  if constexpr (!PADROW) p_dst[0] = p_ch * PPITCH + p_t;
  // buffer loads: resource = this image, scalar offset = the channel plane, vector offset = the lane's byte offset in a plane
  // (flat pointers cost a 64-bit VALU add per load and the registers of the 64-bit lane addresses)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, 0x7fffffff, 0x00020000);
  float preg[PLD], pnext[PLD];
  auto issue_p = [&](int c) {  // chunk c -> pnext (raw values: nothing may consume them before the commit two steps later)
    const int ci = c * IVC + p_ch;
    const int soff = (ci < p.Cin ? ci : 0) * chw * 4;
#pragma unroll
    for (int e = 0; e < PLD; ++e) pnext[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, p_voff[e], soff, 0));
  };
  auto commit_p = [&](float* Pdst, int c) {  // preg = chunk c.  Affine input (folded BatchNorm): the shift belongs to in-image
    const int ci = c * IVC + p_ch;             // pixels only, so it is applied here (plain layers: scale 1, shift 0 from the constants)
    const bool chok = ci < p.Cin;
    const int cc = chok ? ci : p.Cin - 1;
    const float sc = uload(p.wcp, b * p.wc_bs + cc * p.wc_cs), sh = uload(p.wshp, cc * p.wsh_cs);
#pragma unroll
    for (int e = 0; e < PLD; ++e)
      if (PLD * PTH == PLANE || p_t + PTH * e < PLANE)
        Pdst[PADROW ? p_dst[e] : p_dst[0] + PTH * e] = (((p_in >> e) & 1u) && chok) ? fmaf(preg[e], sc, sh) : 0.f;
  };

  // ---- U fragments: [group][co tile][chunk][wave][pp 2][lane][mb MBW] floats (round 4: one contiguous run per wave-wide load)
  const float* ufr = p.w + ((((int64_t)g * p.co_tiles + ct) * nchunk4 * 8 + wave) * 2 * 64 + lane) * MBW;
  auto load_u = [&](int c, float (&u)[UF]) {
    const float* src = ufr + (int64_t)c * (8 * 64 * UF);
    if constexpr (MBW == 4) {
      const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 64 * 4);
      u[0] = a.x; u[1] = a.y; u[2] = a.z; u[3] = a.w; u[4] = bq.x; u[5] = bq.y; u[6] = bq.z; u[7] = bq.w;
    } else if constexpr (MBW == 2) {
      const float2 a = *reinterpret_cast<const float2*>(src), bq = *reinterpret_cast<const float2*>(src + 64 * 2);
      u[0] = a.x; u[1] = a.y; u[2] = bq.x; u[3] = bq.y;
    } else {
      u[0] = src[0]; u[1] = src[64];
    }
  };

  // ---- transform: task = (channel, tile, half); half h produces V rows 2h, 2h+1 of the tile's 4x4 window.
  // Task -> lane (round 3): 32 consecutive tiles of ONE half per half-wave (tile = task bits 0-4 and 6.., half = bit 5).  The V
  // stores of a 32-lane group then hit 32 consecutive words (with the half in bit 0, lanes 2k and 2k+1 wrote the same bank), and
  // a 16-lane group of the window reads covers 8 tile columns x 2 tile rows -- conflict-free on the padded patch rows (Gm::PCP).
  auto task_half = [](int task) { return (task >> 5) & 1; };
  auto task_tile = [](int task) { return (task & 31) | (((task >> 6) & (NTILE / 32 - 1)) << 5); };
  constexpr int TASKS = 2 * IVC * NTILE;
  constexpr int TPT = (TASKS + NTHR - 1) / NTHR;
  static_assert(TASKS % NTHR == 0, "every thread runs TPT transform tasks");
  float tdd[TPT][3][4];  // rows h, h+1, h+2 of each task's 4x4 window (columns d apart)
  auto transform_read = [&](const float* Psrc) {  // the LDS reads of the transform: issued early, consumed under the MFMAs
#pragma unroll
    for (int it = 0; it < TPT; ++it) {
      const int task = tid + it * NTHR;
      const int th = task_half(task), t_tile = task_tile(task);
      const int t_ch = __builtin_amdgcn_readfirstlane(task / (2 * NTILE));
      const int t_ty = t_tile / TLX, t_tx = t_tile - t_ty * TLX;
      // tile column t_tx: column residue t_tx % d, position t_tx / d -> first window column (patch coordinates) rx + 2 d pos
      const int c0 = DMAX == 1 ? 2 * t_tx : (t_tx % d) + 2 * d * (t_tx / d);
      const float* src = Psrc + t_ch * PPITCH + (2 * t_ty + th) * PCP + c0;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) tdd[it][r][cc] = src[r * PCP + cc * d];
    }
  };
  auto transform_write = [&](float* Vdst, int c) {  // chunk c: V = B^T d B, the style scale rides on V
#pragma unroll
    for (int it = 0; it < TPT; ++it) {
      const int task = tid + it * NTHR;
      const int th = task_half(task), t_tile = task_tile(task);
      // (2 NTILE tasks per channel, a multiple of 64: the channel is wave-uniform -> the style scale comes through the scalar
      //  cache on lgkmcnt; as a per-lane global load it put a vmcnt(0) -- i.e. the latency of the patch and U prefetches just
      //  issued -- into every interval)
      const int t_ch = __builtin_amdgcn_readfirstlane(task / (2 * NTILE));
      const int ci = c * IVC + t_ch;
      const float sc = uload(p.wtp, b * p.wt_bs + (ci < p.Cin ? ci : p.Cin - 1) * p.wt_cs);
      const auto& dd = tdd[it];
      float w0[4], w1[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        // h = 0: W0 = d0 - d2, W1 = d1 + d2      h = 1: W2 = d2 - d1, W3 = d1 - d3  (dd[0..2] = window rows h..h+2)
        w0[cc] = (th ? dd[1][cc] - dd[0][cc] : dd[0][cc] - dd[2][cc]) * sc;
        w1[cc] = (th ? dd[0][cc] - dd[2][cc] : dd[1][cc] + dd[2][cc]) * sc;
      }
      const float v0[4] = {w0[0] - w0[2], w0[1] + w0[2], w0[2] - w0[1], w0[1] - w0[3]};
      const float v1[4] = {w1[0] - w1[2], w1[1] + w1[2], w1[2] - w1[1], w1[1] - w1[3]};
      float* dst = Vdst + t_ch * VPITCH + t_tile;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        dst[((2 * th) * 4 + nu) * (IVC * VPITCH)] = v0[nu];
        dst[((2 * th + 1) * 4 + nu) * (IVC * VPITCH)] = v1[nu];
      }
    }
  };
  auto transform = [&](const float* Psrc, float* Vdst, int c) {
    transform_read(Psrc);
    transform_write(Vdst, c);
  };

  f32x4 acc[2][MBW][NBW];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[pp][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply_pp = [&](const float* Vsrc, int ks, int pp, const float (&u)[UF]) {  // one position of k-step ks: MBW x NBW MFMAs
    const float* vp = Vsrc + (2 * wave + pp) * (IVC * VPITCH) + (4 * ks + kq) * VPITCH + lr;
    float bv[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bv[nb] = vp[nb * 16];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
        acc[pp][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pp * MBW + mb], bv[nb], acc[pp][mb][nb], 0, 0, 0);
  };
  auto multiply = [&](const float* Vsrc, int ks, const float (&u)[UF]) {  // k-step ks of the interval: channels 4 ks + kq
    multiply_pp(Vsrc, ks, 0, u);
    multiply_pp(Vsrc, ks, 1, u);
  };

  // ---- pipeline.  State at the top of interval i:  Vl[i&1] = V(i), Pl[(i+1)&1] = patch(i+1), ua = U chunk KS*i (registers,
  //      landed), pnext = patch(i+2) (in flight or landed).  U rolls one 4-channel k-step ahead of the MFMAs that use it.
  float ua[UF], ub[UF];
  issue_p(0);
  load_u(0, ua);
#pragma unroll
  for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];
  commit_p(Pl, 0);                                // patch(0) -> Pl[0]
  if (nchunk > 1) issue_p(1);
  __syncthreads();
  transform(Pl, Vl, 0);                           // V(0) -> Vl[0]
  if (nchunk > 1) {
#pragma unroll
    for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];
    commit_p(Pl + LDS_P, 1);                      // patch(1) -> Pl[1]
    if (nchunk > 2) issue_p(2);
  }
  __syncthreads();
  // The steady-state intervals carry no tail guards: with them every `if (i + k < nchunk)` is a branch whose join makes the
  // compiler's waitcnt pass assume the worst of both paths (vmcnt(0) where a counted wait would do).
#ifdef VSP_WINO_ABLATE  // tuning only (VSP_CONV_DBG): 0x400 no transform, 0x800 no U loads, 0x1000 no patch loads, 0x2000 no MFMAs, 0x4000 no commit
  const int ab = p.dbg;
#else
  constexpr int ab = 0;
#endif
  auto interval = [&](int i, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int cur = i & 1, nxt = cur ^ 1;
    if (KS == 2 && (FULL || KS * i + 1 < nchunk4) && !(ab & 0x800)) load_u(KS * i + 1, ub);
#if VSP_WINO_PIN
    if constexpr (FULL && KS == 2 && TPT == 1 && DMAX == 1) __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x80 | 0x100 | 0x200);
#endif
    if (FULL || i + 2 < nchunk) {
#pragma unroll
      for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];   // patch(i+2), issued one interval ago
    }
    if ((FULL || i + 3 < nchunk) && !(ab & 0x1000)) issue_p(i + 3);
    // program order inside the wave: the transform's LDS reads go out first, its arithmetic and LDS writes follow the first
    // k-step's MFMAs -- an in-order wave overlaps only what sits between its own MFMAs (the matrix pipe takes one every 32 cycles)
    // (the 128-tile geometry runs two tasks per thread: 24 window registers across the MFMAs would spill -- it transforms first).
    // Tried on top of this order and dropped: groups of MBW x NBW MFMAs with the transform pinned between them by scheduling
    // fences (718 us on 512 -> 512 at 64^2 against 688: a fenced block of VALU work leaves the pipe to the other waves only) and
    // sched_group_barrier patterns (one MFMA : two VALU), which the scheduler does not honour across the LDS waits; an
    // anti-phase start of the two workgroups of a CU (s_sleep of half an interval for every other group of 32): no effect.
    // Round 3: OPPOSITE interval orders for the two waves a workgroup places on one SIMD (waves 0-3 multiply first, waves 4-7
    // transform and commit first; every order is legal inside an interval).  As one loop with both orders hipcc spills the
    // accumulators; as two loops the younger half still needs 21 spill slots at 128 VGPRs (512 -> 512 at 64^2: 652 -> 827 us), and
    // with a 256-register budget (one workgroup per CU) the opposite orders are SLOWER than the common one (826 vs 765 us): the
    // transform placed in front of the MFMAs exposes its LDS round trip in every interval, which costs more than the partner's
    // MFMAs cover.  Also measured and dropped: the interval's scalar operands (style scale, affine pair) fetched at its top behind a
    // full scheduling fence instead of right in front of their `s_waitcnt` -- 2 % slower on every layer (655 -> 666 us): the fence
    // costs the scheduler more freedom than the exposed scalar-cache round trips cost time.  And the 64-tile geometry (MBW x NBW = 16:
    // 128 accumulator registers, one workgroup per CU, 256-register budget, same U layout): with this interval program 771 us against
    // 649 on 512 -> 512 at 64^2 (the one-workgroup penalty of the 32-tile kernel, 765, is all it gets back), and as a PINNED sequence
    // of 16 steps -- four MFMAs + one slice of transform / fragment reads / commit, a full fence per step, the way conv_pipe.hip runs
    // the direct convolution -- 1021 us: the slices carry LDS and scalar-cache waits that a fence turns into pipe idle time (the
    // direct kernel's slices are stores and address-free loads).
    constexpr bool SPLIT = TPT == 1;
#if VSP_WINO_PIN
    if constexpr (FULL && KS == 2 && SPLIT && DMAX == 1) {
      // Steady state with the vector-memory issue points pinned INSIDE the MFMA stream (mask 0x386: VALU, SALU and LDS operations
      // may cross a fence, MFMAs and vector-memory instructions may not): the U loads of the second k-step and the patch loads of
      // interval i + 3 leave after the first MFMA groups instead of all before the first one.
      constexpr int M_ = 0x2 | 0x4 | 0x80 | 0x100 | 0x200;
      multiply_pp(Vl + cur * LDS_V, 0, 0, ua);
      __builtin_amdgcn_sched_barrier(M_);
      transform_read(Pl + nxt * LDS_P);
      multiply_pp(Vl + cur * LDS_V, 0, 1, ua);
      __builtin_amdgcn_sched_barrier(M_);
      load_u(KS * i + 2, ua);
      transform_write(Vl + nxt * LDS_V, i + 1);
      multiply_pp(Vl + cur * LDS_V, 1, 0, ub);
      __builtin_amdgcn_sched_barrier(M_);
      multiply_pp(Vl + cur * LDS_V, 1, 1, ub);
      commit_p(Pl + cur * LDS_P, i + 2);
      __syncthreads();
      return;
    }
#endif
    const bool tr = (FULL || i + 1 < nchunk) && !(ab & 0x400);
    if (tr) transform_read(Pl + nxt * LDS_P);
    if (tr && !SPLIT) transform_write(Vl + nxt * LDS_V, i + 1);
    if (KS == 2) {
      if (!(ab & 0x2000)) multiply(Vl + cur * LDS_V, 0, ua);
      if (tr && SPLIT) transform_write(Vl + nxt * LDS_V, i + 1);
      if ((FULL || KS * i + 2 < nchunk4) && !(ab & 0x800)) load_u(KS * i + 2, ua);  // next interval's first k-step (the MFMAs above have read ua)
      if ((FULL || KS * i + 1 < nchunk4) && !(ab & 0x2000)) multiply(Vl + cur * LDS_V, 1, ub);
    } else {
      if ((FULL || i + 1 < nchunk4) && !(ab & 0x800)) load_u(i + 1, ub);
      if (!(ab & 0x2000)) multiply(Vl + cur * LDS_V, 0, ua);
      if (tr && SPLIT) transform_write(Vl + nxt * LDS_V, i + 1);
#pragma unroll
      for (int q = 0; q < UF; ++q) ua[q] = ub[q];
    }
    if ((FULL || i + 2 < nchunk) && !(ab & 0x4000)) commit_p(Pl + cur * LDS_P, i + 2);  // Pl[cur] held patch(i): consumed one interval ago
    __syncthreads();
  };
  int iv = 0;
  // FULL needs i + 3 < nchunk and every 4-channel U chunk of intervals i, i + 1 present (Cin a multiple of IVC)
  const int n_full = (p.Cin % IVC == 0) ? nchunk - 3 : 0;
  for (; iv < n_full; ++iv) interval(iv, std::true_type{});
  for (; iv < nchunk; ++iv) interval(iv, std::false_type{});
#ifdef VSP_WINO_ABLATE
  if (ab & 0x8000) return;
#endif

  // ---- epilogue: per 16-channel block and (at most) 64 tiles, all sixteen positions through LDS, one thread per
  //      (channel, tile): Y = A^T M A, then the fused operand chain of the direct kernel
  constexpr int ETILE = Gm::ETILE, ENB = ETILE / 16, EP = Gm::EP;
  float* Ml = smem;  // [16 pos][16 co][EP]: rows padded so that the four k-slot groups of a store land 16 banks apart
  const int Cout = p.G * p.cout_g;
  const float* osp = p.osp + (int64_t)b * Cout * p.oss;
  const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW * p.nzs;
  const float nw = p.nwp[0];
  float* yb = p.y + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const float* r1b = p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r1s;
  const float* r2b = p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r2s;
  const int y_plane = p.y_h * p.y_w;
  constexpr int EMB = Gm::EMB, ECO = 16 * EMB;
  constexpr int EPT = ECO * ETILE / NTHR;  // (channel, tile) pairs per thread and pass
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  const bool vec2 = d == 1 && p.r1s <= 1 && p.r2s <= 1;  // the two pixels of a tile row are neighbours in memory
  const bool pairs = vec2 && (p.OW & 1) == 0 && p.OW >= 2;
#pragma unroll
  for (int mb0 = 0; mb0 < MBW; mb0 += EMB) {
#pragma unroll
    for (int th = 0; th < NTILE / ETILE; ++th) {  // tile halves (only the 128-tile geometry has two)
      if (mb0 + th > 0) __syncthreads();           // (the chunk loop ended on a barrier)
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
#pragma unroll
      for (int it = 0; it < EPT; ++it) {
        const int pair = tid + it * NTHR;
        const int e_co = pair / ETILE, e_t = pair - e_co * ETILE;
        const int e_tile = th * ETILE + e_t;
        const int e_tx = e_tile % TLX;
        const int sy = oy0 + 2 * (e_tile / TLX);
        const int sx = ox0 + (DMAX == 1 ? 2 * e_tx : (e_tx % d) + 2 * d * (e_tx / d));   // first output column of the tile
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
        // No load sits behind a divergent branch (a ragged channel tile, an edge tile): coordinates are clamped, only the STORE is
        // predicated.  With `continue` / edge tests in front of them the compiler waited for every load in flight at each join --
        // the six per-channel operands and the noise / residual pairs of a thread left one round trip after the other.
        const int cgi = co0 + mb0 * 16 + e_co;  // channel within the group
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
        if (pairs) {   // (uniform) even output width: a tile's two pixels are a whole 8-byte pair or lie outside together
          f32x2u nz[2] = {{0.f, 0.f}, {0.f, 0.f}}, r1v[2] = {{0.f, 0.f}, {0.f, 0.f}}, r2v[2] = {{0.f, 0.f}, {0.f, 0.f}};
          int ro[2];
          bool inside[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = (sy + i) * d + ry;
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
            const int oy = (sy + i) * d + ry, ox = sx;
            if (!cok || oy >= p.OH || ox >= p.OW) continue;
            const int ro = cbase + oy * p.y_w + ox;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int oxj = ox + j * d;
              if (oxj >= p.OW) continue;
              const int rj = ro + j * d;
              yb[rj] = fin(yv[i][j], nzp[(oy * p.OW + oxj) * p.nzs], r1b[rj * p.r1s], r2b[rj * p.r2s]);
            }
          }
        }
      }
    }
  }
}

template <int MBW, int DMAX>
int launch_variant(ConvK q, hipStream_t stream) {
  using Gm = WG<MBW, DMAX>;
  static vsp::LdsAttrOnce attr;   // per device
  const size_t lds = (size_t)Gm::LDS_FLOATS * sizeof(float);
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wino_kernel<MBW, DMAX>), (int)lds, "conv2d_winograd")) return rc;
  q.co_tiles = (q.cout_g + Gm::WCO - 1) / Gm::WCO;
  int blocks = 0;  // the largest per-group tile count (groups with a smaller dilation exit early)
  for (int g = 0; g < q.G; ++g) {
    const int d = q.dil[g];
    const int SH = (q.H + d - 1) / d;
    const int n = ((q.W + 2 * Gm::TLX - 1) / (2 * Gm::TLX)) * ((SH + 2 * Gm::TLY - 1) / (2 * Gm::TLY)) * d;
    blocks = n > blocks ? n : blocks;
  }
  // work order (see the kernel): channel-tile-major once the launch's U no longer fits an L2 beside the patches
  const int64_t u_bytes = (int64_t)q.G * q.co_tiles * Gm::WCO * ((q.Cin + 3) / 4 * 4) * 16 * 4;
  (void)u_bytes;  // measured: the order makes no difference on the big-U layers (512 -> 512 at 64^2: 772 us either way), pixel-major
                  // wins on 32 -> 32 at 1024^2 (1613 -> 1502 us) and on the dilation groups of small maps (512 -> 4 x 128 at 32^2: 437 -> 341 us)
  q.wg_order = (DMAX == 1 && q.G == 1) || q.H * q.W <= 1024 ? 1 : 0;
  if (q.dbg & 0x300) q.wg_order = (q.dbg >> 8) & 3;  // tuning: VSP_CONV_DBG = 256 / 512 / 768 forces order 1 / 2 / dispatch (0)
  if (q.wg_order == 3) q.wg_order = 0;
  if (DMAX > 1 && q.G >= 2 && q.x_gs == 0 && !(q.dbg & 0x800000)) {
    bool ok = true;
    for (int g = 0; g < q.G; ++g) ok = ok && (q.dil[g] == 1 || q.dil[g] == 2 || q.dil[g] == 4 || q.dil[g] == 8);
    if (ok) {  // region-major order over bands of 8 x 2 TLY rows x 4 column tiles
      q.wg_order = 4;
      q.tiles_y = (q.H + 16 * Gm::TLY - 1) / (16 * Gm::TLY);
      q.tiles_x = ((q.W + 2 * Gm::TLX - 1) / (2 * Gm::TLX) + 3) / 4;
      blocks = q.tiles_y * q.tiles_x * 32;
    }
  }
  dim3 grid((unsigned)blocks, (unsigned)(q.co_tiles * q.G), (unsigned)q.B);
  conv_wino_kernel<MBW, DMAX><<<grid, NTHR, lds, stream>>>(q);
  return VSP_OK;
}

}  // namespace

int wino_chunk() { return WCK; }

// 16-channel blocks per workgroup for a layer with cout_g output channels per group (the weight layout depends on it)
int wino_mbw(int cout_g) { return cout_g > 32 ? 4 : (cout_g > 16 ? 2 : 1); }

int wino_launch(ConvK q, int form, hipStream_t stream) {
  // form (vsp_conv_params.tile_hint of the Winograd entry): 0 = automatic; 1 = the task-list kernel of this file; 2 = the row-owner forms
  // (conv_wino_ro.hip / conv_wino_rod.hip); 3 = the register-resident-U form (conv_wino_rs.hip); a named form that does not serve the
  // launch is VSP_ENOTSUP
  int dmax = 1;
  for (int g = 0; g < q.G; ++g) dmax = q.dil[g] > dmax ? q.dil[g] : dmax;
  if (form == 3) {
    if (!wino_rs_eligible(q)) return vsp::fail(VSP_ENOTSUP, "conv2d_winograd: the register-resident-U form does not serve this launch");
    return wino_rs_launch(q, stream);
  }
  if (form == 0) {
    static const int rs = vsp::tune_env("VSP_WINO_RS") ? atoi(vsp::tune_env("VSP_WINO_RS")) : 1;   // 0: never, 2: wherever eligible
    if (rs && wino_rs_eligible(q) && (rs == 2 || wino_rs_profitable(q))) return wino_rs_launch(q, stream);
  }
  if (dmax == 1) {
    // row-owner form (conv_wino_ro.hip) wherever it serves the launch; maps up to 16 x 16 measured equal or slower (512 -> 512 at 16^2:
    // 99 vs 102 us) and stay here.  VSP_WINO_RO = 0 keeps this file's kernel everywhere, 1 / 2 / 4 = barrier period of the other one
    // (8-channel sub-stages; 2 and 4 measured within 1 % of 1 on the deep layers, slower on the shallow ones).
    static const int ro = vsp::tune_env("VSP_WINO_RO") ? atoi(vsp::tune_env("VSP_WINO_RO")) : 1;
    if (form == 2 && !wino_ro_eligible(q)) return vsp::fail(VSP_ENOTSUP, "conv2d_winograd: the row-owner form does not serve this launch");
    if (form != 1 && (ro == 1 || ro == 2 || ro == 4) && (q.H * q.W > 256 || form == 2) && wino_ro_eligible(q))
      return wino_ro_launch(q, wino_mbw(q.cout_g), ro == 0 ? 1 : ro, stream);
    switch (wino_mbw(q.cout_g)) {
      case 4: return launch_variant<4, 1>(q, stream);
      case 2: return launch_variant<2, 1>(q, stream);
      default: return launch_variant<1, 1>(q, stream);
    }
  }
  {   // dilation groups: row-owner form (conv_wino_rod.hip) wherever it serves the launch (>= 32 channels per group) -- round 5, one box, forms
      // named through tile_hint (profiles/r05_ab_rod_dense.log): 128 -> 4 x 32 at 256^2 1066 -> 952 us, 256 -> 4 x 64 at 128^2 774 -> 704,
      // 512 -> 4 x 128 at 64^2 740 -> 667 (round 4 had measured that one equal and kept it here), at 32^2 296 -> 261.  VSP_WINO_ROD = 0: never.
      // (A dense epilogue -- the tile's scattered pixels through LDS, 16-byte stores -- was measured on this kernel in round 5: no difference,
      //  not kept; its workgroups live 16-64 stages and the epilogue of one hides under the main loop of the other.)
    static const int rod = vsp::tune_env("VSP_WINO_ROD") ? atoi(vsp::tune_env("VSP_WINO_ROD")) : 1;
    const bool rod_ok = wino_mbw(q.cout_g) >= 2 && wino_rod_eligible(q);
    if (form == 2 && !rod_ok) return vsp::fail(VSP_ENOTSUP, "conv2d_winograd: the row-owner form does not serve this launch");
    if (form != 1 && rod_ok && (form == 2 || rod)) return wino_rod_launch(q, wino_mbw(q.cout_g), stream);
  }
  switch (wino_mbw(q.cout_g)) {
    case 4: return launch_variant<4, 8>(q, stream);
    case 2: return launch_variant<2, 8>(q, stream);
    default: return launch_variant<1, 8>(q, stream);
  }
}

}  // namespace vspconv

// Winograd F(2x2, 3x3), row-owner form (conv_wino_ro.hip) for the DILATION GROUPS of the SMART layers: up to four groups over one shared
// input, dilation = padding = 1, 2, 4 or 8 per group (reference models/RestoreNet.py:205-215, 604-668).
//
// Geometry of conv_wino.hip's dilated variant: row-polyphase -- a workgroup owns the rows ry, ry + d, ... of one residue class
// (consecutive patch rows) and a DENSE run of 16 columns with a halo of d; its 8 tile columns are the d column residues x 8 / d tile
// positions (tile tx: residue tx % d, first column residue + 2 d (tx / d)), the window columns lie d apart.  What changes against the
// task-list kernel is what conv_wino_ro.hip changes: wave w owns the positions (xi, nu) = (w / 2, 2 (w % 2) + {0, 1}) and computes its own B
// fragments from the staged patch (two window rows x four words d apart per N-block: ds_read_b32), no V image, no transform tasks;
// patch rows staged as aligned 16-byte segments (columns ox0 - max(d, 4) ...); scale / affine at commit time.
// LDS banks (4-byte reads: bank = dword mod 32, 32-lane groups = two channels x (8 tile columns x 2 tile rows)): the tile columns of a
// group occupy dwords c0(tx) in [0, 16) -- every second one for d = 1, pairs for d = 2, ... -- so the row pitch puts the second tile row
// 16 banks away (pitch == 8 mod 16: 24, or 40 for the 32-column rows of d = 8) and the channel pitch shifts the second channel by d
// (pitch == d mod 32): the four 8-dword sets of an access tile the 32 banks.
#include "conv_kernel.h"

namespace vspconv {

namespace {

__device__ __forceinline__ float uload_rod(const float* base, int idx) {  // wave-uniform operand through the scalar cache
  typedef const float __attribute__((address_space(4))) * cfp4;
  return ((cfp4)(uintptr_t)base)[__builtin_amdgcn_readfirstlane(idx)];
}

constexpr int RD_NTHR = 512;
constexpr int RD_IVC = 8;

template <int MBW>
struct RDG {
  static constexpr int NBW = 8 / MBW;
  static constexpr int WCO = 16 * MBW;
  static constexpr int NTILE = 16 * NBW;
  static constexpr int TLX = 8;
  static constexpr int TLY = NTILE / TLX;
  static constexpr int PR = 2 * TLY + 2;
  static constexpr int NLD = (PR * 8 + 63) / 64;                       // wave loads per channel plane at the widest rows (d = 8: 8 segments)
  static constexpr int PPMAX = (PR * 40 + 1 + 31) / 32 * 32 + 8;      // largest channel pitch (d = 8)
  static constexpr int LDS_P = RD_IVC * PPMAX;                        // floats per sub-stage buffer
  static constexpr int ETILE = NTILE > 64 ? 64 : NTILE;
  static constexpr int EMB = (MBW >= 2 && ETILE <= 32) ? 2 : 1;
  static constexpr int EP = ETILE + 4;
  static constexpr int LDS_M = 16 * 16 * EMB * EP;
  static constexpr int LDS_FLOATS = 2 * LDS_P > LDS_M ? 2 * LDS_P : LDS_M;
  static constexpr int UF = 2 * MBW;
};

template <int MBW>
__global__ __launch_bounds__(RD_NTHR, 4) void conv_wino_rod_kernel(const ConvK p) {
  using Gm = RDG<MBW>;
  constexpr int NBW = Gm::NBW, WCO = Gm::WCO, NTILE = Gm::NTILE, TLX = Gm::TLX, TLY = Gm::TLY, PR = Gm::PR;
  constexpr int LDS_P = Gm::LDS_P, UF = Gm::UF, NLD = Gm::NLD, IVC = RD_IVC, KS = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Pl = smem;   // 2 x [IVC][channel pitch of this group]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  // ---- work order (conv_wino.hip): region-major for shared-input dilation groups (order 4), pixel-tile-major (1) or dispatch order
  int b = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  int reg_ry = -1, reg_ty = 0, reg_tx = 0;
  if (p.wg_order == 4) {
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
    const int GN = GX * GY;
    b = lid / GN;
    const int lrem = lid - b * GN;
    bx = lrem / GY;
    by = lrem - bx * GY;
  }
  const int g = by / p.co_tiles, ct = by - g * p.co_tiles;
  const int d = p.dil[g];                                       // 1, 2, 4 or 8
  const int SH = (p.H + d - 1) / d;                             // rows of one residue class
  const int tiles_x = (p.W + 2 * TLX - 1) / (2 * TLX), tiles_y = (SH + 2 * TLY - 1) / (2 * TLY);
  const int per_res = tiles_x * tiles_y;
  int ry, tx_i, ty_i;
  if (reg_ry >= 0) {
    if (reg_ty >= tiles_y || reg_tx >= tiles_x) return;
    ry = reg_ry; ty_i = reg_ty; tx_i = reg_tx;
  } else {
    if (bx >= per_res * d) return;
    ry = bx / per_res;
    const int tile_i = bx - ry * per_res;
    tx_i = tile_i % tiles_x;
    ty_i = tile_i / tiles_x;
  }
  const int oy0 = ty_i * (2 * TLY), ox0 = tx_i * (2 * TLX);     // sub-image rows, image columns
  const int hl = d > 4 ? d : 4;                                 // staged halo: whole segments
  const int SEG = (2 * TLX + 2 * hl) / 4;                       // 6 (d <= 4) or 8 segments per row
  const int PCP = d == 8 ? 40 : 24;                             // row pitch (== 8 mod 16)
  const int PPITCH = (PR * PCP + 1 + 31) / 32 * 32 + d;         // channel pitch (== d mod 32)
  const int co0 = ct * WCO;
  const int chw = p.H * p.W;
  const float* xb = p.x + (int64_t)b * p.x_ch * chw;
  const int nstage = (p.Cin + IVC - 1) / IVC;
  const int nchunk4 = (p.Cin + 3) / 4;

  // ---- patch staging: one channel plane per wave and sub-stage; a lane owns segment (row, seg) = (l / SEG, l % SEG), l = lane + 64 i
  //      (lanes past the plane repeat segment 0: same address, same value)
  int p_voff[NLD], p_dst[NLD];
  unsigned p_ok = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int l = lane + 64 * i;
    const bool live = l < PR * SEG;
    const int r = live ? l / SEG : 0, sg = live ? l - r * SEG : 0;
    const int sy = oy0 - 1 + r, ix = ox0 - hl + 4 * sg;
    const int iy = sy * d + ry;
    const bool in = sy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    p_voff[i] = in ? (iy * p.W + ix) * 4 : 0x7ffffff0;   // padding: a lane offset past the resource, the load returns zeros
    p_ok |= in ? (1u << i) : 0u;
    p_dst[i] = 1 + r * PCP + 4 * sg;
  }
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.x_ch * chw * 4, 0x00020000);   // this image (the range check compares the lane offset with size - scalar offset)
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v preg[NLD];
  auto load_plane = [&](int j) {
    const int jj = j < nstage ? j : nstage - 1;
    const int ci = jj * IVC + wave;
    const bool chin = ci < p.Cin;                                  // (a channel past the layer: every lane offset out of range -> zeros)
    const int soff = (chin ? ci : 0) * chw * 4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) preg[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, chin ? p_voff[i] : 0x7ffffff0, soff, 0));
  };
  const float* wt_b = p.wtp + b * p.wt_bs;
  const float* wc_b = p.wcp + b * p.wc_bs;
  const bool affine = p.wc_cs != 0 || p.wsh_cs != 0 || p.wc_bs != 0;
  auto commit_plane = [&](float* Pdst, int j) {
    const int ci = j * IVC + wave;
    const bool chok = ci < p.Cin;
    const int cc = chok ? ci : p.Cin - 1;
    const float st = uload_rod(wt_b, cc * p.wt_cs);
    float sc = st, sh = 0.f;
    if (affine) {   // (uniform for the launch) folded-BatchNorm input of the IR-SE body; the modulated layers skip two scalar loads and their address arithmetic
      sc = uload_rod(wc_b, cc * p.wc_cs) * st;
      sh = uload_rod(p.wshp, cc * p.wsh_cs) * st;
    }
    float* dst = Pdst + wave * PPITCH;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float shm = (((p_ok >> i) & 1u) && chok) ? sh : 0.f;   // (padding / absent channels arrive as zeros: conv_wino_ro.hip)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[p_dst[i] + e] = fmaf(preg[i][e], sc, shm);
    }
  };

  // ---- U fragments: [group][co tile][chunk][wave][pp 2][lane][mb MBW]; buffer loads (resource = this channel tile's slice, scalar
  //      offset = chunk, fixed lane offset: conv_wino_ro.hip)
  const float* utile = p.w + ((int64_t)g * p.co_tiles + ct) * nchunk4 * (8 * 64 * UF);
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(utile), 0, nchunk4 * (8 * 64 * UF) * 4, 0x00020000);
  const int u_voff = ((wave * 2 * 64 + lane) * MBW) * 4;
  auto load_u = [&](int c, float (&u)[UF]) {
    const int cc = c < nchunk4 ? c : nchunk4 - 1;
    const int soff = cc * (8 * 64 * UF * 4);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      if constexpr (MBW == 4) {
        typedef float f32x4b __attribute__((ext_vector_type(4)));
        const f32x4b a = __builtin_bit_cast(f32x4b, __builtin_amdgcn_raw_buffer_load_b128(ursrc, u_voff + pp * 64 * MBW * 4, soff, 0));
        u[pp * 4 + 0] = a[0]; u[pp * 4 + 1] = a[1]; u[pp * 4 + 2] = a[2]; u[pp * 4 + 3] = a[3];
      } else {
        typedef float f32x2b __attribute__((ext_vector_type(2)));
        const f32x2b a = __builtin_bit_cast(f32x2b, __builtin_amdgcn_raw_buffer_load_b64(ursrc, u_voff + pp * 64 * MBW * 4, soff, 0));
        u[pp * 2 + 0] = a[0]; u[pp * 2 + 1] = a[1];
      }
    }
  };

  // ---- this wave's row of the transformed tile
  const int xi = wave >> 1, nuh = wave & 1;
  const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int rB = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sgn = xi == 1 ? 1.f : -1.f;
  // lane's window origin: tile = lr + 16 nb -> (ty, tx) = (2 nb + lr / 8, lr % 8); first window column (patch coordinates) hl - d + c0(tx)
  const int wty = lr >> 3, wtx = lr & 7;
  const int c0 = (wtx % d) + 2 * d * (wtx / d);
  const int nbstep = 4 * PCP;                                   // the next N-block lies two tile rows down
  const int woffA = kq * PPITCH + (2 * wty + rA) * PCP + 1 + (hl - d) + c0;
  const int woffB = kq * PPITCH + (2 * wty + rB) * PCP + 1 + (hl - d) + c0;
  auto fragments = [&](const float* Psrc, int ks, float (&bv)[2][NBW]) {
    const float* base = Psrc + ks * (4 * PPITCH);
#pragma unroll
    for (int n0 = 0; n0 < NBW; n0 += 2) {
      float wa[2][4], wb[2][4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          wa[n][k] = base[woffA + (n0 + n) * nbstep + k * d];
          wb[n][k] = base[woffB + (n0 + n) * nbstep + k * d];
        }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const float w0 = fmaf(wb[n][0], sgn, wa[n][0]), w1 = fmaf(wb[n][1], sgn, wa[n][1]);
        const float w2 = fmaf(wb[n][2], sgn, wa[n][2]), w3 = fmaf(wb[n][3], sgn, wa[n][3]);
        bv[0][n0 + n] = nuh ? w2 - w1 : w0 - w2;
        bv[1][n0 + n] = nuh ? w1 - w3 : w1 + w2;
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
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
        acc[pp][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[pp * MBW + mb], bv[pp][nb], acc[pp][mb][nb], 0, 0, 0);
  };

  // ---- pipeline (conv_wino_ro.hip, barrier period 1)
  float ua[UF], ub[UF];
  float bva[2][NBW], bvb[2][NBW];
  load_u(0, ua);
  load_plane(0);
  commit_plane(Pl, 0);
  load_plane(1);
  __syncthreads();
  constexpr int SB = 0x2 | 0x4 | 0x80 | 0x100 | 0x200;   // VALU, SALU, LDS may cross; MFMAs and vector-memory instructions may not
  for (int j = 0; j < nstage; ++j) {
    const float* Pcur = Pl + (j & 1) * LDS_P;
    float* Pnxt = Pl + ((j + 1) & 1) * LDS_P;
    fragments(Pcur, 0, bva);
    load_u(j * KS + 1, ub);
    __builtin_amdgcn_sched_barrier(SB);
    multiply_pp(0, ua, bva);
    __builtin_amdgcn_sched_barrier(SB);
    commit_plane(Pnxt, j + 1);
    load_plane(j + 2);
    __builtin_amdgcn_sched_barrier(SB);
    fragments(Pcur, 1, bvb);
    multiply_pp(1, ua, bva);
    __builtin_amdgcn_sched_barrier(SB);
    load_u(j * KS + 2, ua);
    __builtin_amdgcn_sched_barrier(SB);
    multiply_pp(0, ub, bvb);
    __builtin_amdgcn_sched_barrier(SB);
    multiply_pp(1, ub, bvb);
    __builtin_amdgcn_sched_barrier(SB);
    __syncthreads();
  }

  // ---- epilogue (conv_wino.hip): all sixteen positions through LDS, one thread per (channel, tile); pixels of a tile lie d apart
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
  constexpr int EPT = ECO * ETILE / RD_NTHR;
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  const bool pairs = d == 1 && p.r1s <= 1 && p.r2s <= 1 && (p.OW & 1) == 0 && p.OW >= 2;
#pragma unroll
  for (int mb0 = 0; mb0 < MBW; mb0 += EMB) {
#pragma unroll
    for (int th = 0; th < NTILE / ETILE; ++th) {
      if (mb0 + th > 0) __syncthreads();
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
        const int pair = tid + it * RD_NTHR;
        const int e_co = pair / ETILE, e_t = pair - e_co * ETILE;
        const int e_tile = th * ETILE + e_t;
        const int e_tx = e_tile % TLX;
        const int sy = oy0 + 2 * (e_tile / TLX);
        const int sx = ox0 + (e_tx % d) + 2 * d * (e_tx / d);   // first output column of the tile
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
            for (int jx = 0; jx < 2; ++jx) {
              const int oxj = ox + jx * d;
              if (oxj >= p.OW) continue;
              const int rj = ro + jx * d;
              yb[rj] = fin(yv[i][jx], nzp[(oy * p.OW + oxj) * p.nzs], r1b[rj * p.r1s], r2b[rj * p.r2s]);
            }
          }
        }
      }
    }
  }
}

template <int MBW>
int launch_rod(ConvK q, hipStream_t stream) {
  using Gm = RDG<MBW>;
  static vsp::LdsAttrOnce attr;
  const size_t lds = (size_t)Gm::LDS_FLOATS * sizeof(float);
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wino_rod_kernel<MBW>), (int)lds, "conv2d_winograd (row-owner, dilation groups)")) return rc;
  q.co_tiles = (q.cout_g + Gm::WCO - 1) / Gm::WCO;
  int blocks = 0;  // the largest per-group tile count (groups with a smaller dilation exit early)
  for (int g = 0; g < q.G; ++g) {
    const int d = q.dil[g];
    const int SH = (q.H + d - 1) / d;
    const int n = ((q.W + 2 * Gm::TLX - 1) / (2 * Gm::TLX)) * ((SH + 2 * Gm::TLY - 1) / (2 * Gm::TLY)) * d;
    blocks = n > blocks ? n : blocks;
  }
  q.wg_order = q.H * q.W <= 1024 ? 1 : 0;
  if (q.G >= 2 && q.x_gs == 0 && !(q.dbg & 0x800000)) {   // region-major order over bands of 8 x 2 TLY rows x 4 column tiles (conv_wino.hip)
    q.wg_order = 4;
    q.tiles_y = (q.H + 16 * Gm::TLY - 1) / (16 * Gm::TLY);
    q.tiles_x = ((q.W + 2 * Gm::TLX - 1) / (2 * Gm::TLX) + 3) / 4;
    blocks = q.tiles_y * q.tiles_x * 32;
  }
  dim3 grid((unsigned)blocks, (unsigned)(q.co_tiles * q.G), (unsigned)q.B);
  conv_wino_rod_kernel<MBW><<<grid, RD_NTHR, lds, stream>>>(q);
  return VSP_OK;
}

}  // namespace

// dilation-group launches the row-owner form serves: dilations 1, 2, 4, 8, rows of whole 16-byte segments, 32 or more channels per group
bool wino_rod_eligible(const ConvK& q) {
  for (int g = 0; g < q.G; ++g)
    if (q.dil[g] != 1 && q.dil[g] != 2 && q.dil[g] != 4 && q.dil[g] != 8) return false;
  // padding = the raw-buffer range check of ONE image's descriptor (num_records = x_ch * H * W * 4 as an int; out-of-range lanes carry
  // offset 0x7ffffff0): an image of 2 GiB or more would wrap the record count and leave the padding unbacked -- refused here
  if ((int64_t)q.x_ch * q.H * q.W * 4 >= 0x7ffffff0ll) return false;
  return q.cout_g > 16 && q.W % 4 == 0 && (reinterpret_cast<uintptr_t>(q.x) & 15) == 0 && ((int64_t)q.H * q.W) % 4 == 0;
}

int wino_rod_launch(ConvK q, int mbw, hipStream_t stream) { return mbw == 4 ? launch_rod<4>(q, stream) : launch_rod<2>(q, stream); }

}  // namespace vspconv

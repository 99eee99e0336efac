// Winograd F(2x2, 3x3) convolution on fp32 MFMA for gfx950: the stride-1, dilation-1, 3x3 layers of the path
// (StyledConv / SMART fusion / IR-SE convolutions) with 16 multiplies per 2x2 output tile instead of 36.
//
// The direct kernel (conv_kernel.h) is bound by the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 issues at 1/4 of the bf16
// rate): its big layers sit at 118 TFLOP/s of a 131 TFLOP/s MFMA-only ceiling.  The only way past that at fp32 is fewer
// multiplies.  With V = B^T d B (input tile 4x4), U = G g G^T (weights, precomputed once per layer), M = sum_ci U . V per
// Winograd position and Y = A^T M A, a 2x2 output tile costs 16 MACs per (co, ci) -- sixteen independent (Cout x Cin) x
// (Cin x tiles) GEMMs on MFMA -- plus transforms that are additions only.
//
// One workgroup (8 waves) owns 64 output channels x a 16 x 8 pixel region (8 x 4 tiles) of one image; wave w owns the
// Winograd positions 2w and 2w+1.  What shapes the kernel (ablations on the first version: the global loads were 31 % of
// the time, LDS writes 5 %, the transform 5 %, and nothing of it overlapped the MFMAs because three barriers per chunk kept
// the eight waves in lock step):
//   * U never touches LDS.  No two waves share a position, so the host stores U in FRAGMENT order
//     [co tile][chunk][wave][lane][16] and a wave fetches its A fragments of a chunk as four 16-byte loads per lane
//     (4 KB contiguous per wave), one chunk ahead, straight into the registers the MFMAs read.
//   * the input patch is prefetched two chunks ahead (it streams from MALL/HBM) and double-buffered in LDS, V is
//     double-buffered too: the transform of chunk i+1, the MFMAs of chunk i and the patch write of chunk i+2 share ONE
//     barrier interval.
// Epilogue, per 16-channel block: the sixteen position accumulators meet in LDS, one thread per (channel, tile) applies
// A^T . A, the same fused operand chain as the direct kernel (demod, bias, two activations, noise, two residuals) and stores
// the 2x2 pixels.  Numerics: F(2x2,3x3) in fp32 adds ~1e-6 relative error (transform constants are 1 and 1/2).
#include "conv_kernel.h"

namespace vspconv {

namespace {

constexpr int WCK = 4;      // input channels per chunk (the host packs U for this value: vsp_conv2d_winograd_chunk())
constexpr int KS = WCK / 4; // k-steps per chunk
constexpr int WCO = 64;     // output channels per workgroup
constexpr int TLX = 8, TLY = 4, NTILE = TLX * TLY;  // Winograd tiles per workgroup (16 x 8 pixels)
constexpr int PR = 2 * TLY + 2, PC = 2 * TLX + 2;   // input patch 10 x 18
constexpr int PPITCH = 192;                         // >= PR * PC
constexpr int VPITCH = NTILE + 16;                  // 48: k-slot rows 16 banks apart
constexpr int NTHR = 512;
constexpr int LDS_V = 16 * WCK * VPITCH;            // 6144 floats, two buffers
constexpr int LDS_P = WCK * PPITCH;                 // 1536 floats, two buffers
constexpr int LDS_M = 16 * 16 * NTILE;              // 8192 (epilogue, overlays everything)
constexpr int LDS_STAGE = 2 * LDS_V + 2 * LDS_P;
constexpr int LDS_FLOATS = LDS_STAGE > LDS_M ? LDS_STAGE : LDS_M;

__global__ __launch_bounds__(NTHR, 4) void conv_wino_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vl = smem;               // 2 x [16][WCK][VPITCH]
  float* Pl = smem + 2 * LDS_V;   // 2 x [WCK][PPITCH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int tile_i = blockIdx.x;
  const int tx_i = tile_i % p.tiles_x, ty_i = tile_i / p.tiles_x;
  const int co0 = blockIdx.y * WCO;
  const int b = blockIdx.z;
  const int oy0 = ty_i * (2 * TLY), ox0 = tx_i * (2 * TLX);
  const int Cout = p.cout_g;
  const int chw = p.H * p.W;
  const float* xb = p.x + (int64_t)b * p.x_ch * chw;
  const int nchunk = (p.Cin + WCK - 1) / WCK;

  // ---- input patch: chunk-invariant geometry, values prefetched two chunks ahead
  constexpr int PWORDS = WCK * PR * PC;             // 1440 patch words per chunk
  constexpr int PLD = (PWORDS + NTHR - 1) / NTHR;   // 3
  int p_src[PLD], p_dst[PLD], p_ch[PLD];            // image offset (-1: outside / unused), LDS word, channel in chunk
#pragma unroll
  for (int e = 0; e < PLD; ++e) {
    const int i = tid + e * NTHR;
    const int ch = i / (PR * PC), rem = i - ch * (PR * PC);
    const int r = rem / PC, c = rem - r * PC;
    const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
    const bool ok = i < PWORDS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    p_src[e] = ok ? iy * p.W + ix : -1;
    p_dst[e] = i < PWORDS ? ch * PPITCH + rem : -1;
    p_ch[e] = ch;
  }
  float preg[PLD], pnext[PLD];
  auto issue_p = [&](int c) {  // chunk c -> pnext
#pragma unroll
    for (int e = 0; e < PLD; ++e) {
      const int ci = c * WCK + p_ch[e];
      const bool ok = p_src[e] >= 0 && ci < p.Cin;
      float v = 0.f;
      if (ok) {
        v = xb[(int64_t)ci * chw + p_src[e]];
        if (p.in_shift) {  // affine input (folded BatchNorm): the shift belongs to in-image pixels only, so it is applied here
          const float sc = p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_bstride + ci] : 1.f;
          v = fmaf(v, sc, p.in_shift[ci]);
        }
      }
      pnext[e] = v;
    }
  };
  auto commit_p = [&](float* Pdst) {
#pragma unroll
    for (int e = 0; e < PLD; ++e)
      if (p_dst[e] >= 0) Pdst[p_dst[e]] = preg[e];
  };

  // ---- U fragments: [co tile][chunk][wave][lane][pp 2][ks KS][mb 4] floats, 32 KS bytes per lane and chunk
  constexpr int UQ = 2 * KS;  // float4 per lane and chunk
  const float* ufr = p.w + (((int64_t)blockIdx.y * nchunk * 8 + wave) * 64 + lane) * (4 * UQ);
  auto load_u = [&](int c, float4 (&u)[UQ]) {
    const float4* src = reinterpret_cast<const float4*>(ufr + (int64_t)c * (8 * 64 * 4 * UQ));
#pragma unroll
    for (int q = 0; q < UQ; ++q) u[q] = src[q];
  };

  // ---- transform roles: thread pair q = tid >> 1 owns (channel, tile); half h = tid & 1 produces V rows 2h, 2h+1
  const int tq = tid >> 1, th = tid & 1;
  const int t_ch = tq >> 5, t_tile = tq & 31;
  const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
  const int t_src = t_ch * PPITCH + (2 * t_ty) * PC + 2 * t_tx;
  const int t_dst = t_ch * VPITCH + t_tile;
  auto transform = [&](const float* Psrc, float* Vdst, int c) {  // chunk c: V = B^T d B, the style scale rides on V
    if (tid >= 2 * WCK * NTILE) return;
    float d[3][4];  // rows h, h+1, h+2 of the 4x4 window
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) d[r][cc] = Psrc[t_src + (th + r) * PC + cc];
    float sc = 1.f;
    if (p.in_scale && !p.in_shift && c * WCK + t_ch < p.Cin) sc = p.in_scale[(int64_t)b * p.in_scale_bstride + c * WCK + t_ch];
    float w0[4], w1[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      // h = 0: W0 = d0 - d2, W1 = d1 + d2      h = 1: W2 = d2 - d1, W3 = d1 - d3  (d[0..2] = window rows h..h+2)
      w0[cc] = (th ? d[1][cc] - d[0][cc] : d[0][cc] - d[2][cc]) * sc;
      w1[cc] = (th ? d[0][cc] - d[2][cc] : d[1][cc] + d[2][cc]) * sc;
    }
    const float v0[4] = {w0[0] - w0[2], w0[1] + w0[2], w0[2] - w0[1], w0[1] - w0[3]};
    const float v1[4] = {w1[0] - w1[2], w1[1] + w1[2], w1[2] - w1[1], w1[1] - w1[3]};
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      Vdst[t_dst + ((2 * th) * 4 + nu) * (WCK * VPITCH)] = v0[nu];
      Vdst[t_dst + ((2 * th + 1) * 4 + nu) * (WCK * VPITCH)] = v1[nu];
    }
  };

  f32x4 acc[2][4][2];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[pp][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const float* Vsrc, const float4 (&u)[UQ]) {
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const float* vp = Vsrc + (2 * wave + pp) * (WCK * VPITCH) + kq * VPITCH + lr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 a = u[pp * KS + ks];
        const float bv0 = vp[ks * 4 * VPITCH], bv1 = vp[ks * 4 * VPITCH + 16];
        acc[pp][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv0, acc[pp][0][0], 0, 0, 0);
        acc[pp][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv1, acc[pp][0][1], 0, 0, 0);
        acc[pp][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv0, acc[pp][1][0], 0, 0, 0);
        acc[pp][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv1, acc[pp][1][1], 0, 0, 0);
        acc[pp][2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv0, acc[pp][2][0], 0, 0, 0);
        acc[pp][2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv1, acc[pp][2][1], 0, 0, 0);
        acc[pp][3][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv0, acc[pp][3][0], 0, 0, 0);
        acc[pp][3][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv1, acc[pp][3][1], 0, 0, 0);
      }
    }
  };

  // ---- pipeline.  State at the top of step i:  Vl[i&1] = V(i), Pl[(i+1)&1] = patch(i+1), ua = U(i) (registers, landed),
  //      preg = patch(i+2) (landed), in flight: nothing older than one step.
  float4 ua[UQ], ub[UQ];
  issue_p(0);
  load_u(0, ua);
#pragma unroll
  for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];
  commit_p(Pl);                                   // patch(0) -> Pl[0]
  if (nchunk > 1) issue_p(1);
  __syncthreads();
  transform(Pl, Vl, 0);                           // V(0) -> Vl[0]
  if (nchunk > 1) {
#pragma unroll
    for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];
    commit_p(Pl + LDS_P);                         // patch(1) -> Pl[1]
    if (nchunk > 2) issue_p(2);
  }
  __syncthreads();
  auto step = [&](int i, const float4 (&ucur)[UQ], float4 (&unxt)[UQ]) {
    const int cur = i & 1, nxt = cur ^ 1;
    if (i + 1 < nchunk) load_u(i + 1, unxt);
    if (i + 2 < nchunk) {
#pragma unroll
      for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];   // patch(i+2), issued one step ago
    }
    if (i + 3 < nchunk) issue_p(i + 3);
    if (i + 1 < nchunk) transform(Pl + nxt * LDS_P, Vl + nxt * LDS_V, i + 1);
    multiply(Vl + cur * LDS_V, ucur);
    if (i + 2 < nchunk) commit_p(Pl + cur * LDS_P);        // Pl[cur] held patch(i): consumed one step ago
    __syncthreads();
  };
  for (int i = 0; i < nchunk; i += 2) {
    step(i, ua, ub);
    if (i + 1 < nchunk) step(i + 1, ub, ua);
  }

  // ---- epilogue: per 16-channel block, all sixteen positions through LDS, one thread per (channel, tile)
  float* Ml = smem;  // [16 pos][16 co][NTILE]
  const float* osp = p.osp + (int64_t)b * Cout * p.oss;
  const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW * p.nzs;
  const float nw = p.nwp[0];
  float* yb = p.y + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const float* r1b = p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r1s;
  const float* r2b = p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r2s;
  const int y_plane = p.y_h * p.y_w;
  const int e_co = tid >> 5, e_tile = tid & 31;
  const int e_oy = oy0 + 2 * (e_tile >> 3), e_ox = ox0 + 2 * (e_tile & 7);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Ml[((2 * wave + pp) * 16 + kq * 4 + r) * NTILE + nb * 16 + lr] = acc[pp][mb][nb][r];
    __syncthreads();
    float m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = Ml[(q * 16 + e_co) * NTILE + e_tile];
    float t0[4], t1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      t0[nu] = m[nu] + m[4 + nu] + m[8 + nu];
      t1[nu] = m[4 + nu] - m[8 + nu] - m[12 + nu];
    }
    const float yv[2][2] = {{t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3]}, {t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]}};
    const int cg = co0 + mb * 16 + e_co;
    if (cg >= Cout) continue;
    const float os = osp[cg * p.oss], cs = p.csp[cg * p.css], cb = p.cbp[cg * p.cbs];
    const float b1 = p.b1p[cg * p.b1s], b2 = p.b2p[cg * p.b2s], sl2 = p.s2p[cg * p.s2s];
    const int cbase = cg * y_plane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int oy = e_oy + i, ox = e_ox + j;
        if (oy >= p.OH || ox >= p.OW) continue;
        const int ro = cbase + oy * p.y_w + ox;
        float v = yv[i][j] * os;
        v = v * cs + cb;
        v += b1;
        v = (v > 0.f ? v : v * p.s1) * p.g1;
        v += nzp[(oy * p.OW + ox) * p.nzs] * nw;
        v += b2;
        v = (v > 0.f ? v : v * sl2) * p.g2;
        v += r1b[ro * p.r1s];
        v += r2b[ro * p.r2s];
        yb[ro] = v;
      }
  }
}

}  // namespace

int wino_chunk() { return WCK; }

int wino_launch(ConvK q, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d_winograd: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  q.tiles_x = (q.OW + 2 * TLX - 1) / (2 * TLX);
  q.tiles_y = (q.OH + 2 * TLY - 1) / (2 * TLY);
  dim3 grid((unsigned)(q.tiles_x * q.tiles_y), (unsigned)((q.cout_g + WCO - 1) / WCO), (unsigned)q.B);
  conv_wino_kernel<<<grid, NTHR, lds, stream>>>(q);
  return VSP_OK;
}

}  // namespace vspconv

// Winograd F(2x2, 3x3) convolution on fp32 MFMA for gfx950: the stride-1, dilation-1, 3x3 layers of the path
// (StyledConv / SMART fusion / IR-SE convolutions) with 16 multiplies per 2x2 output tile instead of 36.
//
// The direct kernel (conv_kernel.h) is bound by the fp32 matrix pipe (v_mfma_f32_16x16x4_f32 issues at 1/4 of the bf16
// rate): its big layers sit at 118 TFLOP/s of a 131 TFLOP/s MFMA-only ceiling.  The only way past that at fp32 is fewer
// multiplies.  With V = B^T d B (input tile 4x4), U = G g G^T (weights, precomputed once per layer), M = sum_ci U . V per
// Winograd position and Y = A^T M A, a 2x2 output tile costs 16 MACs per (co, ci) -- sixteen independent (Cout x Cin) x
// (Cin x tiles) GEMMs on MFMA -- plus transforms that are additions only.
//
// One workgroup (8 waves) owns 64 output channels x a 16 x 8 pixel region (8 x 4 tiles) of one image.  Per chunk of 8
// input channels:
//   1. the prefetched 18 x 10 input patch and the 16 x 8 x 64 slab of U go from registers to LDS        (barrier)
//   2. every thread pair turns one (channel, tile) window into its 16 V values (LDS -> LDS, adds only),
//      while the global loads of the next chunk are already in flight                                    (barrier)
//   3. wave w multiplies positions 2w, 2w+1: 2 k-steps x (4 co blocks x 2 tile blocks) MFMAs each        (barrier)
// Epilogue, per 16-channel block: the sixteen position accumulators meet in LDS, one thread per (channel, tile) applies
// A^T . A, the same fused operand chain as the direct kernel (demod, bias, two activations, noise, two residuals) and stores
// the 2x2 pixels.  Numerics: F(2x2,3x3) in fp32 adds ~1e-6 relative error (transform constants are 1 and 1/2).
#include "conv_kernel.h"

namespace vspconv {

namespace {

constexpr int WCK = 8;      // input channels per chunk
constexpr int WCO = 64;     // output channels per workgroup
constexpr int TLX = 8, TLY = 4, NTILE = TLX * TLY;  // Winograd tiles per workgroup (16 x 8 pixels)
constexpr int PR = 2 * TLY + 2, PC = 2 * TLX + 2;   // input patch 10 x 18
constexpr int PPITCH = 192;                         // >= PR * PC
constexpr int UPITCH = WCO + 16;                    // 80: k-slot rows 16 banks apart
constexpr int VPITCH = NTILE + 16;                  // 48
constexpr int NTHR = 512;
constexpr int LDS_U = 16 * WCK * UPITCH;            // 10240 floats
constexpr int LDS_V = 16 * WCK * VPITCH;            // 6144
constexpr int LDS_P = WCK * PPITCH;                 // 1536
constexpr int LDS_M = 16 * 16 * NTILE;              // 8192 (epilogue, overlays U)
constexpr int LDS_FLOATS = LDS_U + LDS_V + LDS_P;

__global__ __launch_bounds__(NTHR, 4) void conv_wino_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ul = smem;                 // [16][WCK][UPITCH]
  float* Vl = smem + LDS_U;         // [16][WCK][VPITCH]
  float* Pl = smem + LDS_U + LDS_V; // [WCK][PPITCH]

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

  // ---- chunk-invariant staging geometry (one pass per thread)
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
  constexpr int ULD = 16 * WCK * (WCO / 4) / NTHR;  // 4 float4 of U per thread and chunk
  int u_src[ULD], u_dst[ULD], u_ci[ULD];
#pragma unroll
  for (int e = 0; e < ULD; ++e) {
    const int j = tid + e * NTHR;
    const int pos = j / (WCK * (WCO / 4)), rem = j - pos * (WCK * (WCO / 4));
    const int ci = rem / (WCO / 4), c4 = rem - ci * (WCO / 4);
    u_ci[e] = ci;
    u_dst[e] = (pos * WCK + ci) * UPITCH + c4 * 4;
    u_src[e] = (co0 + c4 * 4 < Cout) ? (pos * p.Cin + ci) * Cout + co0 + c4 * 4 : -1;
  }
  // prefetch registers: U one chunk ahead (L2-resident, 16 KB per layer and position), the input patch TWO chunks ahead
  // (it streams from MALL/HBM: its latency is longer than one chunk of work)
  float preg[PLD], pnext[PLD];
  float4 ureg[ULD];
  auto issue_u = [&](int ci0) {
#pragma unroll
    for (int e = 0; e < ULD; ++e) {
      const bool ok = u_src[e] >= 0 && ci0 + u_ci[e] < p.Cin;
      const float4 v = *reinterpret_cast<const float4*>(p.w + (int64_t)ci0 * Cout + (ok ? u_src[e] : 0));
      ureg[e] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto issue_p = [&](int ci0) {
#pragma unroll
    for (int e = 0; e < PLD; ++e) {
      const int ci = ci0 + p_ch[e];
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

  // ---- transform roles: thread pair q = tid >> 1 owns (channel, tile); half h = tid & 1 produces V rows 2h, 2h+1
  const int tq = tid >> 1, th = tid & 1;
  const int t_ch = tq >> 5, t_tile = tq & 31;
  const int t_ty = t_tile >> 3, t_tx = t_tile & 7;
  const float* t_src = Pl + t_ch * PPITCH + (2 * t_ty) * PC + 2 * t_tx;
  float* t_dst = Vl + t_ch * VPITCH + t_tile;

  f32x4 acc[2][4][2];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[pp][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_u(0);
  issue_p(0);
#pragma unroll
  for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];
  if (WCK < p.Cin) issue_p(WCK);
  for (int ci0 = 0; ci0 < p.Cin; ci0 += WCK) {
    __syncthreads();  // previous MFMA phase has finished with Ul / Vl
#pragma unroll
    for (int e = 0; e < ULD; ++e) *reinterpret_cast<float4*>(Ul + u_dst[e]) = ureg[e];
#pragma unroll
    for (int e = 0; e < PLD; ++e)
      if (p_dst[e] >= 0) Pl[p_dst[e]] = preg[e];
    __syncthreads();
    if (ci0 + WCK < p.Cin) issue_u(ci0 + WCK);
#pragma unroll
    for (int e = 0; e < PLD; ++e) preg[e] = pnext[e];  // patch of chunk ci0 + WCK (issued one iteration ago)
    if (ci0 + 2 * WCK < p.Cin) issue_p(ci0 + 2 * WCK);
    {  // V = B^T d B for this thread's (channel, tile): rows 2h, 2h+1 of W = B^T d, then W B
      float d[3][4];  // rows h, h+1, h+2 of the 4x4 window
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r][c] = t_src[(th + r) * PC + c];
      float sc = 1.f;  // the style scale rides on V (the transform is linear)
      if (p.in_scale && !p.in_shift && ci0 + t_ch < p.Cin) sc = p.in_scale[(int64_t)b * p.in_scale_bstride + ci0 + t_ch];
      float w0[4], w1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // h = 0: W0 = d0 - d2, W1 = d1 + d2      h = 1: W2 = d2 - d1, W3 = d1 - d3  (rows relative to h: d[0..2] = d_h..d_{h+2})
        w0[c] = (th ? d[1][c] - d[0][c] : d[0][c] - d[2][c]) * sc;
        w1[c] = (th ? d[0][c] - d[2][c] : d[1][c] + d[2][c]) * sc;
      }
      const float v0[4] = {w0[0] - w0[2], w0[1] + w0[2], w0[2] - w0[1], w0[1] - w0[3]};
      const float v1[4] = {w1[0] - w1[2], w1[1] + w1[2], w1[2] - w1[1], w1[1] - w1[3]};
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        t_dst[((2 * th) * 4 + nu) * (WCK * VPITCH)] = v0[nu];
        t_dst[((2 * th + 1) * 4 + nu) * (WCK * VPITCH)] = v1[nu];
      }
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int pos = 2 * wave + pp;
      const float* up = Ul + pos * (WCK * UPITCH) + kq * UPITCH + lr;
      const float* vp = Vl + pos * (WCK * VPITCH) + kq * VPITCH + lr;
#pragma unroll
      for (int ks = 0; ks < WCK / 4; ++ks) {
        float a[4], bv[2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = up[ks * 4 * UPITCH + mb * 16];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bv[nb] = vp[ks * 4 * VPITCH + nb * 16];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[pp][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], bv[nb], acc[pp][mb][nb], 0, 0, 0);
      }
    }
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

int wino_launch(ConvK q, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
  static_assert(LDS_M <= LDS_FLOATS, "epilogue buffer overlays the staging area");
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

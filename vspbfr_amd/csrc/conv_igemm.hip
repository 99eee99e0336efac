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
#include "vsp_common.h"

namespace {

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
};

// neutral operands for absent epilogue inputs (read through a zero stride): [0] = 1, [1] = 0, [2] = slope slot
__device__ float kConst[8] = {1.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ int round_pitch(int n, int odd) {
  // smallest p >= n with p % 32 == 16 (unit-stride B reads) or p odd (stride-2 B reads)
  if (odd) return n | 1;
  int p = (n & ~31) + 16;
  return p >= n ? p : p + 32;
}

template <int MB, int NB, int WM, int WN, int CK, int WK, int PMAX>
__global__ __launch_bounds__(64 * WM * WN * WK) void conv_igemm_kernel(const ConvK p) {
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

  const int tile = blockIdx.x;
  const int tx_i = tile % p.tiles_x, ty_i = tile / p.tiles_x;
  const int g = blockIdx.y / p.co_tiles;
  const int co0 = (blockIdx.y % p.co_tiles) * CO_T;  // within the group
  const int b = blockIdx.z;

  const int TW = 1 << p.tw_log2, TH = p.th;
  const int D = p.dil[g];
  const int T = p.KH * p.KW;
  const int PH = (TH - 1) * p.sy + (p.KH - 1) * D + 1;
  const int PW = (TW - 1) * p.sx + (p.KW - 1) * D + 1;
  const int PS = round_pitch(PH * PW, p.ps_odd);
  const int oy0 = ty_i * TH, ox0 = tx_i * TW;
  const int iy0 = oy0 * p.sy - p.pady[g], ix0 = ox0 * p.sx - p.padx[g];

  float* Wl = smem;                // [T][CK][WS]
  float* Pl = smem + T * CK * WS;  // [CK][PS]

  // per-lane patch offsets of the NB pixel blocks this wave owns
  int pixoff[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (wn * NB + nb) * 16 + lr;
    const int py = n >> p.tw_log2, px = n & (TW - 1);
    pixoff[nb] = py * p.sy * PW + px * p.sx + (kq + 4 * wk) * PS;
  }
  const int a_lane = (kq + 4 * wk) * WS + wm * MB * 16 + lr;

  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* xb = p.x + (int64_t)b * p.Cin * p.H * p.W;
  const float* wg = p.w + (int64_t)g * T * p.Cin * p.cout_g;
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
  float preg[PCH][PMAX];
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
    poff[e] = in ? off : 0;
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
    woff[w] = (ok && co0 + c4 * 4 < p.cout_g) ? (tap * p.Cin + cl) * p.cout_g + co0 + c4 * 4 : -1;
  }

  auto issue = [&](int ci0) {
    if (p.w_vec4) {
      const float* wc = wg + (int64_t)ci0 * p.cout_g;
#pragma unroll
      for (int w = 0; w < WMAX; ++w) {
        const int row = (tid + w * NT) / V;
        const int cl = row & (CK - 1);
        const bool ok = woff[w] >= 0 && ci0 + cl < p.Cin;
        const float4 v = *reinterpret_cast<const float4*>(wc + (ok ? woff[w] : 0));
        wreg[w] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int pc = 0; pc < PCH; ++pc) {
      const int cl = wave + pc * NW;
      const int ci = ci0 + cl;
      const bool chok = cl < CK && ci < p.Cin;  // wave-uniform
      const int cic = chok ? ci : 0;
      psc[pc] = chok ? (p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_bstride + cic] : 1.f) : 0.f;
      psh[pc] = chok ? (p.in_shift ? p.in_shift[cic] : 0.f) : 0.f;
      const float* xc = xb + (int64_t)cic * chw;
#pragma unroll
      for (int e = 0; e < PMAX; ++e) preg[pc][e] = xc[poff[e]];
    }
  };

  auto commit = [&](int ci0) {
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
        const bool chok = ci < p.Cin;
        const float* xc = xb + (int64_t)(chok ? ci : 0) * chw;
#pragma unroll 4
        for (int i = lane + 64 * PMAX; i < plane; i += 64) {
          int off;
          float v = 0.f;
          if (chok && patch_src(i, off)) v = fmaf(xc[off], psc[pc], psh[pc]);
          dst[i] = v;
        }
      }
    }
  };

  issue(0);
  for (int ci0 = 0; ci0 < p.Cin; ci0 += CK) {
    __syncthreads();  // previous chunk's fragment reads are done
    commit(ci0);
    __syncthreads();
    if (ci0 + CK < p.Cin) issue(ci0 + CK);
    // ---- MFMA over taps x (CK/4) k-steps
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

  int yoff[NB];  // < 0: pixel outside the image
  float nz[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (wn * NB + nb) * 16 + lr;
    const int oy = oy0 + (n >> p.tw_log2), ox = ox0 + (n & (TW - 1));
    const bool ok = (oy < p.OH && ox < p.OW);
    const int oyc = ok ? oy : 0, oxc = ok ? ox : 0;
    const int off = (oyc * p.osy + p.ooy) * p.y_w + oxc * p.osx + p.oox;
    nz[nb] = nzp[(oyc * p.OW + oxc) * nzs] * nw;
    yoff[nb] = ok ? off : -1;
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float os[4], cs[4], cb[4], b1[4], b2[4], sl2[4];
    int cbase[4];  // channel plane offset, < 0: channel outside the group
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cg = co0 + (wm * MB + mb) * 16 + kq * 4 + r;  // channel within the group
      const bool ok = cg < p.cout_g;
      const int co = g * p.cout_g + (ok ? cg : 0);
      os[r] = osp[co * oss];
      cs[r] = p.csp[co * css];
      cb[r] = p.cbp[co * cbs];
      b1[r] = p.b1p[co * b1s];
      b2[r] = p.b2p[co * b2s];
      sl2[r] = p.s2p[co * s2s];
      cbase[r] = ok ? co * y_plane : -1;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int yo = yoff[nb] < 0 ? 0 : yoff[nb];
      float r1v[4], r2v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ro = (cbase[r] < 0 ? 0 : cbase[r]) + yo;
        r1v[r] = r1b[ro * r1s];
        r2v[r] = r2b[ro * r2s];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[mb][nb][r] * os[r];
        v = v * cs[r] + cb[r];
        v += b1[r];
        v = (v > 0.f ? v : v * s1) * g1;
        v += nz[nb];
        v += b2[r];
        v = (v > 0.f ? v : v * sl2[r]) * g2;
        v += r1v[r];
        v += r2v[r];
        if (yoff[nb] >= 0 && cbase[r] >= 0) yb[cbase[r] + yo] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side: tile configuration table and selection
// ---------------------------------------------------------------------------------------------------------
struct Cfg {
  int MB, NB, WM, WN, CK, WK, PMAX;
  const char* name;
  void (*kern)(const ConvK);
};

#define VSP_CFG(MB, NB, WM, WN, CK) \
  { MB, NB, WM, WN, CK, 1, 12, #MB "x" #NB "x" #WM "x" #WN "x" #CK, conv_igemm_kernel<MB, NB, WM, WN, CK, 1, 12> }
#define VSP_CFGK(MB, NB, WM, WN, CK, WK, PMAX) \
  { MB, NB, WM, WN, CK, WK, PMAX, #MB "x" #NB "x" #WM "x" #WN "x" #CK "k" #WK, conv_igemm_kernel<MB, NB, WM, WN, CK, WK, PMAX> }

static const Cfg kCfgs[] = {
    VSP_CFG(4, 4, 1, 4, 8),  // 0:  64 co x 256 px   (work-horse: C >= 64 at >= 32^2)
    VSP_CFG(4, 4, 2, 2, 8),  // 1: 128 co x 128 px
    VSP_CFG(2, 8, 1, 4, 8),  // 2:  32 co x 512 px   (Cout = 32: 1024^2 level of the StyleGAN prior, dilated @256^2)
    VSP_CFG(1, 8, 1, 4, 8),  // 3:  16 co x 512 px   (Cout = 16 dilated branches @512^2, ToRGB)
    VSP_CFG(4, 1, 1, 4, 8),  // 4:  64 co x  64 px   (8x8 maps)
    VSP_CFG(2, 1, 4, 1, 8),  // 5: 128 co x  16 px   (4x4 and smaller maps)
    VSP_CFG(2, 4, 1, 4, 8),  // 6:  32 co x 256 px
    VSP_CFG(1, 4, 1, 4, 8),  // 7:  16 co x 256 px
    VSP_CFG(4, 4, 1, 4, 4),  // 8:  64 co x 256 px, 4-channel chunks (Cin = 3/4, or very large halos)
    VSP_CFG(1, 8, 1, 4, 4),  // 9:  16 co x 512 px, 4-channel chunks
    VSP_CFG(4, 2, 2, 2, 8),  // 10: 128 co x 64 px
    VSP_CFG(1, 1, 4, 1, 8),  // 11:  64 co x 16 px
    // K-split configurations for deep-K layers on tiny maps (4 wave slices along K, 32-channel chunks)
    VSP_CFGK(1, 1, 1, 1, 32, 4, 2),  // 12: 16 co x 16 px
    VSP_CFGK(1, 4, 1, 1, 32, 4, 2),  // 13: 16 co x 64 px
    VSP_CFGK(2, 4, 1, 1, 32, 4, 2),  // 14: 32 co x 64 px
    VSP_CFGK(2, 2, 1, 2, 32, 2, 4),  // 15: 32 co x 64 px, 2 slices
    VSP_CFGK(2, 1, 1, 1, 32, 4, 2),  // 16: 32 co x 16 px
    VSP_CFGK(1, 2, 1, 2, 32, 2, 4),  // 17: 16 co x 64 px, 2 slices
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

constexpr size_t kMaxLds = 100 * 1024;  // > 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize (set once per kernel)

struct Plan {
  int cfg;
  int tw_log2, th, tiles_x, tiles_y, co_tiles;
  size_t lds;
};

static int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static int host_round_pitch(int n, int odd) {
  if (odd) return n | 1;
  int p = (n & ~31) + 16;
  return p >= n ? p : p + 32;
}

// Fill the geometry of configuration c for problem p; returns false if it does not fit (LDS / index limits).
static bool make_plan(const vsp_conv_params& p, int c, Plan* out) {
  const Cfg& k = kCfgs[c];
  const int CO_T = 16 * k.MB * k.WM, NPIX = 16 * k.NB * k.WN;
  const int WS = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;
  int twl = ilog2_ceil(p.OW);
  // cap the tile width: 16-pixel MFMA column blocks want >= 16 contiguous pixels; wider tiles cut halo re-reads
  int cap = NPIX >= 256 ? 5 : 4;  // 32 or 16
  if (twl > cap) twl = cap;
  if ((1 << twl) > NPIX) twl = ilog2_ceil(NPIX);
  const int TW = 1 << twl, TH = NPIX / TW;
  int dmax = 1;
  for (int g = 0; g < p.G; ++g) dmax = p.dil[g] > dmax ? p.dil[g] : dmax;
  const int PH = (TH - 1) * p.stride_y + (p.KH - 1) * dmax + 1;
  const int PW = (TW - 1) * p.stride_x + (p.KW - 1) * dmax + 1;
  if (PW > 256 || PH * PW >= 65536) return false;
  const int PS = host_round_pitch(PH * PW, p.stride_x != 1);
  size_t lds = ((size_t)p.KH * p.KW * k.CK * WS + (size_t)k.CK * PS) * sizeof(float);
  const size_t red = (size_t)(k.WK - 1) * k.WM * k.WN * k.MB * k.NB * 4 * 64 * sizeof(float);
  if (red > lds) lds = red;
  if (lds > kMaxLds) return false;
  out->cfg = c;
  out->tw_log2 = twl;
  out->th = TH;
  out->tiles_x = (p.OW + TW - 1) / TW;
  out->tiles_y = (p.OH + TH - 1) / TH;
  out->co_tiles = (p.cout_g + CO_T - 1) / CO_T;
  out->lds = lds;
  return true;
}

// Cost model: MFMA slots issued (padded tile work), in waves of resident blocks over 256 CUs, plus a per-chunk
// staging term.  Only relative order matters.
static double plan_cost(const vsp_conv_params& p, const Plan& pl) {
  const Cfg& k = kCfgs[pl.cfg];
  const int CO_T = 16 * k.MB * k.WM, NPIX = 16 * k.NB * k.WN;
  const int waves = k.WM * k.WN * k.WK;
  const double blocks = (double)pl.tiles_x * pl.tiles_y * pl.co_tiles * p.G * p.B;
  const int cin_pad = (p.Cin + k.CK - 1) / k.CK * k.CK;
  // cycles one block needs on one SIMD-set: each wave issues MB*NB MFMAs (32 cyc) per k-step
  const double ksteps = (double)p.KH * p.KW * cin_pad / 4.0;
  const double mfma_cyc = ksteps * k.MB * k.NB * 32.0 / k.WK;  // per wave
  const double reads = ksteps * (k.MB + k.NB) * 8.0 / k.WK;  // LDS issue cost per wave, overlappable: small weight
  const double chunks = (double)cin_pad / k.CK;
  const double stage = chunks * 2500.0;                  // barrier + global latency + LDS writes per chunk
  int per_cu = (int)(160 * 1024 / (pl.lds + 1024));
  const int by_waves = 8 / waves * 4 / 4;  // keep <= 2 waves per SIMD per block set: 8 waves/CU when 4-wave blocks
  (void)by_waves;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  // waves per SIMD when the CU is full
  const double wps = per_cu * waves / 4.0;
  // block time when co-resident with (per_cu-1) others: MFMA pipe shared, staging hidden if wps >= 2
  double t_block = mfma_cyc * (wps < 1.0 ? 1.0 : wps) + (wps >= 2.0 ? 0.15 : 1.0) * stage + 0.05 * reads;
  const double slots = 256.0 * per_cu;
  const double rounds = blocks / slots;
  const double full = (rounds < 1.0) ? 1.0 : rounds;  // a partially filled chip still takes one block time
  // when the chip is under-filled the blocks do not share SIMDs: undo the sharing factor
  if (rounds < 1.0) {
    const double occ = blocks / 256.0;  // blocks per CU
    const double w = occ * waves / 4.0;
    t_block = mfma_cyc * (w < 1.0 ? 1.0 : w) + stage + 0.05 * reads;
  }
  (void)CO_T;
  (void)NPIX;
  return full * t_block;
}

// Address of kConst on the current device (resolved once per device; the first conv call of a process must not
// happen inside a stream capture -- the Python loader makes a warm-up call at import).
static const float* device_consts() {
  static const float* cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    vsp::set_error("hipGetDevice failed");
    return nullptr;
  }
  if (!cache[dev]) {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(kConst));
    if (e != hipSuccess) {
      vsp::set_error("hipGetSymbolAddress(kConst): %s", hipGetErrorString(e));
      return nullptr;
    }
    // slots: [0]=1 [1]=0 [2]=1 [3]=0 [4]=0.2 [5]=0.01 (constant slopes)
    const float init[8] = {1.f, 0.f, 1.f, 0.f, 0.2f, 0.01f, 0.f, 0.f};
    e = hipMemcpy(ptr, init, sizeof(init), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      vsp::set_error("hipMemcpy(kConst): %s", hipGetErrorString(e));
      return nullptr;
    }
    cache[dev] = static_cast<const float*>(ptr);
  }
  return cache[dev];
}

static const float* slope_slot(const float* kc, float slope) {
  if (slope == 0.2f) return kc + 4;
  if (slope == 0.01f) return kc + 5;
  if (slope == 1.f) return kc;
  if (slope == 0.f) return kc + 1;
  return nullptr;
}

}  // namespace

extern "C" int vsp_conv2d_num_configs(void) { return kNumCfgs; }
extern "C" const char* vsp_conv2d_config_name(int i) { return (i >= 0 && i < kNumCfgs) ? kCfgs[i].name : ""; }

extern "C" int vsp_conv2d_f32(const vsp_conv_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d: null params");
  const vsp_conv_params& p = *pp;
  VSP_REQUIRE(p.x && p.w && p.y, "conv2d: null tensor pointer");
  VSP_REQUIRE(p.B >= 0 && p.Cin >= 1 && p.H >= 1 && p.W >= 1, "conv2d: bad input dims B=%d Cin=%d H=%d W=%d", p.B,
              p.Cin, p.H, p.W);
  VSP_REQUIRE(p.G >= 1 && p.G <= 4 && p.cout_g >= 1, "conv2d: bad group spec G=%d cout_g=%d", p.G, p.cout_g);
  VSP_REQUIRE(p.KH >= 1 && p.KW >= 1 && p.KH * p.KW <= 49, "conv2d: unsupported kernel %dx%d", p.KH, p.KW);
  VSP_REQUIRE(p.stride_y >= 1 && p.stride_x >= 1 && p.stride_x <= 2 && p.stride_y <= 2, "conv2d: stride must be 1 or 2");
  VSP_REQUIRE(p.OH >= 0 && p.OW >= 0, "conv2d: negative output size");
  VSP_REQUIRE(p.osy >= 1 && p.osx >= 1 && p.ooy >= 0 && p.oox >= 0, "conv2d: bad output stride/offset");
  VSP_REQUIRE(!p.noise || p.noise_w, "conv2d: noise given without noise_w");
  VSP_REQUIRE(p.act2 != 2 || p.prelu, "conv2d: act2=prelu without slopes");
  for (int g = 0; g < p.G; ++g) VSP_REQUIRE(p.dil[g] >= 1, "conv2d: dilation must be >= 1");
  if (p.B == 0 || p.OH == 0 || p.OW == 0) return VSP_OK;
  const int Cout = p.G * p.cout_g;
  VSP_REQUIRE(p.y_coff >= 0 && p.y_coff + Cout <= p.y_ch, "conv2d: output channel window [%d,%d) outside %d", p.y_coff,
              p.y_coff + Cout, p.y_ch);
  VSP_REQUIRE((p.OH - 1) * p.osy + p.ooy < p.y_h && (p.OW - 1) * p.osx + p.oox < p.y_w,
              "conv2d: output positions exceed the %dx%d output tensor", p.y_h, p.y_w);
  VSP_REQUIRE((int64_t)p.y_ch * p.y_h * p.y_w < ((int64_t)1 << 31) && (int64_t)p.Cin * p.H * p.W < ((int64_t)1 << 31),
              "conv2d: one image must hold fewer than 2^31 elements");
  if (p.res1 || p.res2)
    VSP_REQUIRE(p.res_coff >= 0 && p.res_coff + Cout <= p.res_ch, "conv2d: residual channel window out of range");

  Plan best{};
  bool found = false;
  if (p.tile_hint > 0) {
    VSP_REQUIRE(p.tile_hint <= kNumCfgs, "conv2d: tile_hint %d out of range", p.tile_hint);
    found = make_plan(p, p.tile_hint - 1, &best);
    VSP_REQUIRE(found, "conv2d: configuration %s does not fit this problem", kCfgs[p.tile_hint - 1].name);
  } else {
    double best_cost = 0.0;
    for (int c = 0; c < kNumCfgs; ++c) {
      Plan pl{};
      if (!make_plan(p, c, &pl)) continue;
      const double cost = plan_cost(p, pl);
      if (!found || cost < best_cost) {
        best = pl;
        best_cost = cost;
        found = true;
      }
    }
    if (!found) return vsp::fail(VSP_ENOTSUP, "conv2d: no tile configuration fits (KH=%d dil=%d)", p.KH, p.dil[0]);
  }
  const Cfg& k = kCfgs[best.cfg];
  const int CO_T = 16 * k.MB * k.WM;
  const int64_t gy = (int64_t)best.co_tiles * p.G;
  VSP_REQUIRE(gy <= 65535 && p.B <= 65535, "conv2d: grid too large");

  ConvK q{};
  q.x = p.x; q.w = p.w; q.y = p.y;
  q.B = p.B; q.Cin = p.Cin; q.H = p.H; q.W = p.W; q.G = p.G; q.cout_g = p.cout_g;
  q.OH = p.OH; q.OW = p.OW; q.KH = p.KH; q.KW = p.KW; q.sy = p.stride_y; q.sx = p.stride_x;
  for (int g = 0; g < 4; ++g) { q.dil[g] = p.dil[g]; q.pady[g] = p.pad_y[g]; q.padx[g] = p.pad_x[g]; }
  q.y_ch = p.y_ch; q.y_coff = p.y_coff; q.y_h = p.y_h; q.y_w = p.y_w;
  q.osy = p.osy; q.osx = p.osx; q.ooy = p.ooy; q.oox = p.oox;
  q.in_scale = p.in_scale; q.in_scale_bstride = p.in_scale_bstride; q.in_shift = p.in_shift;
  const float* kc = device_consts();
  VSP_REQUIRE(kc != nullptr, "conv2d: cannot resolve device constants: %s", vsp_last_error());
  const float* kOne = kc;
  const float* kZero = kc + 1;
  auto sel = [](const float* ptr, const float* dflt, const float** outp, int* outs) {
    *outp = ptr ? ptr : dflt;
    *outs = ptr ? 1 : 0;
  };
  sel(p.out_scale, kOne, &q.osp, &q.oss);
  sel(p.ch_scale, kOne, &q.csp, &q.css);
  sel(p.ch_bias, kZero, &q.cbp, &q.cbs);
  sel(p.act1 ? p.bias1 : nullptr, kZero, &q.b1p, &q.b1s);
  q.s1 = p.act1 ? p.slope1 : 1.f;
  q.g1 = p.act1 ? p.gain1 : 1.f;
  sel(p.noise, kZero, &q.nzp, &q.nzs);
  q.nwp = p.noise ? p.noise_w : kZero;
  sel(p.act2 == 1 ? p.bias2 : nullptr, kZero, &q.b2p, &q.b2s);
  if (p.act2 == 2) {
    q.s2p = p.prelu; q.s2s = 1; q.g2 = 1.f;
  } else if (p.act2 == 1) {
    // constant slope: staged in the per-launch slot of the constant block is not possible (async launches share
    // it), so the two slopes used by the path live in fixed slots: 0.2 (FusedLeakyReLU) and 0.01 (nn.LeakyReLU)
    const float* slot = slope_slot(kc, p.slope2);
    VSP_REQUIRE(slot != nullptr, "conv2d: unsupported act2 slope %g (supported: 0.2, 0.01, 0, 1)", p.slope2);
    q.s2p = slot; q.s2s = 0; q.g2 = p.gain2;
  } else {
    q.s2p = kOne; q.s2s = 0; q.g2 = 1.f;
  }
  sel(p.res1, kZero, &q.r1p, &q.r1s);
  sel(p.res2, kZero, &q.r2p, &q.r2s);
  q.res_ch = p.res_ch; q.res_coff = p.res_coff;
  q.tw_log2 = best.tw_log2; q.th = best.th; q.tiles_x = best.tiles_x; q.tiles_y = best.tiles_y;
  q.co_tiles = best.co_tiles;
  q.w_vec4 = (p.cout_g % 4 == 0) && (CO_T % 4 == 0) && vsp::aligned16(p.w) ? 1 : 0;
  q.ps_odd = p.stride_x != 1;

  if (best.lds > 64 * 1024) {
    static bool raised[64] = {};
    if (!raised[best.cfg]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k.kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)kMaxLds);
      if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d: cannot raise the LDS limit: %s", hipGetErrorString(e));
      raised[best.cfg] = true;
    }
  }
  dim3 grid((unsigned)(best.tiles_x * best.tiles_y), (unsigned)gy, (unsigned)p.B);
  dim3 block(64 * k.WM * k.WN * k.WK);
  hipLaunchKernelGGL(k.kern, grid, block, best.lds, vsp::as_stream(stream), q);
  return vsp::check_launch("conv2d");
}

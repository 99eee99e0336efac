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
//   * dilation d is a polyphase problem exactly as in conv_wino.hip: the workgroup addresses the image with stride d from
//     its residue (ry, rx), in LDS every layer is a dilation-1 convolution; the (up to four) dilation groups of a SMART
//     branch launch differ only in d and their weight / channel base.
// Numerics: operands rounded to bf16 (RNE, 8 significant bits), products exact, accumulation fp32; the epilogue chain is
// the fp32 one of the direct kernel.  Not a parity path: tests bound its error against the fp32 kernels.
#include "conv_kernel.h"

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

template <int MB, int NB, int WM, int WN, int PT>
__global__ __launch_bounds__(BNT, 2) void conv_bf16_kernel(const ConvK p) {
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int CO_T = 32 * MB * WM, NPIX = 32 * NB * WN, T = 9;
  constexpr int WSLAB = T * 2 * CO_T;  // 16-byte units per weight buffer: [tap][octet][co]
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l32 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.z;
  const int g = blockIdx.y / p.co_tiles, ct = blockIdx.y - g * p.co_tiles;
  const int d = p.dil[p.G > 4 ? 0 : g];
  const int SH = (p.H + d - 1) / d, SW = (p.W + d - 1) / d;   // sub-image of one residue class
  const int twl = p.tw_log2, TW = 1 << twl, TH = NPIX >> twl;
  const int tiles_x = (SW + TW - 1) >> twl, tiles_y = (SH + TH - 1) / TH;
  const int per_res = tiles_x * tiles_y;
  if ((int)blockIdx.x >= per_res * d * d) return;              // groups with a smaller dilation have fewer, fuller tiles
  const int res = blockIdx.x / per_res, tile_i = blockIdx.x - res * per_res;
  const int ry = res / d, rx = res - ry * d;
  const int ty_i = tile_i / tiles_x, tx_i = tile_i - ty_i * tiles_x;
  const int oy0 = ty_i * TH, ox0 = tx_i << twl;                // sub-image coordinates
  const int co0 = ct * CO_T;
  const int co_pad = (p.cout_g + 31) & ~31;
  const int chw = p.H * p.W;
  const int nchunk = (p.Cin + BCK - 1) / BCK;
  const int pitch = p.bf_pitch, PR = TH + 2, PC = TW + 2, PLANE = PR * pitch;

  u32x4* Wl = smem16;               // 2 x [T][2][CO_T]
  u32x4* Pl = smem16 + 2 * WSLAB;   // 2 x [2][PLANE]

  // ---- patch staging: wave w serves channel octet (w & 1), positions (w >> 1) * 64 + lane + 128 e
  const int oct = wave & 1;
  const int pbase = (wave >> 1) * 64 + lane;
  int poff[PT];
  unsigned pin = 0, pwr = 0;  // bit e: position inside the image / position exists in the plane
#pragma unroll
  for (int e = 0; e < PT; ++e) {
    const int idx = pbase + 128 * e;
    const int r = idx / pitch, c = idx - r * pitch;
    const bool wr = idx < PLANE && c < PC;
    const int sy = oy0 - 1 + r, sx = ox0 - 1 + c;
    const int iy = sy * d + ry, ix = sx * d + rx;
    const bool in = wr && sy >= 0 && sx >= 0 && iy < p.H && ix < p.W;
    poff[e] = in ? iy * p.W + ix : 0;
    pin |= in ? (1u << e) : 0u;
    pwr |= wr ? (1u << e) : 0u;
  }
  const float* xb = p.x + ((int64_t)b * p.x_ch + (int64_t)g * p.x_gs) * chw;
  float preg[PT][8];
  auto issue_p = [&](int c) {
    const int cib = c * BCK + 8 * oct;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = cib + j;
      const float* xc = xb + (int64_t)(ci < p.Cin ? ci : 0) * chw;  // wave-uniform base
#pragma unroll
      for (int e = 0; e < PT; ++e)
        if (128 * e < PLANE) preg[e][j] = xc[poff[e]];
    }
  };
  auto commit_p = [&](u32x4* Pdst, int c) {
    const int cib = c * BCK + 8 * oct;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = cib + j;
      const bool chok = ci < p.Cin;
      sc[j] = chok ? (p.in_scale ? p.in_scale[(int64_t)b * p.in_scale_bstride + ci] : 1.f) : 0.f;
      sh[j] = (chok && p.in_shift) ? p.in_shift[ci] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < PT; ++e) {
      if (!((pwr >> e) & 1u)) continue;
      const bool in = (pin >> e) & 1u;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = in ? fmaf(preg[e][j], sc[j], sh[j]) : 0.f;
      Pdst[oct * PLANE + pbase + 128 * e] =
          u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
    }
  };

  // ---- weight slab by LDS-DMA: the chunk's [tap][octet] rows of this co tile, 64 rows (1 KiB) per wave instruction
  constexpr int NDMA = WSLAB / 64;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w) + (int64_t)g * nchunk * (T * 2) * co_pad;
  auto issue_w = [&](u32x4* Wdst, int c) {
    const u32x4* src = wsrc + (int64_t)c * (T * 2) * co_pad;
    for (int i = wave; i < NDMA; i += 4) {
      const int L = i * 64 + lane;
      const int row = L / CO_T, co = L - row * CO_T;
      if (co0 + co < co_pad)
        __builtin_amdgcn_global_load_lds(src + row * co_pad + co0 + co, Wdst + i * 64, 16, 0, 0);
    }
  };

  // ---- fragments
  int pixpos[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (wn * NB + nb) * 32 + l32;
    pixpos[nb] = (n >> twl) * pitch + (n & (TW - 1)) + kh * PLANE;
  }
  const int a_lane = kh * CO_T + wm * MB * 32 + l32;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  // ---- pipeline: at the top of interval c, W[c&1] and P[c&1] hold chunk c (made visible by the closing barrier of c-1)
  issue_w(Wl, 0);
  issue_p(0);
  commit_p(Pl, 0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int cur = c & 1, nxt = cur ^ 1;
    const bool more = c + 1 < nchunk;
    if (more) {
      issue_w(Wl + nxt * WSLAB, c + 1);
      issue_p(c + 1);
    }
    const u32x4* Wc = Wl + cur * WSLAB + a_lane;
    const u32x4* Pc = Pl + cur * 2 * PLANE;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int toff = (tap / 3) * pitch + (tap % 3);
      bf16x8 a[MB], bq[NB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = __builtin_bit_cast(bf16x8, Wc[tap * 2 * CO_T + mb * 32]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[nb] = __builtin_bit_cast(bf16x8, Pc[pixpos[nb] + toff]);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb], bq[nb], acc[mb][nb], 0, 0, 0);
    }
    if (more) commit_p(Pl + nxt * 2 * PLANE, c + 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds pixel l32 of block nb; accumulator i is channel 8 (i >> 2) + 4 kh + (i & 3) of block mb
  const int Cout = p.G * p.cout_g;
  const float* osp = p.osp + (int64_t)b * Cout * p.oss;
  const float* nzp = p.nzp + (int64_t)b * p.OH * p.OW * p.nzs;
  const float nw = p.nwp[0];
  const float s1 = p.s1, g1 = p.g1, g2 = p.g2;
  float* yb = p.y + ((int64_t)b * p.y_ch + p.y_coff) * p.y_h * p.y_w;
  const float* r1b = p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r1s;
  const float* r2b = p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * p.y_h * p.y_w * p.r2s;
  const int r1s = p.r1s, r2s = p.r2s;
  const int y_plane = p.y_h * p.y_w;
  int yoff[NB];
  float nz[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = (wn * NB + nb) * 32 + l32;
    const int oy = (oy0 + (n >> twl)) * d + ry, ox = (ox0 + (n & (TW - 1))) * d + rx;
    const bool ok = oy < p.OH && ox < p.OW;
    yoff[nb] = ok ? oy * p.y_w + ox : -1;
    nz[nb] = nzp[(ok ? oy * p.OW + ox : 0) * p.nzs] * nw;
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cg = co0 + (wm * MB + mb) * 32 + 8 * (i >> 2) + 4 * kh + (i & 3);
      const bool cok = cg < p.cout_g;
      const int co = g * p.cout_g + (cok ? cg : 0);
      const float os = osp[co * p.oss], cs = p.csp[co * p.css], cb = p.cbp[co * p.cbs];
      const float b1 = p.b1p[co * p.b1s], b2 = p.b2p[co * p.b2s], sl2 = p.s2p[co * p.s2s];
      const int cbase = co * y_plane;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int ro = cbase + (yoff[nb] < 0 ? 0 : yoff[nb]);
        const float r1v = r1b[ro * r1s];
        const float r2v = r2b[ro * r2s];
        float v = acc[mb][nb][i] * os;
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

struct BfGeom {
  int twl, pitch, plane, pt;
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

static BfGeom bf_geom(const ConvK& q, int co_t, int npix) {
  int dmin = q.dil[0];
  for (int g = 1; g < (q.G > 4 ? 1 : q.G); ++g) dmin = q.dil[g] < dmin ? q.dil[g] : dmin;
  int dmax = q.dil[0];
  for (int g = 1; g < (q.G > 4 ? 1 : q.G); ++g) dmax = q.dil[g] > dmax ? q.dil[g] : dmax;
  const int sw = (q.W + dmax - 1) / dmax;  // the narrowest sub-image decides the tile width
  BfGeom r;
  r.twl = sw >= 32 ? 5 : (sw > 8 ? 4 : 3);
  const int TW = 1 << r.twl, TH = npix >> r.twl;
  r.pitch = bf_pitch(r.twl, TW + 2);
  r.plane = (TH + 2) * r.pitch;
  r.pt = (r.plane + 127) / 128;
  r.lds = ((size_t)2 * 9 * 2 * co_t + (size_t)2 * 2 * r.plane) * 16;
  (void)dmin;
  return r;
}

template <int MB, int NB, int WM, int WN, int PT>
int launch_bf(ConvK q, const BfGeom& gm, hipStream_t stream) {
  constexpr int CO_T = 32 * MB * WM, NPIX = 32 * NB * WN;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_kernel<MB, NB, WM, WN, PT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return vsp::fail(VSP_ELAUNCH, "conv2d_bf16: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  q.tw_log2 = gm.twl;
  q.bf_pitch = gm.pitch;
  q.co_tiles = (q.cout_g + CO_T - 1) / CO_T;
  const int TW = 1 << gm.twl, TH = NPIX >> gm.twl;
  int blocks = 0;
  for (int g = 0; g < (q.G > 4 ? 1 : q.G); ++g) {
    const int d = q.dil[g];
    const int SH = (q.H + d - 1) / d, SW = (q.W + d - 1) / d;
    const int n = ((SW + TW - 1) / TW) * ((SH + TH - 1) / TH) * d * d;
    blocks = n > blocks ? n : blocks;
  }
  dim3 grid((unsigned)blocks, (unsigned)(q.co_tiles * q.G), (unsigned)q.B);
  conv_bf16_kernel<MB, NB, WM, WN, PT><<<grid, BNT, gm.lds, stream>>>(q);
  return VSP_OK;
}

template <int MB, int NB, int WM, int WN>
int launch_shape(const ConvK& q, hipStream_t stream) {
  const BfGeom gm = bf_geom(q, 32 * MB * WM, 32 * NB * WN);
  if (gm.lds > 150 * 1024) return vsp::fail(VSP_ENOTSUP, "conv2d_bf16: tile does not fit LDS");
  if (gm.pt <= 3) return launch_bf<MB, NB, WM, WN, 3>(q, gm, stream);
  if (gm.pt <= 7) return launch_bf<MB, NB, WM, WN, 7>(q, gm, stream);
  return vsp::fail(VSP_ENOTSUP, "conv2d_bf16: patch plane of %d positions is too large", gm.plane);
}

}  // namespace

// variant: 0 = automatic, 1 = 32 ch x 256 px, 2 = 64 ch x 256 px, 3 = 128 ch x 128 px, 4 = 64 ch x 128 px
int bf16_launch(const ConvK& q, int variant, hipStream_t stream) {
  if (variant == 0) {
    const int64_t px = (int64_t)q.H * q.W;
    if (q.cout_g <= 32) variant = 1;
    else if (q.cout_g <= 64) variant = px >= 128 * 128 ? 2 : 4;
    else variant = px >= 128 * 128 ? 3 : 4;
  }
  switch (variant) {
    case 1: return launch_shape<1, 2, 1, 4>(q, stream);
    case 2: return launch_shape<2, 2, 1, 4>(q, stream);
    case 3: return launch_shape<2, 2, 2, 2>(q, stream);
    case 4: return launch_shape<2, 1, 1, 4>(q, stream);
    default: return vsp::fail(VSP_EINVAL, "conv2d_bf16: unknown variant %d", variant);
  }
}

}  // namespace vspconv

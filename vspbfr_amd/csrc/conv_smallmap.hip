// Convolution on SMALL maps (a few hundred output positions in the whole batch) as a K-split GEMM, same contract and epilogue as
// vsp_conv2d_f32.  The 4x4 ... 16x16 levels of Restoration_net / the StyleGAN2 prior / the style heads have K = Cin KH KW of
// 4608 against 8 ... 2048 columns: the tiled kernel (conv_kernel.h) walks K in LDS-staged chunks inside ONE workgroup per output
// tile -- two barriers and a global round trip per chunk, 50 - 130 us for layers whose data is one pass over 9 MB of weights.
// Here: rows of the GEMM = output positions (b, oy, ox), columns = output channels; both MFMA operands are loaded from global
// memory straight into fragments (A: x at the tap's shifted position, zero outside, input scale / shift applied in registers;
// B: packed weights Wp[g][tap][ci][co], 16 consecutive channels per row), K = (tap, ci) is split over the 8 wavefronts of a
// workgroup in 16-channel sub-steps with the next sub-steps prefetched in registers, and the partial tiles meet once in LDS.
#include "conv_kernel.h"
#include "vsp_common.h"

namespace vspconv {
namespace {

constexpr int kSW = 8;   // wavefronts per workgroup = K split

// wave tile 16 MI positions x 16 NI channels; KU sub-steps (16 input channels of one tap) in flight.
// VEC: the NI column tiles interleave -- tile ni holds channels co0 + NI r + ni -- so that ONE NI-float load per lane and weight row
// feeds all NI tiles (cout_g a multiple of NI, 16-byte aligned weights); otherwise tile ni = channels co0 + 16 ni + r, scalar loads.
// (The kernel is bound by instruction issue, not by memory: ~22 instructions per MFMA with scalar weight loads on 16-position tiles.)
template <int MI, int NI, int KU, bool VEC>
__global__ __launch_bounds__(64 * kSW) void conv_smallmap_kernel(const ConvK p, const int N, const int co_tiles) {
  __shared__ float red[kSW][MI * NI * 4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
  // XCD-aware order: the dispatcher deals workgroups round-robin over the eight XCDs (each with its own L2), so the row tiles of ONE channel
  // tile -- which read the same weights -- landed on all eight and every XCD fetched the whole weight set: FETCH_SIZE 0.86 GB for the
  // 104 MB of the 512 -> 5632 level-8 launch, 6.2 TB/s of L2 misses for 139 us.  Here every XCD gets a contiguous range of the order
  // row tile (fastest) -> channel tile: a channel tile's weights are fetched by one XCD, once.
  int bx = blockIdx.x, by = blockIdx.y;
  if (!(p.dbg & 0x40000)) {
    const int GX = gridDim.x, GT = GX * gridDim.y;
    const int wgid = blockIdx.x + GX * blockIdx.y;
    const int xcd = wgid & 7, xq = GT >> 3, xr = GT & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wgid >> 3);
    by = lid / GX;
    bx = lid - by * GX;
  }
  const int n0 = bx * 16 * MI;
  const int g = by / co_tiles, co0 = (by - g * co_tiles) * 16 * NI;
  const int gg = p.G > 4 ? 0 : g;   // more than four groups share one geometry
  const int dil = p.dil[gg], pady = p.pady[gg], padx = p.padx[gg];
  const int P = p.OH * p.OW, HW = p.H * p.W, T = p.KH * p.KW;
  const bool scaled = p.in_scale != nullptr, shifted = p.in_shift != nullptr;

  int iy0[MI], ix0[MI];
  const float* xb[MI];
  const float* sb[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int n = min(n0 + 16 * mi + r, N - 1);
    const int b = n / P, q = n - b * P;
    const int oy = q / p.OW, ox = q - oy * p.OW;
    iy0[mi] = oy * p.sy - pady;
    ix0[mi] = ox * p.sx - padx;
    xb[mi] = p.x + ((int64_t)b * p.x_ch + g * p.x_gs + 4 * kq) * HW;
    sb[mi] = scaled ? p.in_scale + (int64_t)b * p.in_scale_bstride + g * p.x_gs + 4 * kq : p.x;
  }
  const float* shb = shifted ? p.in_shift + 4 * kq : p.x;
  typedef float wvec __attribute__((ext_vector_type(NI)));
  const float* wb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    wb[ni] = p.w + ((int64_t)g * T * p.Cin + 4 * kq) * p.cout_g +
             (VEC ? min(co0 + NI * r, p.cout_g - NI) : min(co0 + 16 * ni + r, p.cout_g - 1));   // VEC: only wb[0] is used

  const int c16 = p.Cin >> 4;
  const int S = T * c16;
  // The K walk keeps (tap row, tap column, 16-channel step) as wavefront-uniform counters advanced by 8 KU sub-steps per iteration
  // (no integer division per sub-step).
  struct KPos { int ky, kx, cc; };
  auto advance = [&](KPos& k, int steps) {
    k.cc += steps;
    while (k.cc >= c16) {
      k.cc -= c16;
      if (++k.kx == p.KW) {
        k.kx = 0;
        ++k.ky;
      }
    }
  };
  auto fetch = [&](KPos k, float (*A)[4], float (*Bf)[4]) {   // k: uniform per wavefront; past the end -> the last sub-step (unused)
    if (k.ky >= p.KH) k = KPos{p.KH - 1, p.KW - 1, c16 - 1};
    const int ky = k.ky, kx = k.kx, cc = k.cc, tap = ky * p.KW + kx;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int iy = iy0[mi] + ky * dil, ix = ix0[mi] + kx * dil;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float* xp = xb[mi] + (int64_t)(16 * cc) * HW + (ok ? iy * p.W + ix : 0);
      float v[4], sc[4], sh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = xp[e * HW];
      if (scaled) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[e] = sb[mi][16 * cc + e];
      }
      if (shifted) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sh[e] = shb[16 * cc + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = v[e];
        if (scaled) t *= sc[e];
        if (shifted) t += sh[e];
        A[mi][e] = ok ? t : 0.f;
      }
    }
    const int64_t wo = ((int64_t)tap * p.Cin + 16 * cc) * p.cout_g;
    if constexpr (VEC) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const wvec v = *reinterpret_cast<const wvec*>(wb[0] + wo + (int64_t)e * p.cout_g);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Bf[ni][e] = v[ni];
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) Bf[ni][e] = wb[ni][wo + (int64_t)e * p.cout_g];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[KU][MI][4], bq[KU][NI][4], an[KU][MI][4], bn[KU][NI][4];
  const int iters = (S + kSW * KU - 1) / (kSW * KU);
  KPos kc[KU], kn[KU];   // position of the operands in hand / of the next fetch
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int s0 = wv * KU + u, tap0 = s0 / c16;
    kc[u] = KPos{tap0 / p.KW, tap0 % p.KW, s0 - tap0 * c16};
    fetch(kc[u], a[u], bq[u]);
    kn[u] = kc[u];
    advance(kn[u], kSW * KU);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < KU; ++u) fetch(kn[u], an[u], bn[u]);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (kc[u].ky < p.KH) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mi][e], bq[u][ni][e], acc[mi][ni], 0, 0, 0);
      }
      kc[u] = kn[u];
      advance(kn[u], kSW * KU);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[u][mi][e] = an[u][mi][e];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[u][ni][e] = bn[u][ni][e];
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wv][(mi * NI + ni) * 4 + i][lane] = acc[mi][ni][i];
  __syncthreads();

  // ---- epilogue (the operand conventions of conv_kernel.h: absent operands read a constant through a zero stride)
  const int Cout = p.G * p.cout_g;
  const float nw = p.nwp[0];
  const int y_plane = p.y_h * p.y_w;
  for (int q = wv; q < MI * NI * 4; q += kSW) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kSW; ++k) s += red[k][q][lane];
    const int i = q & 3, ni = (q >> 2) % NI, mi = (q >> 2) / NI;
    const int n = n0 + 16 * mi + 4 * kq + i, cg = VEC ? co0 + NI * r + ni : co0 + 16 * ni + r;
    if (n >= N || cg >= p.cout_g) continue;
    const int b = n / P, pq = n - b * P;
    const int oy = pq / p.OW, ox = pq - oy * p.OW;
    const int co = g * p.cout_g + cg;
    const int64_t ro = ((int64_t)b * p.y_ch + p.y_coff + co) * y_plane + (oy * p.osy + p.ooy) * p.y_w + ox * p.osx + p.oox;
    const int64_t rr = ((int64_t)b * p.res_ch + p.res_coff + co) * y_plane + (oy * p.osy + p.ooy) * p.y_w + ox * p.osx + p.oox;
    float v = s * p.osp[((int64_t)b * Cout + co) * p.oss];
    v = v * p.csp[co * p.css] + p.cbp[co * p.cbs];
    v += p.b1p[co * p.b1s];
    v = (v > 0.f ? v : v * p.s1) * p.g1;
    v += p.nzp[((int64_t)b * P + pq) * p.nzs] * nw;
    v += p.b2p[co * p.b2s];
    v = (v > 0.f ? v : v * p.s2p[co * p.s2s]) * p.g2;
    v += p.r1p[rr * p.r1s];
    v += p.r2p[rr * p.r2s];
    p.y[ro] = v;
  }
}

}  // namespace

bool smallmap_eligible(const ConvK& q, bool transposed) {
  return !transposed && q.io_bf16 == 0 && q.Cin % 16 == 0 && (int64_t)q.B * q.OH * q.OW <= 8192;
}

int smallmap_launch(const ConvK& q, hipStream_t stream) {
  const int N = q.B * q.OH * q.OW;
  const int n16 = (N + 15) / 16, n32 = (N + 31) / 32;
  const bool al = vsp::aligned16(q.w);
  if (al && q.cout_g % 4 == 0) {
    // 16 positions x 64 channels per workgroup when that fills the chip; else 32-channel tiles (twice the workgroups)
    const int ct4 = (q.cout_g + 63) / 64, ct2 = (q.cout_g + 31) / 32;
    if (n16 * ct4 * q.G >= 192)
      conv_smallmap_kernel<1, 4, 2, true><<<dim3(n16, ct4 * q.G), 64 * kSW, 0, stream>>>(q, N, ct4);
    else if (n32 * ct2 * q.G >= 256)
      conv_smallmap_kernel<2, 2, 2, true><<<dim3(n32, ct2 * q.G), 64 * kSW, 0, stream>>>(q, N, ct2);
    else
      conv_smallmap_kernel<1, 2, 2, true><<<dim3(n16, ct2 * q.G), 64 * kSW, 0, stream>>>(q, N, ct2);
    return vsp::check_launch("conv2d (small-map kernel)");
  }
  const int ct = (q.cout_g + 31) / 32;
  if (n32 * ct * q.G >= 256)
    conv_smallmap_kernel<2, 2, 2, false><<<dim3(n32, ct * q.G), 64 * kSW, 0, stream>>>(q, N, ct);
  else
    conv_smallmap_kernel<1, 2, 2, false><<<dim3(n16, ct * q.G), 64 * kSW, 0, stream>>>(q, N, ct);
  return vsp::check_launch("conv2d (small-map kernel)");
}

}  // namespace vspconv

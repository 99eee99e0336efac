// 1x1 convolution on SMALL maps as the GEMM it is:  y[b][co][p] = act(sum_ci W[co][ci] x[b][ci][p] + bias[co]),  P = H W.
// The late stages of the ArcFace ResNet-101 of the identity loss (Loss/id_loss.py:13: 23 bottlenecks at 7x7, 3 at 4x4, batch 8) are
// ~140 such products per training iteration with 128-392 columns and K = 256-2048: the tiled conv kernel walks K in 32 LDS-staged
// chunks per workgroup (two barriers each) and takes 50-125 us where the data is 1-4 MB.  Here the operands go from global memory
// straight into MFMA fragments (no LDS staging, no barrier in the K loop), K is split over the 8 wavefronts of a workgroup and
// reduced once through LDS; the weight is read as it lies (OIHW with 1x1 taps = row-major (Cout, Cin): no packing).
//   A fragment: lane (r, kq) loads W[co0 + r][16 t + 4 kq .. + 3] as one 16-byte load -> element e feeds k-slot kq of MFMA e;
//   B fragment: lane (n, kq) loads x[b(n)][16 t + 4 kq + e][p(n)], e = 0..3 (the same k mapping), 16 consecutive columns per row.
#include "vsp_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;

template <int MI, int NI>   // wave tile = 16 MI output channels x 16 NI columns
__global__ __launch_bounds__(64 * kWaves) void conv1x1_small_kernel(float* __restrict__ y, const float* __restrict__ w,
                                                                    const float* __restrict__ x, const float* __restrict__ bias, int N,
                                                                    int P, int Cout, int Cin, int act, float slope, float gain) {
  __shared__ float red[kWaves][MI * NI * 4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int co0 = blockIdx.y * 16 * MI, n0 = blockIdx.x * 16 * NI;
  const float* ap[MI];
  const float* bp[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) ap[mi] = w + (int64_t)min(co0 + 16 * mi + r, Cout - 1) * Cin + 4 * kq;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = min(n0 + 16 * ni + r, N - 1);
    const int b = n / P, p = n - b * P;
    bp[ni] = x + ((int64_t)b * Cin + 4 * kq) * P + p;
  }
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = Cin >> 4;                                   // 16-channel steps; wave wv takes t = wv, wv + 8, ...
  f32x4 a[MI], an[MI];
  float bq[NI][4], bn[NI][4];
  auto fetch = [&](int t, f32x4* A, float (*Bf)[4]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) A[mi] = *reinterpret_cast<const f32x4*>(ap[mi] + 16 * t);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) Bf[ni][e] = bp[ni][(int64_t)(16 * t + e) * P];
  };
  int t = wv;
  if (t < nt) fetch(t, a, bq);
  for (; t < nt; t += kWaves) {
    const int tn = t + kWaves < nt ? t + kWaves : t;         // uniform; the last step re-reads its own operands
    fetch(tn, an, bn);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][e], bq[ni][e], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = an[mi];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[ni][e] = bn[ni][e];
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wv][(mi * NI + ni) * 4 + i][lane] = acc[mi][ni][i];
  __syncthreads();
  // 64 MI NI 4 sums over the 8 waves; thread (wv, lane) takes registers q = wv, wv + 8, ...
  for (int q = wv; q < MI * NI * 4; q += kWaves) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) s += red[k][q][lane];
    const int i = q & 3, ni = (q >> 2) % NI, mi = (q >> 2) / NI;
    const int co = co0 + 16 * mi + 4 * kq + i, n = n0 + 16 * ni + r;
    if (co < Cout && n < N) {
      if (bias) s += bias[co];
      if (act) s = (s < 0.f ? s * slope : s) * gain;
      const int b = n / P, p = n - b * P;
      y[((int64_t)b * Cout + co) * P + p] = s;
    }
  }
}

}  // namespace

extern "C" int vsp_conv1x1_small_f32(float* y, const float* w, const float* x, const float* bias, int B, int Cout, int Cin, int P, int act,
                                     float slope, float gain, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && Cout >= 1 && Cin >= 16 && P >= 1, "conv1x1_small: bad dims");
  VSP_REQUIRE(Cin % 16 == 0, "conv1x1_small: the input channel count must be a multiple of 16 (got %d)", Cin);
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(y && w && x, "conv1x1_small: null pointer");
  VSP_REQUIRE(vsp::aligned16(w), "conv1x1_small: the weight must be 16-byte aligned (its rows are read as 16-byte fragments)");
  const int64_t N64 = (int64_t)B * P;
  VSP_REQUIRE(N64 < (1 << 30), "conv1x1_small: too many columns");
  const int N = (int)N64;
  hipStream_t st = vsp::as_stream(stream);
  // 32 x 32 tiles when that still gives the chip a workgroup per CU, else 16 x 32 (more, smaller workgroups)
  const int big = ((Cout + 31) / 32) * ((N + 31) / 32);
  if (big >= 200)
    conv1x1_small_kernel<2, 2><<<dim3((N + 31) / 32, (Cout + 31) / 32), 64 * kWaves, 0, st>>>(y, w, x, bias, N, P, Cout, Cin, act, slope, gain);
  else
    conv1x1_small_kernel<1, 2><<<dim3((N + 31) / 32, (Cout + 15) / 16), 64 * kWaves, 0, st>>>(y, w, x, bias, N, P, Cout, Cin, act, slope, gain);
  return vsp::check_launch("conv1x1_small");
}

// Strided batched small GEMM ("NT": both operands indexed [row][k]) on fp32 MFMA for gfx950.
//
// Used for every F.linear / torch.matmul of the path (EqualLinear style MLPs and modulations:
// models/RestoreNet.py:161-171; TACC_block / spatial_attention: models/CodeDiffuser.py:35-47,86-116).  These have
// 8..288 rows (M = batch or batch*18 tokens) and K = 512..8192: far too few rows for classic tiles, and a single wave
// walking K alone is a chain of dependent L2 round trips.  So: one workgroup (4 wave64s) owns a 16(m) x 32(n) tile and
// the four waves split K (wave w takes the 16-wide k-steps w, w+4, ...); operands stream straight from global/L2 into
// MFMA fragments (no LDS staging: nothing is reused across waves) with two k-steps of loads in flight; the four partial
// accumulators meet in LDS and wave 0 runs the epilogue.  Grids are (N/32) x (M/16) x Z workgroups: 576 for the
// [144 x 2048 x 512] projection of a TACC block, so all 256 CUs work even at M = 144.
//
// k-permutation trick (VEC path, k contiguous): lane (row = l&15, q = l>>4) loads ONE float4 at [row][k0+4q..4q+3];
// MFMA j (j=0..3) of the step takes component j of that float4 as its k-slot q, so slot q of MFMA j is k0+4q+j
// for A and B alike: the products pair up correctly and a 16-wide k-step costs one 16-byte load per operand
// block (64-byte segments per row instead of the 16-byte segments a scalar fragment load would touch).
#include "vsp_common.h"
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GemmK {
  const float* A;
  const float* Bm;
  float* C;
  int M, N, K;
  int64_t a_zs, a_ms, a_ks, b_zs, b_ns, b_ks, c_zs, c_ms;
  float alpha;
  const float* bias;
  float bias_scale;
  int act;
  float slope, gain;
  int64_t bias_zs;
};

constexpr int NBL = 2;  // 16-column blocks per workgroup tile
constexpr int KW = 4;   // waves splitting K

// MBK = 16-row blocks per workgroup.  1: the latency form above.  4 (round 3, M >= 1024 rows: the [7200 x 512] x [512 x 512] head GEMMs of
// the sampler's prepare step): the B fragments of a k-step are loaded once and multiply four A blocks -- with one block per workgroup
// every 16 x 32 tile streams both operands from L2 (690 MB for that GEMM: 84.7 us, L2-bound at 42 TFLOP/s).  The arithmetic per output
// element (k order, the 4-wave split, the LDS reduction order) is the same for every MBK: results do not depend on M.
template <bool VEC, int MBK>
__global__ __launch_bounds__(64 * KW) void gemm_nt_kernel(const GemmK p) {
  __shared__ float red[(KW - 1) * MBK * NBL * 4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16 * NBL;
  const int m0 = blockIdx.y * 16 * MBK;
  const int z = blockIdx.z;

  const float* arow[MBK];
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb) arow[mb] = p.A + z * p.a_zs + (int64_t)min(m0 + mb * 16 + lr, p.M - 1) * p.a_ms;
  const float* brow[NBL];
#pragma unroll
  for (int j = 0; j < NBL; ++j) brow[j] = p.Bm + z * p.b_zs + (int64_t)min(n0 + j * 16 + lr, p.N - 1) * p.b_ns;

  f32x4 acc[MBK][NBL];
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int j = 0; j < NBL; ++j) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kdone = 0;  // columns [0, kdone) are covered by the vector loop
  if (VEC) {
    kdone = p.K & ~15;
    auto step = [&](const float4 (&a)[MBK], const float4 (&b)[NBL]) {
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
        for (int j = 0; j < NBL; ++j) {
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].x, b[j].x, acc[mb][j], 0, 0, 0);
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].y, b[j].y, acc[mb][j], 0, 0, 0);
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].z, b[j].z, acc[mb][j], 0, 0, 0);
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].w, b[j].w, acc[mb][j], 0, 0, 0);
        }
    };
    int k0 = wave * 16;
    // two k-steps in flight: the loads of step s+1 are issued before the MFMAs of step s
    float4 a0[MBK], b0[NBL];
    if (k0 < kdone) {
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb) a0[mb] = *reinterpret_cast<const float4*>(arow[mb] + k0 + 4 * kq);
#pragma unroll
      for (int j = 0; j < NBL; ++j) b0[j] = *reinterpret_cast<const float4*>(brow[j] + k0 + 4 * kq);
    }
    for (; k0 < kdone; k0 += 16 * KW) {
      const int k1 = k0 + 16 * KW;
      float4 a1[MBK], b1[NBL];
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb) a1[mb] = a0[mb];
#pragma unroll
      for (int j = 0; j < NBL; ++j) b1[j] = b0[j];
      if (k1 < kdone) {
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) a1[mb] = *reinterpret_cast<const float4*>(arow[mb] + k1 + 4 * kq);
#pragma unroll
        for (int j = 0; j < NBL; ++j) b1[j] = *reinterpret_cast<const float4*>(brow[j] + k1 + 4 * kq);
      }
      step(a0, b0);
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb) a0[mb] = a1[mb];
#pragma unroll
      for (int j = 0; j < NBL; ++j) b0[j] = b1[j];
    }
  }
  // scalar k-steps of 4: the K % 16 tail of the vector path, or everything for arbitrary strides
  for (int k0 = kdone + wave * 4; k0 < p.K; k0 += 4 * KW) {
    const int k = k0 + kq;
    const bool ok = k < p.K;
    const int kc = ok ? k : 0;
    float b[NBL];
#pragma unroll
    for (int j = 0; j < NBL; ++j) b[j] = ok ? brow[j][kc * p.b_ks] : 0.f;
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb) {
      const float a = ok ? arow[mb][kc * p.a_ks] : 0.f;
#pragma unroll
      for (int j = 0; j < NBL; ++j) acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[mb][j], 0, 0, 0);
    }
  }

  // K-slice reduction through LDS
  if (wave > 0) {
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
      for (int j = 0; j < NBL; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((((wave - 1) * MBK + mb) * NBL + j) * 4 + r) * 64 + lane] = acc[mb][j][r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 1; w < KW; ++w)
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
      for (int j = 0; j < NBL; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mb][j][r] += red[((((w - 1) * MBK + mb) * NBL + j) * 4 + r) * 64 + lane];

  // D layout: lane holds column lr, rows kq*4 + r
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int j = 0; j < NBL; ++j) {
      const int n = n0 + j * 16 + lr;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[z * p.bias_zs + n] * p.bias_scale : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mb * 16 + kq * 4 + r;
        if (m >= p.M) continue;
        float v = acc[mb][j][r] * p.alpha + bv;
        if (p.act == 1) v = (v > 0.f ? v : v * p.slope) * p.gain;
        else if (p.act == 2) v = 1.f / (1.f + expf(-v));
        p.C[z * p.c_zs + m * p.c_ms + n] = v;
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Few-row form (round 3): C[m, n] for M <= 16 rows -- every EqualLinear of the path at batch 8 / 16: the style modulations of all
// modulated convolutions (8 x 2048 against 512 x 2048: a 4 MB weight per launch), the style MLP, final_linear.  The tiled kernel
// above puts (N / 64) x 1 workgroups on the chip and walks K in one wave-quartet each: 25 us for 4 MB (160 GB/s) -- 45 such launches
// per inference step.  Here ONE WAVE owns one output column: it streams its weight row as 16-byte loads (K / 256 per lane, all
// issued before the first use), multiplies it against the M rows of x (the same addresses in every wave: L1 / L2 hits), reduces
// with DPP adds; N / 4 workgroups of four waves.
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_mov_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x141>(v);  // row_half_mirror
  v += dpp_mov_f<0x140>(v);  // row_mirror
  const int iv = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}

template <int MR, int KI>   // MR rows at most, KI = K / 256 16-byte pieces per lane (0: run-time trip count)
__global__ __launch_bounds__(256) void gemv_rows_kernel(const GemmK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= p.N) return;
  const float* wrow = p.Bm + (int64_t)n * p.b_ns + lane * 4;
  const float* arow = p.A + lane * 4;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const int ki = KI ? KI : p.K / 256;
#pragma unroll KI ? KI : 4
  for (int i = 0; i < ki; ++i) {
    const float4 w = *reinterpret_cast<const float4*>(wrow + i * 256);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float4 a = *reinterpret_cast<const float4*>(arow + (int64_t)(m < p.M ? m : 0) * p.a_ms + i * 256);
      acc[m] = fmaf(a.x, w.x, fmaf(a.y, w.y, fmaf(a.z, w.z, fmaf(a.w, w.w, acc[m]))));
    }
  }
  const float bv = p.bias ? p.bias[n] * p.bias_scale : 0.f;
  float out = 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const float v = wave_sum_f(acc[m]);
    if (lane == m) out = v;
  }
  if (lane < p.M) {
    float v = out * p.alpha + bv;
    if (p.act == 1) v = (v > 0.f ? v : v * p.slope) * p.gain;
    else if (p.act == 2) v = 1.f / (1.f + expf(-v));
    p.C[(int64_t)lane * p.c_ms + n] = v;
  }
}

template <int MR>
static void launch_gemv(const GemmK& q, hipStream_t st) {
  const dim3 grid((unsigned)((q.N + 3) / 4));
  switch (q.K / 256) {
    case 2: gemv_rows_kernel<MR, 2><<<grid, 256, 0, st>>>(q); break;
    case 4: gemv_rows_kernel<MR, 4><<<grid, 256, 0, st>>>(q); break;
    case 8: gemv_rows_kernel<MR, 8><<<grid, 256, 0, st>>>(q); break;
    default: gemv_rows_kernel<MR, 0><<<grid, 256, 0, st>>>(q); break;
  }
}

}  // namespace

extern "C" int vsp_gemm_f32(const vsp_gemm_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "gemm: null params");
  const vsp_gemm_params& p = *pp;
  VSP_REQUIRE(p.Z >= 0 && p.M >= 0 && p.N >= 0 && p.K >= 0, "gemm: negative dimension");
  if (p.Z == 0 || p.M == 0 || p.N == 0) return VSP_OK;
  VSP_REQUIRE(p.A && p.Bm && p.C, "gemm: null tensor pointer");
  VSP_REQUIRE(p.act >= 0 && p.act <= 2, "gemm: unknown activation %d", p.act);
  VSP_REQUIRE(p.Z <= 65535, "gemm: batch too large");
  GemmK q{p.A, p.Bm, p.C, p.M, p.N, p.K, p.a_zs, p.a_ms, p.a_ks, p.b_zs, p.b_ns, p.b_ks, p.c_zs, p.c_ms,
          p.alpha, p.bias, p.bias_scale, p.act, p.slope, p.gain, p.bias_zs};
  const bool vec = p.a_ks == 1 && p.b_ks == 1 && p.a_ms % 4 == 0 && p.b_ns % 4 == 0 && p.a_zs % 4 == 0 &&
                   p.b_zs % 4 == 0 && vsp::aligned16(p.A) && vsp::aligned16(p.Bm);
  if (vec && p.Z == 1 && p.M <= 16 && p.K >= 512 && p.K % 256 == 0 && p.b_ns >= p.K && p.a_ms >= p.K) {   // few rows against a deep K
    static const bool off = vsp::tune_env("VSP_GEMV_OFF") != nullptr;   // (A/B runs)
    if (!off) {
      if (p.M <= 8) launch_gemv<8>(q, vsp::as_stream(stream)); else launch_gemv<16>(q, vsp::as_stream(stream));
      return vsp::check_launch("gemm");
    }
  }
  const bool wide = vec && p.M >= 1024;   // several row blocks per workgroup share the B fragments (same arithmetic per element)
  static const int wide_mbk = [] {   // tuning switch; only the instantiated forms (the grid below is sized for the SAME value)
    const int v = vsp::tune_env("VSP_GEMM_MBK") ? atoi(vsp::tune_env("VSP_GEMM_MBK")) : 4;
    return (v == 2 || v == 4 || v == 8) ? v : 4;
  }();
  const int mrows = wide ? 16 * wide_mbk : 16;
  dim3 grid((unsigned)((p.N + 16 * NBL - 1) / (16 * NBL)), (unsigned)((p.M + mrows - 1) / mrows), (unsigned)p.Z);
  VSP_REQUIRE(grid.y <= 65535, "gemm: M too large");
  if (wide && wide_mbk == 8)
    gemm_nt_kernel<true, 8><<<grid, 64 * KW, 0, vsp::as_stream(stream)>>>(q);
  else if (wide && wide_mbk == 2)
    gemm_nt_kernel<true, 2><<<grid, 64 * KW, 0, vsp::as_stream(stream)>>>(q);
  else if (wide)
    gemm_nt_kernel<true, 4><<<grid, 64 * KW, 0, vsp::as_stream(stream)>>>(q);
  else if (vec)
    gemm_nt_kernel<true, 1><<<grid, 64 * KW, 0, vsp::as_stream(stream)>>>(q);
  else
    gemm_nt_kernel<false, 1><<<grid, 64 * KW, 0, vsp::as_stream(stream)>>>(q);
  return vsp::check_launch("gemm");
}

// Shared device code of the TACC kernels (tacc.hip: per-launch entry points; tacc_chain.hip: the whole sampler chain).
#pragma once
#include "vsp_common.h"

namespace vsptacc {

constexpr int NTOK = 18;
constexpr int D = 512;

// ---------------------------------------------------------------------------------------------------------------------
// Channel attention on fp32 MFMA (same contract as tacc_chan_attn_kernel; this is the one the sampler uses).
//   GEMM 1  logit[r][c] = sum_tok k2[tok][r] * q2[tok][c]      M = 512 rows (wave w owns rows 128w..128w+127 = 8 m-tiles),
//                                                              N = 32 columns (2 n-tiles), K = 18 tokens padded to 20
//   softmax over r: per-lane partial max / sum over the 32 values a lane holds per column, xor-shuffles across the 4 row
//           groups of a wave, 4 x 32 floats of LDS across waves
//   GEMM 2  t[tok][c] = sum_r v2[tok][r] * e[r][c]             K = rows: the e values never move -- register j of m-tile mt
//           in the D layout IS the B fragment of the k-step whose 4 slots are rows 16 mt + 4 q + j (q = lane >> 4), so only
//           the A fragment (v2) is read from LDS with the matching row; each wave reduces over its own 128 rows and the four
//           partial t tiles meet in LDS.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CA_KP = D + 16;  // k2 pitch: k-slot groups land on different bank halves
constexpr int CA_VP = D + 1;   // v2 pitch: 16 token rows hit 16 different banks
constexpr int CA_NW = 8;       // waves per workgroup (each owns 512 / CA_NW rows of the attention matrix)
constexpr size_t CA_LDS_FLOATS = 20 * CA_KP + NTOK * CA_VP + 2 * CA_NW * 32 + (CA_NW - 1) * 4 * 4 * 64;

// blockDim.x must be 64 * CA_NW.
__device__ __forceinline__ void chan_attn_mfma_body(float* __restrict__ tout, const float* __restrict__ P, int ldp,
                                                    int q2_off, int v2_off, const float* __restrict__ ek,
                                                    const float* __restrict__ wk, int wk_stride, float tf, float scale,
                                                    const int cb, const int b) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NW = CA_NW, NT = 64 * NW;
  constexpr int MT = D / 16 / NW;  // 16-row tiles per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* k2s = smem;                    // [20][CA_KP], token rows 18, 19 are zero
  float* v2s = k2s + 20 * CA_KP;        // [18][CA_VP]
  float* redm = v2s + NTOK * CA_VP;     // [NW][32]
  float* reds = redm + NW * 32;         // [NW][32]
  float* tpart = reds + NW * 32;        // [NW-1][4 tiles][4][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, kq = lane >> 4;

  float qb[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int tok = 4 * s + kq;
      qb[s][nt] = tok < NTOK ? P[((int64_t)b * NTOK + tok) * ldp + q2_off + cb * 32 + nt * 16 + lr] * scale : 0.f;
    }
  if (wk_stride == 1) {  // contiguous condition column (the sampler passes it that way): 16-byte staging loads
    // every global load of a thread is issued before the first LDS write: one L2 round trip for the whole staging
    constexpr int TR = NT / 128;               // token rows covered per pass
    constexpr int NIT = (20 + TR - 1) / TR;
    const int c4 = tid & 127, tr = tid >> 7;   // float4 column; thread covers tokens tr, tr + TR, ...
    const float4 w = *reinterpret_cast<const float4*>(wk + c4 * 4);
    float4 kreg[NIT], vreg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int tok = tr + TR * it;
      const int tc = tok < NTOK ? tok : NTOK - 1;
      kreg[it] = *reinterpret_cast<const float4*>(ek + ((int64_t)b * NTOK + tc) * D + c4 * 4);
      vreg[it] = *reinterpret_cast<const float4*>(P + ((int64_t)b * NTOK + tc) * ldp + v2_off + c4 * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int tok = tr + TR * it;
      if (tok >= 20) continue;
      float4 v = kreg[it];
      v.x = fmaf(tf, w.x, v.x); v.y = fmaf(tf, w.y, v.y); v.z = fmaf(tf, w.z, v.z); v.w = fmaf(tf, w.w, v.w);
      if (tok >= NTOK) v = make_float4(0.f, 0.f, 0.f, 0.f);  // pad tokens 18, 19
      *reinterpret_cast<float4*>(k2s + tok * CA_KP + c4 * 4) = v;
      if (tok < NTOK) {
        float* d = v2s + tok * CA_VP + c4 * 4;
        d[0] = vreg[it].x; d[1] = vreg[it].y; d[2] = vreg[it].z; d[3] = vreg[it].w;
      }
    }
  } else {
    for (int i = tid; i < 20 * D; i += NT) {
      const int tok = i / D, r = i - tok * D;
      float v = 0.f;
      if (tok < NTOK) v = ek[((int64_t)b * NTOK + tok) * D + r] + tf * wk[(int64_t)r * wk_stride];
      k2s[tok * CA_KP + r] = v;
    }
    for (int i = tid; i < NTOK * D; i += NT) {
      const int tok = i / D, r = i - tok * D;
      v2s[tok * CA_VP + r] = P[((int64_t)b * NTOK + tok) * ldp + v2_off + r];
    }
  }
  __syncthreads();

  f32x4 L[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    L[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    L[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float a = k2s[(4 * s + kq) * CA_KP + (wave * MT + mt) * 16 + lr];
      L[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qb[s][0], L[mt][0], 0, 0, 0);
      L[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qb[s][1], L[mt][1], 0, 0, 0);
    }
  }
  // column max over the 512 rows (lane: column lr of n-tile nt, rows 16 mt + 4 kq + j of this wave's m-tiles)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float m = L[0][nt][0];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) m = fmaxf(m, L[mt][nt][j]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (kq == 0) redm[wave * 32 + nt * 16 + lr] = m;
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float m = redm[nt * 16 + lr];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, redm[w * 32 + nt * 16 + lr]);
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = expf(L[mt][nt][j] - m);
        L[mt][nt][j] = e;
        s += e;
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kq == 0) reds[wave * 32 + nt * 16 + lr] = s;
  }
  // t partial over this wave's rows: the e values never move -- register j of m-tile mt in the D layout IS the B fragment
  // of the k-step whose 4 slots are rows 16 mt + 4 kq + j
  f32x4 T[2][2];
#pragma unroll
  for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) T[mt2][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tok1 = 16 + lr;
  const bool ok1 = tok1 < NTOK;
  const float* va0 = v2s + lr * CA_VP + wave * (MT * 16) + kq * 4;
  const float* va1 = v2s + (ok1 ? tok1 : NTOK - 1) * CA_VP + wave * (MT * 16) + kq * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a0 = va0[mt * 16 + j];
      const float a1 = ok1 ? va1[mt * 16 + j] : 0.f;
      T[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, L[mt][0][j], T[0][0], 0, 0, 0);
      T[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, L[mt][1][j], T[0][1], 0, 0, 0);
      T[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, L[mt][0][j], T[1][0], 0, 0, 0);
      T[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, L[mt][1][j], T[1][1], 0, 0, 0);
    }
  if (wave > 0) {
#pragma unroll
    for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) tpart[(((wave - 1) * 4 + mt2 * 2 + nt) * 4 + j) * 64 + lane] = T[mt2][nt][j];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float den = reds[nt * 16 + lr];
#pragma unroll
    for (int w = 1; w < NW; ++w) den += reds[w * 32 + nt * 16 + lr];
    const float inv = 1.f / den;
#pragma unroll
    for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = T[mt2][nt][j];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += tpart[(((w - 1) * 4 + mt2 * 2 + nt) * 4 + j) * 64 + lane];
        const int tok = mt2 * 16 + kq * 4 + j;
        if (tok < NTOK) tout[((int64_t)b * NTOK + tok) * D + cb * 32 + nt * 16 + lr] = v * inv;
      }
  }
}

}  // namespace vsptacc

// The whole Code_diffuser sampler chain (DDPM posterior-mean chain or deterministic DDIM) behind ONE C call, for gfx950.
//
// Reference: ldm/ddpm.py:400-429 (p_sample_loop over My_DDPM.p_sample) around models/CodeDiffuser.py:86-140 (four
// TACC_blocks per denoiser call).  At batch 8 a TACC block is ~0.3 GFLOP on 144 token rows: the chain is 4*T dependent
// block evaluations whose cost is launch count and dependent memory round trips, not arithmetic.  Hence per block
// THREE latency-shaped launches (instead of ~56 ATen ops in the reference), all enqueued from C in one go:
//
//   tacc_proj   P = pixelnorm(y) @ [Wk; Wv; Wq2; Wv2]^T.  One workgroup per (sample, 32 output columns), 8 waves
//               splitting K.  Every operand load of a wave is issued up front (one L2 round trip); the PixelNorm over the
//               18 tokens is computed from the A fragments the wave holds anyway (DPP row reductions), so the previous
//               block never has to produce a normalised copy and needs no cross-row step.
//   tacc_attn   heterogeneous grid: workgroups [0, 16B) run the 512x512 channel attention on MFMA (tacc_kernels.h),
//               workgroups after that run the 18x18 token attention, one wave per token row (8 rows per workgroup), Q/V rows straight from L2.
//   tacc_post   one wave per token row: LN(t), LN(h + LN(t)), FiLM, and after the last block the sampler update
//               x' = c1[k] f(x) + c2[k] x, in place.
#include "tacc_kernels.h"
#include <cstdlib>

namespace {

using vsptacc::D;
using vsptacc::NTOK;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row: xor 1, xor 2 (quad_perm), then the two mirrors
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
// sum over the wave, uniform result (no LDS round trips: 4 DPP adds + 4 lane reads)
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const int iv = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}

// ---------------------------------------------------------------------------------------------------------------------
// P[b*18 + i][n] = sum_c y[b,i,c] * r[b,c] * W[n][c],  r[b,c] = rsqrt(mean_i y[b,i,c]^2 + 1e-8)   (PixelNorm over tokens,
// models/CodeDiffuser.py:11-12).  grid (N/32) * B, 64*KW threads.  k-permutation fragments as in gemm_small.hip: lane
// (row lr, slot kq) holds the float4 [row][k0 + 4 kq .. + 3]; MFMA j of a k-step uses component j on both sides.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PJ_KW = 8;                  // waves splitting K
constexpr int PJ_NS = D / 16 / PJ_KW;     // 16-wide k-steps per wave
constexpr int PJ_NJ = 4;                  // 16-column blocks per workgroup: the PixelNorm VALU work is redone per workgroup
                                          // of a sample, so wider tiles (fewer workgroups per sample) cut it

__global__ __launch_bounds__(64 * PJ_KW) void tacc_proj_kernel(float* __restrict__ P, const float* __restrict__ y,
                                                               const float* __restrict__ W, int ldp, const float* __restrict__ Wf) {
  constexpr int NJ = PJ_NJ;
  extern __shared__ __attribute__((aligned(16))) float pj_smem[];
  float* red = pj_smem;                            // MFMA partials of every wave: [wave][j][r][lane]
  float* red2 = pj_smem + PJ_KW * NJ * 4 * 64;     // rows 16, 17: [wave][kq][row][j][lr]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  // XCD-aware mapping: workgroups go round-robin over the 8 XCDs by linear id; XCD x always gets the SAME eighth of the
  // output columns (for every sample and every step), so each XCD only ever touches its 0.5 MB slice of a block's weights.
  constexpr int TPX = 4 * D / (16 * NJ) / 8;  // column tiles per XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n0 = (xcd * TPX + slot % TPX) * (16 * NJ), b = slot / TPX;
  const float* y0 = y + ((int64_t)b * NTOK + lr) * D + 4 * kq;
  const float* y16 = y + ((int64_t)b * NTOK + 16) * D + 4 * kq;  // rows 16, 17: same address in the 16 lanes of a k slot
  const float* w0 = W + (int64_t)(n0 + lr) * D + 4 * kq;

  float4 a0[PJ_NS], r16[PJ_NS], r17[PJ_NS], bw[PJ_NS][NJ];
#pragma unroll
  for (int s = 0; s < PJ_NS; ++s) {
    const int k0 = (wave + PJ_KW * s) * 16;
    a0[s] = *reinterpret_cast<const float4*>(y0 + k0);
    r16[s] = *reinterpret_cast<const float4*>(y16 + k0);
    r17[s] = *reinterpret_cast<const float4*>(y16 + D + k0);
#pragma unroll
    for (int j = 0; j < NJ; ++j)   // Wf: the matrix in fragment order [n / 16][k / 16][lane][4] (one wave instruction = 1 KiB of consecutive memory)
      bw[s][j] = Wf ? reinterpret_cast<const float4*>(Wf)[((int64_t)((n0 >> 4) + j) * (D / 16) + (k0 >> 4)) * 64 + lane]
                    : *reinterpret_cast<const float4*>(w0 + (int64_t)j * 16 * D + k0);
  }
  __builtin_amdgcn_sched_barrier(0);  // keep every load in flight together (the scheduler would sink half of them)
  // Rows 0..15 of the sample run on MFMA; rows 16 and 17 would cost a second, 7/8 empty MFMA row block (half of all matrix
  // time), so they are plain FMAs against the B fragments the lane already holds: e[row][j] = partial of
  // (row, column n0 + 16 j + lr) over this lane's four k -- summed over k slots and waves through LDS at the end.
  f32x4 acc[NJ];
  float e[2][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    e[0][j] = e[1][j] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < PJ_NS; ++s) {
    float4 a = a0[s], p = r16[s], q = r17[s];
    const float rx = rsqrtf((row16_sum(a.x * a.x) + fmaf(p.x, p.x, q.x * q.x)) * (1.f / NTOK) + 1e-8f);
    const float ry = rsqrtf((row16_sum(a.y * a.y) + fmaf(p.y, p.y, q.y * q.y)) * (1.f / NTOK) + 1e-8f);
    const float rz = rsqrtf((row16_sum(a.z * a.z) + fmaf(p.z, p.z, q.z * q.z)) * (1.f / NTOK) + 1e-8f);
    const float rw = rsqrtf((row16_sum(a.w * a.w) + fmaf(p.w, p.w, q.w * q.w)) * (1.f / NTOK) + 1e-8f);
    a.x *= rx; a.y *= ry; a.z *= rz; a.w *= rw;
    p.x *= rx; p.y *= ry; p.z *= rz; p.w *= rw;
    q.x *= rx; q.y *= ry; q.z *= rz; q.w *= rw;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 w = bw[s][j];
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[j], 0, 0, 0);
      e[0][j] = fmaf(p.x, w.x, fmaf(p.y, w.y, fmaf(p.z, w.z, fmaf(p.w, w.w, e[0][j]))));
      e[1][j] = fmaf(q.x, w.x, fmaf(q.y, w.y, fmaf(q.z, w.z, fmaf(q.w, w.w, e[1][j]))));
    }
  }
  // K-slice reduction through LDS, spread over all waves: wave w finishes the (j, r) accumulator registers w, w + KW, ...
  // (D layout: lane holds column lr, row kq*4 + r) and the first two waves the rows 16 / 17.
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * NJ + j) * 4 + r) * 64 + lane] = acc[j][r];
    red2[(((wave * 4 + kq) * 2 + 0) * NJ + j) * 16 + lr] = e[0][j];
    red2[(((wave * 4 + kq) * 2 + 1) * NJ + j) * 16 + lr] = e[1][j];
  }
  __syncthreads();
  for (int jr = wave; jr < NJ * 4; jr += PJ_KW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < PJ_KW; ++w) v += red[(w * NJ * 4 + jr) * 64 + lane];
    P[((int64_t)b * NTOK + kq * 4 + (jr & 3)) * ldp + n0 + (jr >> 2) * 16 + lr] = v;
  }
  if (wave < 2) {  // row 16 + wave: lane -> column (kq, lr) of the first four column blocks, then the next four, ...
    for (int j = kq; j < NJ; j += 4) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < PJ_KW * 4; ++i) v += red2[((i * 2 + wave) * NJ + j) * 16 + lr];
      P[((int64_t)b * NTOK + 16 + wave) * ldp + n0 + j * 16 + lr] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Channel attention (workgroups [0, 16 B)) and token attention (the rest, one token row per wave) in one launch.
// Token attention, wave = row (b, i), lane owns channels 4 lane + 256 u + {0..3}:
//   s_j = K_i . (eQ_j + tf wq) / sqrt(18);  p = softmax_j(s);  h_i = sum_j p_j V_j
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * vsptacc::CA_NW) void tacc_attn_kernel(float* __restrict__ tout, float* __restrict__ hout,
                                                        const float* __restrict__ P, int ldp, const float* __restrict__ ek,
                                                        const float* __restrict__ wk, const float* __restrict__ eQ,
                                                        const float* __restrict__ wq, float tf, int B) {
  const int nca = 16 * B;
  if ((int)blockIdx.x < nca) {
    vsptacc::chan_attn_mfma_body(tout, P, ldp, 2 * D, 3 * D, ek, wk, 1, tf, 0.044194173824159216f /* 1/sqrt(512) */,
                                 blockIdx.x & 15, blockIdx.x >> 4);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int row = ((int)blockIdx.x - nca) * vsptacc::CA_NW + (threadIdx.x >> 6);
  if (row >= B * NTOK) return;
  const int b = row / NTOK;
  const float sscale = 0.23570226039551584f;  // 1/sqrt(18)
  float4 kv[2], wv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    kv[u] = *reinterpret_cast<const float4*>(P + (int64_t)row * ldp + 4 * lane + 256 * u);
    wv[u] = *reinterpret_cast<const float4*>(wq + 4 * lane + 256 * u);
    wv[u].x *= tf; wv[u].y *= tf; wv[u].z *= tf; wv[u].w *= tf;
  }
  float s[NTOK];
#pragma unroll
  for (int j = 0; j < NTOK; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 q = *reinterpret_cast<const float4*>(eQ + ((int64_t)b * NTOK + j) * D + 4 * lane + 256 * u);
      acc = fmaf(kv[u].x, q.x + wv[u].x, acc);
      acc = fmaf(kv[u].y, q.y + wv[u].y, acc);
      acc = fmaf(kv[u].z, q.z + wv[u].z, acc);
      acc = fmaf(kv[u].w, q.w + wv[u].w, acc);
    }
    s[j] = acc;
  }
#pragma unroll
  for (int j = 0; j < NTOK; ++j) s[j] = wave_sum(s[j]) * sscale;
  float m = s[0];
#pragma unroll
  for (int j = 1; j < NTOK; ++j) m = fmaxf(m, s[j]);
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < NTOK; ++j) {
    s[j] = expf(s[j] - m);
    den += s[j];
  }
  const float rden = 1.f / den;
  float4 h[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
  for (int j = 0; j < NTOK; ++j) {
    const float pj = s[j] * rden;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 v = *reinterpret_cast<const float4*>(P + ((int64_t)b * NTOK + j) * ldp + D + 4 * lane + 256 * u);
      h[u].x = fmaf(pj, v.x, h[u].x);
      h[u].y = fmaf(pj, v.y, h[u].y);
      h[u].z = fmaf(pj, v.z, h[u].z);
      h[u].w = fmaf(pj, v.w, h[u].w);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(hout + (int64_t)row * D + 4 * lane + 256 * u) = h[u];
}

// ---------------------------------------------------------------------------------------------------------------------
// y = LN(h + LN(t)) * (1 + gamma) + beta;  with xold: y = c1[idx] * y + c2[idx] * xold  (in place allowed: yout == xold).
// One wave per token row, 4 rows per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tacc_post_kernel(float* __restrict__ yout, const float* __restrict__ h,
                                                        const float* __restrict__ t, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* xold,
                                                        const float* __restrict__ c1, const float* __restrict__ c2, int idx,
                                                        int rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t o = (int64_t)row * D + 4 * lane;
  float4 tv[2], hv[2], gv[2], bv[2], xv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    tv[u] = *reinterpret_cast<const float4*>(t + o + 256 * u);
    hv[u] = *reinterpret_cast<const float4*>(h + o + 256 * u);
    gv[u] = *reinterpret_cast<const float4*>(gamma + o + 256 * u);
    bv[u] = *reinterpret_cast<const float4*>(beta + o + 256 * u);
    xv[u] = xold ? *reinterpret_cast<const float4*>(xold + o + 256 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float a1 = xold ? c1[idx] : 1.f, a2 = xold ? c2[idx] : 0.f;
  auto stats = [&](const float4 (&v)[2], float& mean, float& inv) {
    float sm = (v[0].x + v[0].y) + (v[0].z + v[0].w) + (v[1].x + v[1].y) + (v[1].z + v[1].w);
    mean = wave_sum(sm) * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float d0 = v[u].x - mean, d1 = v[u].y - mean, d2 = v[u].z - mean, d3 = v[u].w - mean;
      var = fmaf(d0, d0, var); var = fmaf(d1, d1, var); var = fmaf(d2, d2, var); var = fmaf(d3, d3, var);
    }
    inv = rsqrtf(wave_sum(var) * (1.f / D) + eps);
  };
  float mean, inv;
  stats(tv, mean, inv);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    hv[u].x += (tv[u].x - mean) * inv; hv[u].y += (tv[u].y - mean) * inv;
    hv[u].z += (tv[u].z - mean) * inv; hv[u].w += (tv[u].w - mean) * inv;
  }
  stats(hv, mean, inv);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float4 yv;
    yv.x = (hv[u].x - mean) * inv * (1.f + gv[u].x) + bv[u].x;
    yv.y = (hv[u].y - mean) * inv * (1.f + gv[u].y) + bv[u].y;
    yv.z = (hv[u].z - mean) * inv * (1.f + gv[u].z) + bv[u].z;
    yv.w = (hv[u].w - mean) * inv * (1.f + gv[u].w) + bv[u].w;
    if (xold) {
      yv.x = a1 * yv.x + a2 * xv[u].x; yv.y = a1 * yv.y + a2 * xv[u].y;
      yv.z = a1 * yv.z + a2 * xv[u].z; yv.w = a1 * yv.w + a2 * xv[u].w;
    }
    *reinterpret_cast<float4*>(yout + o + 256 * u) = yv;
  }
}

}  // namespace

extern "C" {

size_t vsp_tacc_chain_work_floats(int B) {
  if (B <= 0) return 0;
  const size_t M = (size_t)B * NTOK;
  return M * 4 * D /* P */ + M * D /* t */ + M * D /* h */ + 2 * M * D /* y ping-pong */;
}

int vsp_tacc_chain_f32(const vsp_tacc_chain_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "tacc_chain: null params");
  const vsp_tacc_chain_params& p = *pp;
  VSP_REQUIRE(p.n_tok == NTOK && p.dim == D, "tacc_chain: built for 18 tokens x 512 channels (got %d x %d)", p.n_tok, p.dim);
  VSP_REQUIRE(p.B >= 0 && p.n_blocks >= 0 && p.n_steps >= 0, "tacc_chain: negative size");
  if (p.B == 0 || p.n_blocks == 0 || p.n_steps == 0) return VSP_OK;
  VSP_REQUIRE(p.B <= 4095, "tacc_chain: batch too large for one grid");
  VSP_REQUIRE(p.blocks && p.x && p.work && p.step, "tacc_chain: null pointer");
  VSP_REQUIRE(p.work_floats >= vsp_tacc_chain_work_floats(p.B), "tacc_chain: work buffer too small (%zu < %zu floats)",
              (size_t)p.work_floats, vsp_tacc_chain_work_floats(p.B));
  VSP_REQUIRE(p.t_div > 0.f, "tacc_chain: t_div must be positive");
  VSP_REQUIRE(!p.c1 == !p.c2, "tacc_chain: c1 and c2 come together");
  VSP_REQUIRE(vsp::aligned16(p.x) && vsp::aligned16(p.work), "tacc_chain: x and work must be 16-byte aligned");
  for (int i = 0; i < p.n_blocks; ++i) {
    const vsp_tacc_block& k = p.blocks[i];
    VSP_REQUIRE(k.wcat && k.eQ && k.ek && k.wq && k.wk && k.gamma && k.beta, "tacc_chain: block %d has a null pointer", i);
    VSP_REQUIRE(vsp::aligned16(k.wcat) && vsp::aligned16(k.eQ) && vsp::aligned16(k.ek) && vsp::aligned16(k.wq) &&
                    vsp::aligned16(k.wk) && vsp::aligned16(k.gamma) && vsp::aligned16(k.beta),
                "tacc_chain: block %d operands must be 16-byte aligned", i);
  }
  static vsp::LdsAttrOnce attr_a, attr_p;   // per device
  const size_t lds = vsptacc::CA_LDS_FLOATS * sizeof(float);
  constexpr size_t pj_lds = (size_t)(PJ_KW * PJ_NJ * 4 * 64 + PJ_KW * 4 * 2 * PJ_NJ * 16) * sizeof(float);
  if (int rc = attr_a.ensure(reinterpret_cast<const void*>(tacc_attn_kernel), 150 * 1024, "tacc_chain")) return rc;
  if (int rc = attr_p.ensure(reinterpret_cast<const void*>(tacc_proj_kernel), (int)pj_lds, "tacc_chain")) return rc;
  hipStream_t st = vsp::as_stream(stream);
  const int M = p.B * NTOK;
  float* P = p.work;
  float* tb = P + (size_t)M * 4 * D;
  float* hb = tb + (size_t)M * D;
  float* yb[2] = {hb + (size_t)M * D, hb + (size_t)2 * M * D};
  const int row_blocks = (M + 3) / 4;
  for (int s = 0; s < p.n_steps; ++s) {
    const int step = p.step[s];
    VSP_REQUIRE(step >= 0 && step < p.head_steps, "tacc_chain: step %d outside the prepared heads [0, %d)", step, p.head_steps);
    const float tf = (float)step / p.t_div;
    const int cidx = p.coef_idx ? p.coef_idx[s] : step;
    const float* cur = p.x;
    for (int bi = 0; bi < p.n_blocks; ++bi) {
      const vsp_tacc_block& k = p.blocks[bi];
      const bool last = bi == p.n_blocks - 1;
      tacc_proj_kernel<<<(4 * D / (16 * PJ_NJ)) * p.B, 64 * PJ_KW, pj_lds, st>>>(P, cur, k.wcat, 4 * D, k.wcat_frag);
      tacc_attn_kernel<<<16 * p.B + (M + vsptacc::CA_NW - 1) / vsptacc::CA_NW, 64 * vsptacc::CA_NW, lds, st>>>(tb, hb, P, 4 * D, k.ek, k.wk, k.eQ, k.wq, tf, p.B);
      float* out = last ? p.x : yb[bi & 1];
      const size_t hoff = (size_t)step * M * D;
      tacc_post_kernel<<<row_blocks, 256, 0, st>>>(out, hb, tb, k.gamma + hoff, k.beta + hoff,
                                                  (last && p.c1) ? p.x : nullptr, p.c1, p.c2, cidx, M, 1e-5f);
      cur = out;
    }
  }
  return vsp::check_launch("tacc_chain");
}

}  // extern "C"

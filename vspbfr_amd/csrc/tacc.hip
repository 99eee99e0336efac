// Fused kernels of one TACC_block step of Code_diffuser (reference models/CodeDiffuser.py:86-116, :35-47) for gfx950.
//
// Per block and DDPM step the x-dependent work is (B samples, 18 tokens, D = 512 channels):
//   P = pixelnorm(x) @ [Wk; Wv; Wq2; Wv2]^T                    one [B*18, 4D] small MFMA GEMM (gemm_small.hip)
//   score = softmax(K Q^T / sqrt(18))          (18 x 18)        tacc_scores_kernel      (one workgroup per sample)
//   A = softmax_dim1(k2^T q2 / sqrt(D))        (D x D)   \      tacc_chan_attn_kernel   (A never leaves LDS: a 512 x 32
//   t = v2 A                                   (18 x D)  /                               column slab per workgroup)
//   h = LN(score V + LN(t)); y = h (1+gamma) + beta; [x' = c1 y + c2 x]; pixelnorm(y or x')   tacc_tail_kernel
// The condition c = [embd, t/T] enters through Linear(D+1 -> D) layers; their D-wide part is step-independent (computed once
// per batch by the host), so Q and k2 are rebuilt on the fly as  e + (t/T) * w_lastcol  and never stored per step.
// All four kernels are VALU/LDS kernels: per sample the block costs ~60 MFLOP, the problem is latency and launch count
// (4 launches per block and step instead of ~56 in the reference), not arithmetic.
#include "tacc_kernels.h"
#include <cstdlib>

namespace {

using vsptacc::NTOK;
using vsptacc::D;
using vsptacc::CA_KP;
using vsptacc::CA_VP;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// score[b][i][j] = softmax_j( sum_d K[b,i,d] * (eQ[b,j,d] + tf*wq[d]) * scale ).  grid = B, 320 threads (5 waves):
// wave w owns rows i = w, w+5, ...; the K row sits in registers (8 per lane), Q rows stream from L2.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(320) void tacc_scores_kernel(float* __restrict__ score, const float* __restrict__ P, int ldp,
                                                           int k_off, const float* __restrict__ eQ,
                                                           const float* __restrict__ wq, int wq_stride, float tf,
                                                           float scale) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wcol[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) wcol[u] = wq[(int64_t)(lane + 64 * u) * wq_stride] * tf;
  for (int i = wave; i < NTOK; i += 5) {
    const float* kr = P + ((int64_t)b * NTOK + i) * ldp + k_off;
    float kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kv[u] = kr[lane + 64 * u];
    float s[NTOK];
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
      const float* qr = eQ + ((int64_t)b * NTOK + j) * D;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(kv[u], qr[lane + 64 * u] + wcol[u], acc);
      s[j] = wsum(acc) * scale;
    }
    float m = s[0];
#pragma unroll
    for (int j = 1; j < NTOK; ++j) m = fmaxf(m, s[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
      s[j] = expf(s[j] - m);
      sum += s[j];
    }
    if (lane < NTOK) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < NTOK; ++j) v = (lane == j) ? s[j] : v;
      score[((int64_t)b * NTOK + i) * NTOK + lane] = v / sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Channel attention.  grid = (D / 32, B), 256 threads.  Workgroup (cb, b) owns columns c in [32 cb, 32 cb + 32):
//   logit[r][c] = sum_tok k2[tok][r] * q2[tok][c] * scale,  k2 = ek[b] + tf * wk           (r = 0..D-1)
//   e = exp(logit - max_r), den[c] = sum_r e;  t[tok][c] = (sum_r v2[tok][r] * e[r][c]) / den[c]
// LDS: k2/v2 staging (18 x 512 each), the 512 x 32 slab of e (pitch 33), reduction scratch.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CA_COLS = 32;
constexpr int CA_PITCH = CA_COLS + 1;

__global__ __launch_bounds__(256) void tacc_chan_attn_kernel(float* __restrict__ tout, const float* __restrict__ P, int ldp,
                                                              int q2_off, int v2_off, const float* __restrict__ ek,
                                                              const float* __restrict__ wk, int wk_stride, float tf,
                                                              float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* k2s = smem;                    // [18][512]
  float* v2s = k2s + NTOK * D;          // [18][512]
  float* es = v2s + NTOK * D;           // [512][33]
  float* red = es + D * CA_PITCH;       // [8][32]
  const int cb = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const int c = tid & 31, rg = tid >> 5;  // column, row group (8 groups of 64 rows)

  for (int i = tid; i < NTOK * D; i += 256) {
    const int tok = i / D, r = i - tok * D;
    k2s[i] = ek[((int64_t)b * NTOK + tok) * D + r] + tf * wk[(int64_t)r * wk_stride];
    v2s[i] = P[((int64_t)b * NTOK + tok) * ldp + v2_off + r];
  }
  float q[NTOK];
#pragma unroll
  for (int tok = 0; tok < NTOK; ++tok) q[tok] = P[((int64_t)b * NTOK + tok) * ldp + q2_off + cb * CA_COLS + c] * scale;
  __syncthreads();

  float mx = -INFINITY;
  for (int rr = 0; rr < 64; ++rr) {
    const int r = rg * 64 + rr;
    float a = 0.f;
#pragma unroll
    for (int tok = 0; tok < NTOK; ++tok) a = fmaf(k2s[tok * D + r], q[tok], a);
    es[r * CA_PITCH + c] = a;
    mx = fmaxf(mx, a);
  }
  red[rg * 32 + c] = mx;
  __syncthreads();
  mx = red[c];
#pragma unroll
  for (int g = 1; g < 8; ++g) mx = fmaxf(mx, red[g * 32 + c]);
  __syncthreads();
  float sum = 0.f;
  for (int rr = 0; rr < 64; ++rr) {
    const int r = rg * 64 + rr;
    const float e = expf(es[r * CA_PITCH + c] - mx);
    es[r * CA_PITCH + c] = e;
    sum += e;
  }
  red[rg * 32 + c] = sum;
  __syncthreads();
  float den = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) den += red[g * 32 + c];
  // t[tok][c] for tok = rg, rg + 8, rg + 16
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  const bool has2 = rg + 16 < NTOK;
  const float* v0 = v2s + rg * D;
  const float* v1 = v2s + (rg + 8) * D;
  const float* v2p = v2s + (has2 ? rg + 16 : 0) * D;
  for (int r = 0; r < D; ++r) {
    const float e = es[r * CA_PITCH + c];
    acc0 = fmaf(v0[r], e, acc0);
    acc1 = fmaf(v1[r], e, acc1);
    acc2 = fmaf(v2p[r], e, acc2);
  }
  const float inv = 1.f / den;
  float* tb = tout + (int64_t)b * NTOK * D + cb * CA_COLS + c;
  tb[(int64_t)rg * D] = acc0 * inv;
  tb[(int64_t)(rg + 8) * D] = acc1 * inv;
  if (has2) tb[(int64_t)(rg + 16) * D] = acc2 * inv;
}


__global__ __launch_bounds__(64 * vsptacc::CA_NW) void tacc_chan_attn_mfma_kernel(float* __restrict__ tout, const float* __restrict__ P,
                                                                   int ldp, int q2_off, int v2_off,
                                                                   const float* __restrict__ ek, const float* __restrict__ wk,
                                                                   int wk_stride, float tf, float scale) {
  vsptacc::chan_attn_mfma_body(tout, P, ldp, q2_off, v2_off, ek, wk, wk_stride, tf, scale, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tail with the token attention folded in.  grid = B, 576 threads (9 waves, 2 token rows each).  V and Q = eQ + tf*wq are
// staged once per sample in LDS (float4, coalesced); per row i (8 channels per lane):
//   s = softmax_j(K_i . Q_j / sqrt(18));  h = sum_j s_j V_j;  tn = LN(t_i);  hn = LN(h + tn);  y = hn * (1 + gamma) + beta
//   mix: y = c1[idx] * y + c2[idx] * xold                       (DDPM posterior mean after the 4th block)
//   yout = y;  pn = y * rsqrt(mean_i y^2 + 1e-8)                 (PixelNorm over the 18 tokens for the next GEMM)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(576) void tacc_tail_kernel(float* __restrict__ yout, float* __restrict__ pnout,
                                                         const float* __restrict__ P, int ldp, int k_off, int v_off,
                                                         const float* __restrict__ eQ, const float* __restrict__ wq,
                                                         float tf, float sscale, const float* __restrict__ t,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ xold, const float* __restrict__ c1,
                                                         const float* __restrict__ c2, int idx, float eps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;             // [18][512]
  float* Qs = Vs + NTOK * D;    // [18][512]
  float* ys = Qs + NTOK * D;    // [18][512]
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NTOK * D / 4; i += 576) {
    const int tok = i / (D / 4), c4 = i - tok * (D / 4);
    const float4 v = *reinterpret_cast<const float4*>(P + ((int64_t)b * NTOK + tok) * ldp + v_off + c4 * 4);
    float4 q = *reinterpret_cast<const float4*>(eQ + ((int64_t)b * NTOK + tok) * D + c4 * 4);
    const float4 w = *reinterpret_cast<const float4*>(wq + c4 * 4);
    q.x = fmaf(tf, w.x, q.x); q.y = fmaf(tf, w.y, q.y); q.z = fmaf(tf, w.z, q.z); q.w = fmaf(tf, w.w, q.w);
    *reinterpret_cast<float4*>(Vs + tok * D + c4 * 4) = v;
    *reinterpret_cast<float4*>(Qs + tok * D + c4 * 4) = q;
  }
  __syncthreads();
  for (int half = 0; half < 2; ++half) {
    const int i = wave * 2 + half;
    const int64_t row = (int64_t)b * NTOK + i;
    float kv[8], h[8], tn[8];
    const float* kr = P + row * ldp + k_off;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      kv[u] = kr[lane + 64 * u];
      tn[u] = t[row * D + lane + 64 * u];
      h[u] = 0.f;
    }
    float s[NTOK];
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(kv[u], Qs[j * D + lane + 64 * u], acc);
      s[j] = wsum(acc) * sscale;
    }
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
#pragma unroll
    for (int j = 0; j < NTOK; ++j) {
      const float sj = s[j] * rden;
#pragma unroll
      for (int u = 0; u < 8; ++u) h[u] = fmaf(sj, Vs[j * D + lane + 64 * u], h[u]);
    }
    float sm = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) sm += tn[u];
    float mean = wsum(sm) * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float d = tn[u] - mean;
      var = fmaf(d, d, var);
    }
    float inv = rsqrtf(wsum(var) * (1.f / D) + eps);
    sm = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h[u] += (tn[u] - mean) * inv;
      sm += h[u];
    }
    mean = wsum(sm) * (1.f / D);
    var = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float d = h[u] - mean;
      var = fmaf(d, d, var);
    }
    inv = rsqrtf(wsum(var) * (1.f / D) + eps);
    const float a1 = xold ? c1[idx] : 1.f, a2 = xold ? c2[idx] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t o = row * D + lane + 64 * u;
      float y = (h[u] - mean) * inv * (1.f + gamma[o]) + beta[o];
      if (xold) y = a1 * y + a2 * xold[o];
      yout[o] = y;
      ys[i * D + lane + 64 * u] = y;
    }
  }
  __syncthreads();
  if (pnout && threadIdx.x < D) {
    const int ch = threadIdx.x;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NTOK; ++i) sq = fmaf(ys[i * D + ch], ys[i * D + ch], sq);
    const float inv = rsqrtf(sq * (1.f / NTOK) + 1e-8f);
#pragma unroll
    for (int i = 0; i < NTOK; ++i) pnout[((int64_t)b * NTOK + i) * D + ch] = ys[i * D + ch] * inv;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// First layer of the gamma_/beta_ heads for ALL steps at once: out[s, m, :] = lrelu(LN(e[m, :] + tf_s * wcol) * g + b) * sqrt2
// with tf_s = s / t_div.  One wave per (s, m) row.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tacc_head_pre_kernel(float* __restrict__ out, const float* __restrict__ e,
                                                             const float* __restrict__ wcol, int w_stride,
                                                             const float* __restrict__ g, const float* __restrict__ bb,
                                                             int S, int M, float t_div, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)S * M) return;
  const int s = (int)(row / M), m = (int)(row % M);
  const float tf = (float)s / t_div;
  float v[8];
  float sm = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int ch = lane + 64 * u;
    v[u] = e[(int64_t)m * D + ch] + tf * wcol[(int64_t)ch * w_stride];
    sm += v[u];
  }
  const float mean = wsum(sm) * (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float d = v[u] - mean;
    var = fmaf(d, d, var);
  }
  const float inv = rsqrtf(wsum(var) * (1.f / D) + eps);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int ch = lane + 64 * u;
    float y = (v[u] - mean) * inv * g[ch] + bb[ch];
    y = (y > 0.f ? y : y * 0.2f) * 1.41421356237309515f;
    out[row * D + ch] = y;
  }
}

}  // namespace

extern "C" {

int vsp_tacc_scores_f32(float* score, const float* P, int ldp, int k_off, const float* eQ, const float* wq, int wq_stride,
                        float tfrac, int B, int n_tok, int dim, vsp_stream_t stream) {
  VSP_REQUIRE(n_tok == NTOK && dim == D, "tacc_scores: built for 18 tokens x 512 channels (got %d x %d)", n_tok, dim);
  if (B <= 0) return VSP_OK;
  VSP_REQUIRE(score && P && eQ && wq, "tacc_scores: null pointer");
  tacc_scores_kernel<<<B, 320, 0, vsp::as_stream(stream)>>>(score, P, ldp, k_off, eQ, wq, wq_stride, tfrac,
                                                            1.0f / sqrtf((float)NTOK));
  return vsp::check_launch("tacc_scores");
}

int vsp_tacc_chan_attn_f32(float* t, const float* P, int ldp, int q2_off, int v2_off, const float* ek, const float* wk,
                           int wk_stride, float tfrac, int B, int n_tok, int dim, vsp_stream_t stream) {
  VSP_REQUIRE(n_tok == NTOK && dim == D, "tacc_chan_attn: built for 18 tokens x 512 channels (got %d x %d)", n_tok, dim);
  if (B <= 0) return VSP_OK;
  VSP_REQUIRE(t && P && ek && wk, "tacc_chan_attn: null pointer");
  static const bool use_valu = vsp::tune_env("VSP_TACC_VALU") != nullptr;  // the first (VALU/LDS) version, kept for A/B runs
  const size_t lds = use_valu ? (size_t)(2 * NTOK * D + D * CA_PITCH + 8 * 32) * sizeof(float)
                              : vsptacc::CA_LDS_FLOATS * sizeof(float);
  static vsp::LdsAttrOnce attr_a, attr_b;   // per device
  if (int rc = attr_a.ensure(reinterpret_cast<const void*>(tacc_chan_attn_kernel), 150 * 1024, "tacc_chan_attn")) return rc;
  if (int rc = attr_b.ensure(reinterpret_cast<const void*>(tacc_chan_attn_mfma_kernel), 150 * 1024, "tacc_chan_attn")) return rc;
  dim3 grid(D / CA_COLS, B);
  if (use_valu)
    tacc_chan_attn_kernel<<<grid, 256, lds, vsp::as_stream(stream)>>>(t, P, ldp, q2_off, v2_off, ek, wk, wk_stride, tfrac,
                                                                      1.0f / sqrtf((float)D));
  else
    tacc_chan_attn_mfma_kernel<<<grid, 64 * vsptacc::CA_NW, lds, vsp::as_stream(stream)>>>(t, P, ldp, q2_off, v2_off, ek, wk, wk_stride,
                                                                           tfrac, 1.0f / sqrtf((float)D));
  return vsp::check_launch("tacc_chan_attn");
}

int vsp_tacc_tail_f32(float* y, float* pn, const float* P, int ldp, int k_off, int v_off, const float* eQ, const float* wq,
                      float tfrac, const float* t, const float* gamma, const float* beta, const float* xold, const float* c1,
                      const float* c2, int idx, int B, int n_tok, int dim, vsp_stream_t stream) {
  VSP_REQUIRE(n_tok == NTOK && dim == D, "tacc_tail: built for 18 tokens x 512 channels (got %d x %d)", n_tok, dim);
  if (B <= 0) return VSP_OK;
  VSP_REQUIRE(y && P && eQ && wq && t && gamma && beta, "tacc_tail: null pointer");
  VSP_REQUIRE(!xold || (c1 && c2 && idx >= 0), "tacc_tail: posterior mix needs c1, c2 and a step index");
  VSP_REQUIRE(ldp % 4 == 0 && k_off % 4 == 0 && v_off % 4 == 0 && vsp::aligned16(P) && vsp::aligned16(eQ) && vsp::aligned16(wq),
              "tacc_tail: operands must be 16-byte aligned (contiguous wq column expected)");
  const size_t lds = (size_t)3 * NTOK * D * sizeof(float);
  static vsp::LdsAttrOnce attr;   // per device
  if (int rc = attr.ensure(reinterpret_cast<const void*>(tacc_tail_kernel), (int)lds, "tacc_tail")) return rc;
  tacc_tail_kernel<<<B, 576, lds, vsp::as_stream(stream)>>>(y, pn, P, ldp, k_off, v_off, eQ, wq, tfrac,
                                                            1.0f / sqrtf((float)NTOK), t, gamma, beta, xold, c1, c2, idx, 1e-5f);
  return vsp::check_launch("tacc_tail");
}

int vsp_tacc_head_pre_f32(float* out, const float* e, const float* wcol, int w_stride, const float* ln_w, const float* ln_b,
                          int S, int M, int dim, float t_div, vsp_stream_t stream) {
  VSP_REQUIRE(dim == D, "tacc_head_pre: built for 512 channels (got %d)", dim);
  if (S <= 0 || M <= 0) return VSP_OK;
  VSP_REQUIRE(out && e && wcol && ln_w && ln_b && t_div > 0.f, "tacc_head_pre: bad argument");
  const int64_t rows = (int64_t)S * M;
  tacc_head_pre_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, vsp::as_stream(stream)>>>(out, e, wcol, w_stride, ln_w, ln_b,
                                                                                      S, M, t_div, 1e-5f);
  return vsp::check_launch("tacc_head_pre");
}

}  // extern "C"

// Small HBM/L2-bound helpers of the path (normalisations, softmaxes, pooling, resampling, gating) for gfx950.
// Every kernel here moves O(bytes) once with coalesced accesses along the contiguous dimension; reductions over a
// row use one wave64 per row with __shfl_xor butterflies (64 lanes, no LDS).  Reference statements are cited on the
// declarations in include/vspbfr_hip.h.
#include "vsp_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int stream_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > vsp::kMaxStreamBlocks) b = vsp::kMaxStreamBlocks;
  return (int)(b < 1 ? 1 : b);
}

// ---- PixelNorm over dim 1 of [Z,R,C]: thread per (z,c), coalesced along c
__global__ __launch_bounds__(256) void pixelnorm_dim1_kernel(float* out, const float* x, int Z, int R, int C, float eps) {
  const int64_t total = (int64_t)Z * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = (int)(i / C), c = (int)(i % C);
    const float* xp = x + (int64_t)z * R * C + c;
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
      const float v = xp[(int64_t)r * C];
      s = fmaf(v, v, s);
    }
    const float inv = rsqrtf(s / (float)R + eps);
    float* op = out + (int64_t)z * R * C + c;
    for (int r = 0; r < R; ++r) op[(int64_t)r * C] = xp[(int64_t)r * C] * inv;
  }
}

// ---- LayerNorm over the last dim: one wave per row
__global__ __launch_bounds__(256) void layernorm_kernel(float* out, const float* x, const float* add, const float* gamma,
                                                         const float* beta, int rows, int cols, float eps, int post,
                                                         float slope, float gain) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * cols;
  const float* ar = add ? add + (int64_t)row * cols : nullptr;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c] + (ar ? ar[c] : 0.f);
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float d = xr[c] + (ar ? ar[c] : 0.f) - mean;
    q = fmaf(d, d, q);
  }
  const float inv = rsqrtf(wave_sum(q) / (float)cols + eps);
  float* orow = out + (int64_t)row * cols;
  for (int c = lane; c < cols; c += 64) {
    float v = (xr[c] + (ar ? ar[c] : 0.f) - mean) * inv;
    if (gamma) v = v * gamma[c] + (beta ? beta[c] : 0.f);
    if (post == 1) v = (v > 0.f ? v : v * slope) * gain;
    orow[c] = v;
  }
}

// ---- softmax over the last dim: one wave per row
__global__ __launch_bounds__(256) void softmax_last_kernel(float* out, const float* x, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += expf(xr[c] - m);
  s = wave_sum(s);
  float* orow = out + (int64_t)row * cols;
  for (int c = lane; c < cols; c += 64) orow[c] = expf(xr[c] - m) / s;
}

// ---- softmax over dim 1 of [Z,R,C]: thread per (z,c)
__global__ __launch_bounds__(256) void softmax_dim1_kernel(float* out, const float* x, int Z, int R, int C) {
  const int64_t total = (int64_t)Z * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = (int)(i / C), c = (int)(i % C);
    const float* xp = x + (int64_t)z * R * C + c;
    float m = -INFINITY;
    for (int r = 0; r < R; ++r) m = fmaxf(m, xp[(int64_t)r * C]);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += expf(xp[(int64_t)r * C] - m);
    float* op = out + (int64_t)z * R * C + c;
    for (int r = 0; r < R; ++r) op[(int64_t)r * C] = expf(xp[(int64_t)r * C] - m) / s;
  }
}

__global__ __launch_bounds__(256) void film_kernel(float* out, const float* h, const float* gamma, const float* beta, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = h[i] * (1.f + gamma[i]) + beta[i];
}

__global__ __launch_bounds__(256) void axpby_idx_kernel(float* out, const float* x, const float* y, const float* a, const float* b,
                                                         int idx, int64_t n) {
  const float ca = a[idx], cb = b[idx];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = ca * x[i] + cb * y[i];
}

// ---- demodulation coefficients: one wave per (b, co)
__global__ __launch_bounds__(256) void demod_kernel(float* out, const float* style, const float* wsq, int B, int Cin, int Cout,
                                                     float wscale2, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Cout) return;
  const int b = (int)(row / Cout), co = (int)(row % Cout);
  const float* sp = style + (int64_t)b * Cin;
  const float* wp = wsq + (int64_t)co * Cin;
  float s = 0.f;
  for (int c = lane; c < Cin; c += 64) {
    const float st = sp[c];
    s = fmaf(st * st, wp[c], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = rsqrtf(s * wscale2 + eps);
}

// ---- demodulation under autograd (training): the coefficients straight from the weight, and their gradient.  The torch expression
//      rsqrt(linear(s^2, w^2.sum(taps)) * c + eps) is 7 launches forward and ~14 backward per modulated layer (61 layers per iteration,
//      every one a few microseconds of a 2 MB tensor); here: one launch forward, two backward.
constexpr int kDemodMaxB = 16;
// one workgroup per output channel: wsq[co, :] = sum_taps w^2 (kept for the backward), out[b, co] for every sample.  The channel's
// Cin x K weights are one contiguous run: it is staged through LDS with coalesced 4-byte loads (a thread reading its own K taps at
// stride K touched 18 cache lines per load instruction), then every thread sums the taps of its input channels from LDS.
__global__ __launch_bounds__(256) void demod_weight_kernel(float* __restrict__ out, float* __restrict__ wsq, const float* __restrict__ style,
                                                            const float* __restrict__ w, int B, int Cin, int Cout, int K, float wscale2,
                                                            float eps) {
  extern __shared__ float dmw[];   // [chunk of 256 input channels][K] (+1 per row: odd pitch, conflict-free tap walks)
  const int co = blockIdx.x;
  const int KP = K | 1;
  float acc[kDemodMaxB];
#pragma unroll
  for (int b = 0; b < kDemodMaxB; ++b) acc[b] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += 256) {
    const int nc = Cin - c0 < 256 ? Cin - c0 : 256;
    const float* src = w + ((int64_t)co * Cin + c0) * K;
    __syncthreads();
    for (int i = threadIdx.x; i < nc * K; i += 256) dmw[(i / K) * KP + i % K] = src[i];
    __syncthreads();
    const int ci = c0 + threadIdx.x;
    if ((int)threadIdx.x < nc) {
      const float* wp = dmw + threadIdx.x * KP;
      float q = 0.f;
      for (int k = 0; k < K; ++k) q = fmaf(wp[k], wp[k], q);
      wsq[(int64_t)co * Cin + ci] = q;
#pragma unroll
      for (int b = 0; b < kDemodMaxB; ++b)
        if (b < B) {
          const float st = style[(int64_t)b * Cin + ci];
          acc[b] = fmaf(st * st, q, acc[b]);
        }
    }
  }
  __shared__ float red[4][kDemodMaxB];
#pragma unroll
  for (int b = 0; b < kDemodMaxB; ++b) {
    const float v = wave_sum(acc[b]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][b] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < B) {
    const int b = threadIdx.x;
    out[(int64_t)b * Cout + co] = rsqrtf((red[0][b] + red[1][b] + red[2][b] + red[3][b]) * wscale2 + eps);
  }
}
// t[b, co] = -0.5 wscale^2 out^3 g  (d out / d (sum s^2 wsq));  dw[co, ci, k] (+)= 2 w[co, ci, k] sum_b t[b, co] s[b, ci]^2: one thread per (co, ci).
// `acc`: add to what dw holds (the convolution's own weight gradient: one tensor leaves the layer's backward, autograd adds nothing).
__global__ __launch_bounds__(256) void demod_weight_bwd_w_kernel(float* __restrict__ dw, const float* __restrict__ g, int gs,
                                                                  const float* __restrict__ out, const float* __restrict__ style,
                                                                  const float* __restrict__ w, int B, int Cin, int Cout, int K, float c, int acc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Cout * Cin) return;
  const int co = (int)(i / Cin), ci = (int)(i - (int64_t)co * Cin);
  float d = 0.f;
  for (int b = 0; b < B; ++b) {
    const float o = out[(int64_t)b * gs + co], st = style[(int64_t)b * Cin + ci];
    d = fmaf(c * g[(int64_t)b * gs + co] * o * o * o, st * st, d);
  }
  d *= 2.f;
  const float* wp = w + i * K;
  float* dp = dw + i * K;
  if (acc) {
    for (int k = 0; k < K; ++k) dp[k] = fmaf(wp[k], d, dp[k]);
  } else {
    for (int k = 0; k < K; ++k) dp[k] = wp[k] * d;
  }
}
// dstyle[b, ci] (+)= 2 s[b, ci] sum_co t[b, co] wsq[co, ci]: a workgroup = one sample x 64 input channels; t of the sample goes to LDS once
// (the first version recomputed it from three global loads per (co, ci): 16 us per launch), the output channels run in 4 quarters with the
// wsq loads independent of each other
__global__ __launch_bounds__(256) void demod_weight_bwd_s_kernel(float* __restrict__ ds, const float* __restrict__ g, int gs,
                                                                  const float* __restrict__ out, const float* __restrict__ style,
                                                                  const float* __restrict__ wsq, int Cin, int Cout, float c, int acc) {
  extern __shared__ float dw_t[];   // [Cout] t, then [4][64] partial sums
  const int b = blockIdx.y, ci = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  for (int co = threadIdx.x; co < Cout; co += 256) {
    const float o = out[(int64_t)b * gs + co];
    dw_t[co] = c * g[(int64_t)b * gs + co] * o * o * o;
  }
  __syncthreads();
  float a0 = 0.f, a1 = 0.f;
  if (ci < Cin) {
    int co = q;
    for (; co + 4 < Cout; co += 8) {
      a0 = fmaf(dw_t[co], wsq[(int64_t)co * Cin + ci], a0);
      a1 = fmaf(dw_t[co + 4], wsq[(int64_t)(co + 4) * Cin + ci], a1);
    }
    for (; co < Cout; co += 4) a0 = fmaf(dw_t[co], wsq[(int64_t)co * Cin + ci], a0);
  }
  float* red = dw_t + Cout;
  red[q * 64 + (threadIdx.x & 63)] = a0 + a1;
  __syncthreads();
  if (q == 0 && ci < Cin) {
    const int l = threadIdx.x;
    const float v = 2.f * style[(int64_t)b * Cin + ci] * ((red[l] + red[64 + l]) + (red[128 + l] + red[192 + l]));
    ds[(int64_t)b * Cin + ci] = acc ? ds[(int64_t)b * Cin + ci] + v : v;
  }
}

__global__ __launch_bounds__(256) void avgpool2x2_kernel(float* out, const float* x, int64_t planes, int OH, int OW) {
  const int64_t total = planes * OH * OW;
  const int IW = 2 * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    const float* xp = x + (pl * 2 * OH + 2 * oy) * IW + 2 * ox;
    const float2 r0 = *reinterpret_cast<const float2*>(xp);
    const float2 r1 = *reinterpret_cast<const float2*>(xp + IW);
    out[i] = ((r0.x + r0.y) + (r1.x + r1.y)) * 0.25f;
  }
}

// bilinear, align_corners=True: src = dst * (I-1)/(O-1)
__global__ __launch_bounds__(256) void upsample_add_kernel(float* out, const float* x, const float* y, int64_t planes, int IH,
                                                            int IW, int OH, int OW, float ry, float rx) {
  const int64_t total = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    const float fy = ry * (float)oy, fx = rx * (float)ox;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > IH - 1) y0 = IH - 1;
    if (x0 > IW - 1) x0 = IW - 1;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* xp = x + pl * IH * IW;
    const float v = hy * (hx * xp[y0 * IW + x0] + lx * xp[y0 * IW + x1]) + ly * (hx * xp[y1 * IW + x0] + lx * xp[y1 * IW + x1]);
    out[i] = v + y[i];
  }
}

// bilinear, align_corners=False, no antialiasing (F.interpolate(mode="bilinear"), the 256^2 resize in front of the e4e encoder,
// Loss/e4e_embedding.py:91-100): src = (dst + 0.5) * I / O - 0.5 clamped at 0, neighbours clamped at I - 1
__global__ __launch_bounds__(256) void resize_bilinear_kernel(float* out, const float* x, int64_t planes, int IH, int IW, int OH,
                                                               int OW, float sy, float sx) {
  const int64_t total = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > IH - 1) y0 = IH - 1;
    if (x0 > IW - 1) x0 = IW - 1;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* xp = x + pl * IH * IW;
    out[i] = hy * (hx * xp[y0 * IW + x0] + lx * xp[y0 * IW + x1]) + ly * (hx * xp[y1 * IW + x0] + lx * xp[y1 * IW + x1]);
  }
}

// one wave per plane
__global__ __launch_bounds__(256) void plane_mean_kernel(float* out, const float* x, int64_t planes, int hw) {
  const int lane = threadIdx.x & 63;
  const int64_t pl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const float* xp = x + pl * hw;
  float s = 0.f;
  for (int i = lane; i < hw; i += 64) s += xp[i];
  s = wave_sum(s);
  if (lane == 0) out[pl] = s / (float)hw;
}

__global__ __launch_bounds__(256) void scale_add_kernel(float* out, const float* x, const float* gate, const float* y, int64_t planes,
                                                         int hw) {
  const int64_t total = planes * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = gate[i / hw];
    out[i] = x[i] * g + (y ? y[i] : 0.f);
  }
}

__global__ __launch_bounds__(256) void subsample_kernel(float* out, const float* x, int64_t planes, int IH, int IW, int OH, int OW,
                                                         int s) {
  const int64_t total = planes * OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t pl = t / OH;
    out[i] = x[(pl * IH + (int64_t)oy * s) * IW + (int64_t)ox * s];
  }
}

__global__ __launch_bounds__(256) void quantize_u8_nhwc_kernel(uint8_t* out, const float* x, int B, int C, int H, int W, float lo,
                                                               float hi, float range) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int xw = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float v = x[(((int64_t)b * C + c) * H + y) * W + xw];
    v = fminf(fmaxf(v, lo), hi);
    v = (v - lo) / range;
    v = fminf(fmaxf(v * 255.f + 0.5f, 0.f), 255.f);
    out[i] = (uint8_t)v;
  }
}

__global__ __launch_bounds__(256) void add3_kernel(float* out, const float* a, const float* b, const float* c, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i] + (c ? c[i] : 0.f);
}

}  // namespace

// The latent plumbing of a batch (a few hundred KB; the reference runs it as repeat / cat / flip / add on (B, 18, 512) tensors):
//   e4e codes: codes[b, t, :] = heads[t, b, :] + (t > 0 ? heads[0, b, :] : 0) + latent_avg[t, :]     (psp_encoders.py:188-199, psp.py:159-165)
//   row concat: out[b, t, :] = [seg0 | seg1 | seg2], seg k = src_k[b * bs_k + tt * ts_k + j], tt = t or T-1-t (flip), ts_k = 0 broadcasts
__global__ __launch_bounds__(256) void e4e_codes_kernel(float* __restrict__ out, const float* __restrict__ heads,
                                                         const float* __restrict__ avg, int B, int T, int D) {
  const int64_t total = (int64_t)B * T * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % D);
    const int64_t r = i / D;
    const int t = (int)(r % T), b = (int)(r / T);
    float v = heads[((int64_t)t * B + b) * D + j];
    if (t > 0) v += heads[(int64_t)b * D + j];
    if (avg) v += avg[t * D + j];
    out[i] = v;
  }
}

struct ConcatSeg {
  const float* src;
  int bs, ts, width, flip;
};
struct ConcatArgs {
  ConcatSeg seg[3];
  int nseg, B, T, W;
};

__global__ __launch_bounds__(256) void rows_concat_kernel(float* __restrict__ out, ConcatArgs a) {
  const int64_t total = (int64_t)a.B * a.T * a.W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int j = (int)(i % a.W);
    const int64_t r = i / a.W;
    const int t = (int)(r % a.T), b = (int)(r / a.T);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k >= a.nseg) break;
      const ConcatSeg sg = a.seg[k];
      if (j < sg.width) {
        const int tt = sg.flip ? a.T - 1 - t : t;
        v = sg.src[(int64_t)b * sg.bs + (int64_t)tt * sg.ts + j];
        break;
      }
      j -= sg.width;
    }
    out[i] = v;
  }
}

#define VSP_LAUNCH_1D(kern, n, s, ...)                                         \
  do {                                                                         \
    kern<<<stream_blocks(n), 256, 0, vsp::as_stream(s)>>>(__VA_ARGS__);        \
  } while (0)

extern "C" {

int vsp_pixelnorm_dim1_f32(float* out, const float* x, int Z, int R, int C, float eps, vsp_stream_t stream) {
  VSP_REQUIRE(Z >= 0 && R >= 1 && C >= 0, "pixelnorm: bad dims");
  if ((int64_t)Z * C == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "pixelnorm: null pointer");
  VSP_LAUNCH_1D(pixelnorm_dim1_kernel, (int64_t)Z * C, stream, out, x, Z, R, C, eps);
  return vsp::check_launch("pixelnorm_dim1");
}

int vsp_layernorm_f32(float* out, const float* x, const float* add, const float* gamma, const float* beta, int rows,
                      int cols, float eps, int post, float slope, float gain, vsp_stream_t stream) {
  VSP_REQUIRE(rows >= 0 && cols >= 1, "layernorm: bad dims");
  if (rows == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "layernorm: null pointer");
  layernorm_kernel<<<(rows + 3) / 4, 256, 0, vsp::as_stream(stream)>>>(out, x, add, gamma, beta, rows, cols, eps, post,
                                                                        slope, gain);
  return vsp::check_launch("layernorm");
}

int vsp_softmax_lastdim_f32(float* out, const float* x, int rows, int cols, vsp_stream_t stream) {
  VSP_REQUIRE(rows >= 0 && cols >= 1, "softmax: bad dims");
  if (rows == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "softmax: null pointer");
  softmax_last_kernel<<<(rows + 3) / 4, 256, 0, vsp::as_stream(stream)>>>(out, x, rows, cols);
  return vsp::check_launch("softmax_lastdim");
}

int vsp_softmax_dim1_f32(float* out, const float* x, int Z, int R, int C, vsp_stream_t stream) {
  VSP_REQUIRE(Z >= 0 && R >= 1 && C >= 0, "softmax_dim1: bad dims");
  if ((int64_t)Z * C == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "softmax_dim1: null pointer");
  VSP_LAUNCH_1D(softmax_dim1_kernel, (int64_t)Z * C, stream, out, x, Z, R, C);
  return vsp::check_launch("softmax_dim1");
}

int vsp_film_f32(float* out, const float* h, const float* gamma, const float* beta, int64_t n, vsp_stream_t stream) {
  if (n <= 0) return VSP_OK;
  VSP_REQUIRE(out && h && gamma && beta, "film: null pointer");
  VSP_LAUNCH_1D(film_kernel, n, stream, out, h, gamma, beta, n);
  return vsp::check_launch("film");
}

int vsp_axpby_idx_f32(float* out, const float* x, const float* y, const float* a, const float* b, int idx, int64_t n,
                      vsp_stream_t stream) {
  if (n <= 0) return VSP_OK;
  VSP_REQUIRE(out && x && y && a && b && idx >= 0, "axpby: bad argument");
  VSP_LAUNCH_1D(axpby_idx_kernel, n, stream, out, x, y, a, b, idx, n);
  return vsp::check_launch("axpby_idx");
}

int vsp_demod_f32(float* out, const float* style, const float* wsq, int B, int Cin, int Cout, float wscale, float eps,
                  vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1, "demod: bad dims");
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(out && style && wsq, "demod: null pointer");
  const int64_t rows = (int64_t)B * Cout;
  demod_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, vsp::as_stream(stream)>>>(out, style, wsq, B, Cin, Cout,
                                                                                wscale * wscale, eps);
  return vsp::check_launch("demod");
}

int vsp_demod_weight_f32(float* out, float* wsq, const float* style, const float* w, int B, int Cin, int Cout, int K, float wscale, float eps,
                         vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && K >= 1, "demod_weight: bad dims");
  if (B > kDemodMaxB) return vsp::fail(VSP_ENOTSUP, "demod_weight: at most %d samples per call (got %d)", kDemodMaxB, B);
  VSP_REQUIRE(out && wsq && style && w, "demod_weight: null pointer");
  VSP_REQUIRE(K <= 49, "demod_weight: at most 49 taps (got %d)", K);
  demod_weight_kernel<<<(unsigned)Cout, 256, (size_t)256 * (K | 1) * sizeof(float), vsp::as_stream(stream)>>>(out, wsq, style, w, B, Cin, Cout, K,
                                                                                                        wscale * wscale, eps);
  return vsp::check_launch("demod_weight");
}

int vsp_demod_weight_bwd_f32(float* dstyle, float* dw, const float* g, const float* out, const float* style, const float* wsq, const float* w,
                             int B, int Cin, int Cout, int K, float wscale, vsp_stream_t stream) {
  return vsp_demod_weight_bwd_acc_f32(dstyle, dw, g, Cout, out, style, wsq, w, B, Cin, Cout, K, wscale, 0, stream);
}

int vsp_demod_weight_bwd_acc_f32(float* dstyle, float* dw, const float* g, int g_stride, const float* out, const float* style, const float* wsq,
                                 const float* w, int B, int Cin, int Cout, int K, float wscale, int accumulate, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && K >= 1, "demod_weight_bwd: bad dims");
  VSP_REQUIRE(g && out && style && (!dstyle || wsq) && (!dw || w), "demod_weight_bwd: null pointer");
  VSP_REQUIRE(B <= 65535 && g_stride >= Cout, "demod_weight_bwd: batch too large / gradient row pitch below Cout");
  const float c = -0.5f * wscale * wscale;
  hipStream_t st = vsp::as_stream(stream);
  if (dw) {
    const int64_t n = (int64_t)Cout * Cin;
    demod_weight_bwd_w_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dw, g, g_stride, out, style, w, B, Cin, Cout, K, c, accumulate ? 1 : 0);
  }
  if (dstyle && B > 0) {
    const size_t lds = (size_t)(Cout + 256) * sizeof(float);
    VSP_REQUIRE(lds <= 64 * 1024, "demod_weight_bwd: too many output channels (%d)", Cout);
    demod_weight_bwd_s_kernel<<<dim3((unsigned)((Cin + 63) / 64), (unsigned)B), 256, lds, st>>>(dstyle, g, g_stride, out, style, wsq, Cin, Cout, c,
                                                                                                   accumulate ? 1 : 0);
  }
  return vsp::check_launch("demod_weight_bwd");
}

int vsp_avgpool2x2_f32(float* out, const float* x, int64_t planes, int OH, int OW, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && OH >= 0 && OW >= 0, "avgpool2x2: bad dims");
  const int64_t n = planes * OH * OW;
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "avgpool2x2: null pointer");
  VSP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7u) == 0, "avgpool2x2: input must be 8-byte aligned");
  VSP_LAUNCH_1D(avgpool2x2_kernel, n, stream, out, x, planes, OH, OW);
  return vsp::check_launch("avgpool2x2");
}

int vsp_upsample_add_f32(float* out, const float* x, const float* y, int64_t planes, int IH, int IW, int OH, int OW,
                         vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "upsample_add: bad dims");
  const int64_t n = planes * OH * OW;
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x && y, "upsample_add: null pointer");
  const float ry = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
  const float rx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
  VSP_LAUNCH_1D(upsample_add_kernel, n, stream, out, x, y, planes, IH, IW, OH, OW, ry, rx);
  return vsp::check_launch("upsample_add");
}

int vsp_resize_bilinear_f32(float* out, const float* x, int64_t planes, int IH, int IW, int OH, int OW, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "resize_bilinear: bad dims");
  const int64_t n = planes * OH * OW;
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "resize_bilinear: null pointer");
  VSP_LAUNCH_1D(resize_bilinear_kernel, n, stream, out, x, planes, IH, IW, OH, OW, (float)IH / (float)OH, (float)IW / (float)OW);
  return vsp::check_launch("resize_bilinear");
}

int vsp_plane_mean_f32(float* out, const float* x, int64_t planes, int hw, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && hw >= 1, "plane_mean: bad dims");
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "plane_mean: null pointer");
  plane_mean_kernel<<<(unsigned)((planes + 3) / 4), 256, 0, vsp::as_stream(stream)>>>(out, x, planes, hw);
  return vsp::check_launch("plane_mean");
}

int vsp_scale_add_f32(float* out, const float* x, const float* gate, const float* y, int64_t planes, int hw,
                      vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && hw >= 1, "scale_add: bad dims");
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(out && x && gate, "scale_add: null pointer");
  VSP_LAUNCH_1D(scale_add_kernel, planes * hw, stream, out, x, gate, y, planes, hw);
  return vsp::check_launch("scale_add");
}

int vsp_subsample_f32(float* out, const float* x, int64_t planes, int IH, int IW, int s, vsp_stream_t stream) {
  VSP_REQUIRE(planes >= 0 && IH >= 1 && IW >= 1 && s >= 1, "subsample: bad dims");
  const int OH = (IH - 1) / s + 1, OW = (IW - 1) / s + 1;
  if (planes == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "subsample: null pointer");
  VSP_LAUNCH_1D(subsample_kernel, planes * OH * OW, stream, out, x, planes, IH, IW, OH, OW, s);
  return vsp::check_launch("subsample");
}

int vsp_quantize_u8_nhwc(uint8_t* out, const float* x, int B, int C, int H, int W, float lo, float hi, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && C >= 1 && H >= 0 && W >= 0, "quantize_u8: bad dims");
  const int64_t n = (int64_t)B * C * H * W;
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "quantize_u8: null pointer");
  const float range = hi - lo > 1e-5f ? hi - lo : 1e-5f;
  VSP_LAUNCH_1D(quantize_u8_nhwc_kernel, n, stream, out, x, B, C, H, W, lo, hi, range);
  return vsp::check_launch("quantize_u8_nhwc");
}

int vsp_e4e_codes_f32(float* out, const float* heads, const float* latent_avg, int B, int T, int D, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && T >= 1 && D >= 1, "e4e_codes: bad dims");
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(out && heads, "e4e_codes: null pointer");
  VSP_LAUNCH_1D(e4e_codes_kernel, (int64_t)B * T * D, stream, out, heads, latent_avg, B, T, D);
  return vsp::check_launch("e4e_codes");
}

int vsp_rows_concat_f32(float* out, int B, int T, int nseg, const float* const* src, const int* batch_stride, const int* token_stride,
                        const int* width, const int* flip, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && T >= 1 && nseg >= 1 && nseg <= 3, "rows_concat: 1..3 segments");
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(out && src && batch_stride && token_stride && width && flip, "rows_concat: null pointer");
  ConcatArgs a{};
  a.nseg = nseg; a.B = B; a.T = T; a.W = 0;
  for (int k = 0; k < nseg; ++k) {
    VSP_REQUIRE(src[k] && width[k] >= 1, "rows_concat: bad segment %d", k);
    a.seg[k] = ConcatSeg{src[k], batch_stride[k], token_stride[k], width[k], flip[k]};
    a.W += width[k];
  }
  VSP_LAUNCH_1D(rows_concat_kernel, (int64_t)B * T * a.W, stream, out, a);
  return vsp::check_launch("rows_concat");
}

int vsp_add3_f32(float* out, const float* a, const float* b, const float* c, int64_t n, vsp_stream_t stream) {
  if (n <= 0) return VSP_OK;
  VSP_REQUIRE(out && a && b, "add3: null pointer");
  VSP_LAUNCH_1D(add3_kernel, n, stream, out, a, b, c, n);
  return vsp::check_launch("add3");
}

}  // extern "C"

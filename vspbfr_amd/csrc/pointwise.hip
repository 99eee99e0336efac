// 1x1 convolutions with at most four channels on ONE side, for gfx950: the ToRGB layers (Cin -> 3, reference
// models/RestoreNet.py:647-666) and the 3 -> 64 input layer.  These are pure HBM streams (2*Cin*Cout <= 512 FLOP per pixel
// against 4*(Cin+Cout) bytes): an MFMA tile kernel pads the tiny side to 16 and pays LDS staging and barriers for nothing
// (measured 2.0-2.3 TB/s on the conv kernel).  Here each thread owns 4 consecutive pixels, every plane is touched with
// 16-byte accesses issued in batches, the (style-scaled) weights sit in LDS and are read as wave-uniform broadcasts.
#include "vsp_common.h"
#include "vsp_bf16.h"

namespace {

using vsp::f32x4u;

__device__ __forceinline__ float lrelu2(float v, float bias, int act) {
  if (!act) return v;
  v += bias;
  return (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
}

// Cout <= 4:  y[b,co,p] = sum_ci x[b,ci,p] * (w[co,ci] * s[b,ci]) + bias[co] + res[b,co,p]
// TX = float or vsp::bf16_t: element type of the WIDE side (the Cin input planes); the 3-channel image side stays fp32
template <int CO, typename TX>
__global__ __launch_bounds__(256) void pw_few_out_kernel(float* __restrict__ y, const TX* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ in_scale,
                                                         const float* __restrict__ ch_bias, const float* __restrict__ res,
                                                         const float* __restrict__ up_src, const float* __restrict__ up_k,
                                                         int W, int Cin, int64_t HW) {
  extern __shared__ float wl[];  // [CO][Cin], style folded in
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < CO * Cin; i += 256) {
    const int ci = i % Cin;
    wl[i] = w[i] * (in_scale ? in_scale[(int64_t)b * Cin + ci] : 1.f);
  }
  __syncthreads();
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= HW) return;
  const bool full = p0 + 3 < HW;
  const TX* xb = x + (int64_t)b * Cin * HW + p0;
  f32x4u acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = f32x4u{0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 8;
  if ((HW & 3) == 0 && Cin % UN == 0) {
    // The shape every layer of the path has: whole quads, whole groups of 8 channels.  No branch around a load (a divergent `full`
    // test per load made the compiler wait for the loads in flight at every join: the 8 loads of a group left one by one), and the
    // next group is requested before the FMAs of the current one (clamped channel index: the last group re-reads itself).
    f32x4u v[UN], n[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = vsp::Elem<TX>::load4(xb + (int64_t)u * HW);
    for (int c0 = 0; c0 < Cin; c0 += UN) {
#pragma unroll
      for (int u = 0; u < UN; ++u) n[u] = vsp::Elem<TX>::load4(xb + (int64_t)min(c0 + UN + u, Cin - 1) * HW);
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        // the group's eight weights of this output channel: two 16-byte LDS reads (wave-uniform address) instead of eight 4-byte ones
        const float4 wa = *reinterpret_cast<const float4*>(wl + co * Cin + c0), wb = *reinterpret_cast<const float4*>(wl + co * Cin + c0 + 4);
        const float wv[UN] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(v[u][j], wv[u], acc[co][j]);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = n[u];
    }
  } else {
    for (int c0 = 0; c0 < Cin; c0 += UN) {
      f32x4u v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        v[u] = f32x4u{0.f, 0.f, 0.f, 0.f};
        if (c0 + u < Cin) {
          const TX* src = xb + (int64_t)(c0 + u) * HW;
          if (full) {
            v[u] = vsp::Elem<TX>::load4(src);
          } else {
            for (int j = 0; j < 4 && p0 + j < HW; ++j) v[u][j] = vsp::Elem<TX>::load1(src + j);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (c0 + u >= Cin) break;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const float wv = wl[co * Cin + c0 + u];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(v[u][j], wv, acc[co][j]);
        }
      }
    }
  }
  // up-sampled residual, in-row quads: this thread's row / column, source window origin and the four taps per source row
  const bool quad_row = up_src && (W & 3) == 0 && (HW & 3) == 0 && HW < ((int64_t)1 << 31);
  int uhh = 0, uhw = 0, usy = 0, usx = 0;
  float uk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (quad_row) {
    const unsigned p32 = (unsigned)p0, oy = p32 / (unsigned)W, ox = p32 - oy * (unsigned)W;
    uhh = (int)((unsigned)HW / (unsigned)W) >> 1;
    uhw = W >> 1;
    const int py = oy & 1;
    usy = (int)(oy >> 1) - 1 + py;       // source rows usy, usy + 1 carry the taps ky = py, py + 2
    usx = (int)(ox >> 1) - 1;            // source columns usx .. usx + 3
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // q = 2 (column parity) + tx: kx = parity + 2 tx
        const int ky = py + 2 * ty, kx = (q >> 1) + 2 * (q & 1);
        uk[ty][q] = up_k[(3 - ky) * 4 + (3 - kx)];
      }
  }
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    const int64_t o = ((int64_t)b * CO + co) * HW + p0;
    const float cb = ch_bias ? ch_bias[co] : 0.f;
    f32x4u r = f32x4u{0.f, 0.f, 0.f, 0.f};
    if (res) {
      if (full) r = *reinterpret_cast<const f32x4u*>(res + o);
      else for (int j = 0; j < 4 && p0 + j < HW; ++j) r[j] = res[o + j];
    }
    if (up_src && quad_row) {
      // residual = upfirdn2d(up_src, up_k (4x4), up = 2, pad = (2, 1)) evaluated in place (the Upsample of the RGB skip,
      // models/RestoreNet.py:100-118): zero insertion leaves one tap per parity and axis, 2 x 2 taps of the half-size map.  A lane's four
      // pixels lie in ONE row (W % 4 == 0): their sources are 2 rows x 4 columns of the half-size map, the taps depend on the row parity
      // only -- row / column once per thread in 32-bit arithmetic, 8 clamped loads per channel (the per-pixel form below: four 64-bit
      // divisions and 16 gathers per channel, as many instructions as the 64-channel main loop).  Same taps, same order of the four
      // fused multiply-adds per pixel as the per-pixel form: bit-identical.
      const float* sp = up_src + ((int64_t)b * CO + co) * uhh * uhw;
      float sv[2][4];
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        const int sy = usy + ty;
        const bool rok = sy >= 0 && sy < uhh;
        const float* srow = sp + min(max(sy, 0), uhh - 1) * uhw;
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) {
          const int sx = usx + cx;
          const float val = srow[min(max(sx, 0), uhw - 1)];
          sv[ty][cx] = (rok && sx >= 0 && sx < uhw) ? val : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {      // pixel ox + j: column parity j & 1, taps kx = (j & 1), (j & 1) + 2 on columns (j + 1) / 2, (j + 1) / 2 + 1
        float a = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            // (a source outside the map: the per-pixel form multiplies a clamped value by a zero tap; here the value itself is the zero --
            //  fma(k, 0, a) = fma(0, s, a) = a for finite operands)
            a = fmaf(uk[ty][(j & 1) * 2 + tx], sv[ty][((j + 1) >> 1) + tx], a);
          }
        r[j] += a;
      }
    } else if (up_src) {
      const int Hh = (int)(HW / W), hh = Hh >> 1, hw = W >> 1;
      const float* sp = up_src + ((int64_t)b * CO + co) * hh * hw;
#pragma unroll
      for (int j = 0; j < 4; ++j) {      // (clamped gathers, validity as a factor: no branch around the 16 loads of a quad)
        const int64_t pj = p0 + j < HW ? p0 + j : HW - 1;
        const int oy = (int)(pj / W), ox = (int)(pj - (int64_t)oy * W);
        float a = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty) {
          const int ky = (oy & 1) + 2 * ty, sy = (oy + ky - 2) >> 1;  // oy + ky even
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            const int kx = (ox & 1) + 2 * tx, sx = (ox + kx - 2) >> 1;
            const bool ok = sy >= 0 && sy < hh && sx >= 0 && sx < hw;
            const int syc = min(max(sy, 0), hh - 1), sxc = min(max(sx, 0), hw - 1);
            a = fmaf(ok ? up_k[(3 - ky) * 4 + (3 - kx)] : 0.f, sp[syc * hw + sxc], a);
          }
        }
        r[j] += a;
      }
    }
    f32x4u out;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = acc[co][j] + cb + r[j];
    if (full) *reinterpret_cast<f32x4u*>(y + o) = out;
    else for (int j = 0; j < 4 && p0 + j < HW; ++j) y[o + j] = out[j];
  }
}

// Cin <= 4:  y[b,co,p] = act2(act1(sum_ci x[b,ci,p] * w[co,ci] * s[b,ci] + ch_bias[co]))   (bias + leaky-ReLU(0.2)*sqrt2 each)
// TY = float or vsp::bf16_t: element type of the WIDE side (the Cout output planes)
template <int CI, typename TY>
__global__ __launch_bounds__(256) void pw_few_in_kernel(TY* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ in_scale,
                                                        const float* __restrict__ ch_bias, const float* __restrict__ bias1,
                                                        int act1, const float* __restrict__ bias2, int act2, int Cout,
                                                        int64_t HW) {
  extern __shared__ float wl[];  // [Cout][CI + 3]: weights, ch_bias, bias1, bias2
  const int b = blockIdx.y;
  for (int co = threadIdx.x; co < Cout; co += 256) {
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) wl[co * (CI + 3) + ci] = w[co * CI + ci] * (in_scale ? in_scale[(int64_t)b * CI + ci] : 1.f);
    wl[co * (CI + 3) + CI] = ch_bias ? ch_bias[co] : 0.f;
    wl[co * (CI + 3) + CI + 1] = (act1 && bias1) ? bias1[co] : 0.f;
    wl[co * (CI + 3) + CI + 2] = (act2 && bias2) ? bias2[co] : 0.f;
  }
  __syncthreads();
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= HW) return;
  const bool full = p0 + 3 < HW;
  f32x4u v[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    const float* src = x + ((int64_t)b * CI + ci) * HW + p0;
    v[ci] = f32x4u{0.f, 0.f, 0.f, 0.f};
    if (full) v[ci] = *reinterpret_cast<const f32x4u*>(src);
    else for (int j = 0; j < 4 && p0 + j < HW; ++j) v[ci][j] = src[j];
  }
  for (int co = 0; co < Cout; ++co) {
    const float* wr = wl + co * (CI + 3);
    f32x4u out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = wr[CI];
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) a = fmaf(v[ci][j], wr[ci], a);
      a = lrelu2(a, wr[CI + 1], act1);
      out[j] = lrelu2(a, wr[CI + 2], act2);
    }
    TY* dst = y + ((int64_t)b * Cout + co) * HW + p0;
    if (full) vsp::Elem<TY>::store4(dst, out);
    else for (int j = 0; j < 4 && p0 + j < HW; ++j) vsp::Elem<TY>::store1(dst + j, out[j]);
  }
}

}  // namespace

// W = element type of the wide side: x when Cout <= 4, y when Cin <= 4 (the other side and every operand are fp32)
template <typename TW>
static int pointwise_impl(void* y, const void* x, const float* w, const float* in_scale, const float* ch_bias,
                          const float* bias1, int act1, const float* bias2, int act2, const float* res,
                          const float* up_src, const float* up_kernel, int W, int B, int Cin, int Cout, int64_t HW,
                          vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && HW >= 0, "pointwise: bad dims");
  if (B == 0 || HW == 0) return VSP_OK;
  VSP_REQUIRE(y && x && w, "pointwise: null pointer");
  VSP_REQUIRE(Cin <= 4 || Cout <= 4, "pointwise: built for <= 4 channels on one side (got %d -> %d); use vsp_conv2d_f32", Cin, Cout);
  VSP_REQUIRE(B <= 65535, "pointwise: batch too large");
  const int64_t blocks = (HW + 1023) / 1024;
  VSP_REQUIRE(blocks < ((int64_t)1 << 31), "pointwise: plane too large");
  dim3 grid((unsigned)blocks, (unsigned)B);
  hipStream_t s = vsp::as_stream(stream);
  VSP_REQUIRE(!up_src || (up_kernel && Cout <= 4 && W >= 2 && W % 2 == 0 && HW % W == 0 && (HW / W) % 2 == 0),
              "pointwise: the up-sampled residual needs the few-output form, a 4x4 kernel and even H, W");
  if (Cout <= 4) {
    VSP_REQUIRE(!act1 && !act2, "pointwise: activations are only implemented on the few-input-channels form");
    VSP_REQUIRE(Cin <= 8192, "pointwise: too many input channels");
    const size_t lds = (size_t)Cout * Cin * sizeof(float);
    float* yf = static_cast<float*>(y);
    const TW* xw = static_cast<const TW*>(x);
    switch (Cout) {
      case 1: pw_few_out_kernel<1, TW><<<grid, 256, lds, s>>>(yf, xw, w, in_scale, ch_bias, res, up_src, up_kernel, W, Cin, HW); break;
      case 2: pw_few_out_kernel<2, TW><<<grid, 256, lds, s>>>(yf, xw, w, in_scale, ch_bias, res, up_src, up_kernel, W, Cin, HW); break;
      case 3: pw_few_out_kernel<3, TW><<<grid, 256, lds, s>>>(yf, xw, w, in_scale, ch_bias, res, up_src, up_kernel, W, Cin, HW); break;
      default: pw_few_out_kernel<4, TW><<<grid, 256, lds, s>>>(yf, xw, w, in_scale, ch_bias, res, up_src, up_kernel, W, Cin, HW); break;
    }
  } else {
    VSP_REQUIRE(!res, "pointwise: a residual is only implemented on the few-output-channels form");
    VSP_REQUIRE(Cout <= 4096, "pointwise: too many output channels");
    const size_t lds = (size_t)Cout * (Cin + 3) * sizeof(float);
    TW* yw = static_cast<TW*>(y);
    const float* xf = static_cast<const float*>(x);
    switch (Cin) {
      case 1: pw_few_in_kernel<1, TW><<<grid, 256, lds, s>>>(yw, xf, w, in_scale, ch_bias, bias1, act1, bias2, act2, Cout, HW); break;
      case 2: pw_few_in_kernel<2, TW><<<grid, 256, lds, s>>>(yw, xf, w, in_scale, ch_bias, bias1, act1, bias2, act2, Cout, HW); break;
      case 3: pw_few_in_kernel<3, TW><<<grid, 256, lds, s>>>(yw, xf, w, in_scale, ch_bias, bias1, act1, bias2, act2, Cout, HW); break;
      default: pw_few_in_kernel<4, TW><<<grid, 256, lds, s>>>(yw, xf, w, in_scale, ch_bias, bias1, act1, bias2, act2, Cout, HW); break;
    }
  }
  return vsp::check_launch("pointwise");
}

extern "C" int vsp_pointwise_f32(float* y, const float* x, const float* w, const float* in_scale, const float* ch_bias,
                                 const float* bias1, int act1, const float* bias2, int act2, const float* res,
                                 const float* up_src, const float* up_kernel, int W, int B, int Cin, int Cout, int64_t HW,
                                 vsp_stream_t stream) {
  return pointwise_impl<float>(y, x, w, in_scale, ch_bias, bias1, act1, bias2, act2, res, up_src, up_kernel, W, B, Cin, Cout, HW, stream);
}

extern "C" int vsp_pointwise_bf16(void* y, const void* x, const float* w, const float* in_scale, const float* ch_bias,
                                  const float* bias1, int act1, const float* bias2, int act2, const float* res,
                                  const float* up_src, const float* up_kernel, int W, int B, int Cin, int Cout, int64_t HW,
                                  vsp_stream_t stream) {
  return pointwise_impl<vsp::bf16_t>(y, x, w, in_scale, ch_bias, bias1, act1, bias2, act2, res, up_src, up_kernel, W, B, Cin, Cout, HW,
                                     stream);
}

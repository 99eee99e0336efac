// Weight re-layout for the convolution kernels, as kernels of their own: a TRAINED weight changes every iteration
// (restoration_train.py:123-131, 207-212), so its packed form -- and the packed form of its adjoint (channels exchanged, taps
// flipped: the data gradient) and the Winograd transform of both -- is rebuilt ~100 times per iteration.  As torch algebra that is
// ~20 small launches per weight (scale, transpose, flip, two permuted copies, an fp64 einsum with its own GEMMs and casts);
// here it is one launch for the packing and one for the transform.
//   vsp_pack_weight_f32      (Cout, Cin, KH, KW) -> Wp[g][tap][i][o]      (include/vspbfr_hip.h: layout of vsp_conv_params.w)
//   vsp_winograd_weight_f32  Wp -> U = G g G^T in the fragment order of vsp_conv2d_winograd_f32 (products and sums in fp64,
//                            rounded once -- the values the host-side float64 einsum produced)
#include "conv_kernel.h"
#include "vsp_common.h"

namespace {

// Straight packing: o = output channel inside its group (fastest in Wp), i = input channel (fastest of the two in the source).
// A 32 (o) x 32 (i) tile with up to 9 taps goes through LDS: the source is read as the contiguous (ci, tap) run of each output
// channel, the destination written as runs along o.
constexpr int kTapChunk = 9;
__global__ __launch_bounds__(256) void pack_weight_kernel(float* __restrict__ wp, const float* __restrict__ w, int cout_g, int cin, int T,
                                                          int flip, float scale) {
  __shared__ float tile[kTapChunk][32][33];
  const int g = blockIdx.z, i0 = blockIdx.y * 32, o0 = blockIdx.x * 32;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = w + (int64_t)g * cout_g * cin * T;
  float* dst = wp + (int64_t)g * T * cin * cout_g;
  const int ni = min(32, cin - i0);
  for (int t0 = 0; t0 < T; t0 += kTapChunk) {
    const int nt = min(kTapChunk, T - t0);
    for (int ol = wv; ol < 32; ol += 4) {                       // a wavefront per output channel: (ci, tap) run of ni * nt floats
      const int o = o0 + ol;
      if (o >= cout_g) break;
      const float* row = src + ((int64_t)o * cin + i0) * T + t0;
      for (int e = lane; e < ni * nt; e += 64) {
        const int il = e / nt, tl = e - il * nt;
        tile[tl][il][ol] = row[(int64_t)il * T + tl] * scale;
      }
    }
    __syncthreads();
    for (int r = ty; r < nt * 32; r += 8) {                     // rows (tap, i) of 32 output channels
      const int tl = r >> 5, il = r & 31;
      const int tap = flip ? T - 1 - (t0 + tl) : t0 + tl;
      if (il < ni && o0 + tx < cout_g) dst[((int64_t)tap * cin + i0 + il) * cout_g + o0 + tx] = tile[tl][il][tx];
    }
    __syncthreads();
  }
}

// Adjoint packing (one group): the source (Cout, Cin, T) is read as "output = ci, input = co": Wp[tap][i = co][o = ci].  The
// source's contiguous channel dimension is already the destination's: no transpose, threads run along ci.
__global__ __launch_bounds__(256) void pack_weight_adjoint_kernel(float* __restrict__ wp, const float* __restrict__ w, int cout, int cin,
                                                                  int T, int flip, float scale) {
  const int o = blockIdx.x * 64 + (threadIdx.x & 63);          // ci
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);           // co
  if (o >= cin || i >= cout) return;
  const float* src = w + ((int64_t)i * cin + o) * T;
  for (int tap = 0; tap < T; ++tap) wp[((int64_t)tap * cout + i) * cin + o] = src[flip ? T - 1 - tap : tap] * scale;
}

// One wavefront per (group, co tile, chunk): lane = (kq, lr) holds input channel 4 chunk + kq and, for each of the MB 16-channel
// blocks, output channel 16 MB tile + 16 mb + lr; it writes [wave 8][pp 2][lane][mb MB], position = 2 wave + pp = 4 a + b.
template <int MB>
__global__ __launch_bounds__(256) void winograd_weight_kernel(float* __restrict__ U, const float* __restrict__ wp, int cin, int cout_g,
                                                              int nct, int nch, int64_t units) {
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int lane = threadIdx.x & 63, kq = lane >> 4, lr = lane & 15;
  const int ch = (int)(unit % nch);
  const int t = (int)((unit / nch) % nct);
  const int g = (int)(unit / ((int64_t)nch * nct));
  const int ci = 4 * ch + kq;
  double u[MB][16];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int co = 16 * MB * t + 16 * mb + lr;
    double gk[3][3];
    const bool in = ci < cin && co < cout_g;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      gk[tap / 3][tap % 3] = in ? (double)wp[(((int64_t)g * 9 + tap) * cin + ci) * cout_g + co] : 0.0;
    // rows of G: (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1)
    double r[4][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      r[0][x] = gk[0][x];
      r[1][x] = 0.5 * (gk[0][x] + gk[1][x] + gk[2][x]);
      r[2][x] = 0.5 * (gk[0][x] - gk[1][x] + gk[2][x]);
      r[3][x] = gk[2][x];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      u[mb][4 * a + 0] = r[a][0];
      u[mb][4 * a + 1] = 0.5 * (r[a][0] + r[a][1] + r[a][2]);
      u[mb][4 * a + 2] = 0.5 * (r[a][0] - r[a][1] + r[a][2]);
      u[mb][4 * a + 3] = r[a][2];
    }
  }
  // [wave 8][pp 2][lane 64][mb MB]: a wave's A fragments of one position are ONE contiguous run (a 16-byte load per lane = 1 KiB of
  // consecutive memory; with [lane][pp][mb] each wave-wide load touched twice the cache lines it used)
  float* dst = U + unit * (int64_t)(1024 * MB) + (int64_t)lane * MB;
#pragma unroll
  for (int wv = 0; wv < 8; ++wv)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) dst[(int64_t)(wv * 2 + pp) * 64 * MB + mb] = (float)u[mb][2 * wv + pp];
}

}  // namespace

namespace {
// Per-image modulated weights of vsp_conv2d_bf16 (w_bstride): out[b][g][chunk][tap][octet][co][j] = bf16(wp[g][tap][16 chunk + 8 octet + j][co] *
// style[b][ci]), zero past Cin / cout_g -- the LDS-image order of conv_bf16.hip, one 16-byte unit per thread (neighbouring threads = neighbouring
// output channels: the eight strided reads of a thread coalesce across the wave).
__global__ __launch_bounds__(256) void modulate_weight_bf16_kernel(uint4* __restrict__ out, const float* __restrict__ wp, const float* __restrict__ style,
                                                                    int64_t style_bstride, int G, int cin, int cout_g, int nch, int co_pad, int units) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int b = blockIdx.y;
  int r = u;
  const int co = r % co_pad; r /= co_pad;
  const int oct = r & 1; r >>= 1;
  const int tap = r % 9; r /= 9;
  const int chunk = r % nch;
  const int g = r / nch;
  const float* sp = style + (int64_t)b * style_bstride;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = 16 * chunk + 8 * oct + j;
    const bool ok = ci < cin && co < cout_g;
    const int cic = ok ? ci : 0, coc = ok ? co : 0;
    const float w = wp[(((int64_t)g * 9 + tap) * cin + cic) * cout_g + coc];
    v[j] = ok ? w * sp[cic] : 0.f;
  }
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  uint4 o;
  o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
  o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
  o.z = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[4], v[5]}, bf16x2));
  o.w = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[6], v[7]}, bf16x2));
  out[(int64_t)b * units + u] = o;
}

}  // namespace

extern "C" {

int vsp_pack_weight_f32(float* wp, const float* w, int G, int cout_g, int cin, int KH, int KW, int adjoint, int flip, float scale,
                        vsp_stream_t stream) {
  VSP_REQUIRE(G >= 1 && cout_g >= 1 && cin >= 1 && KH >= 1 && KW >= 1, "pack_weight: bad dims");
  VSP_REQUIRE(wp && w, "pack_weight: null pointer");
  VSP_REQUIRE(!adjoint || G == 1, "pack_weight: the adjoint form is packed one group at a time");
  hipStream_t st = vsp::as_stream(stream);
  const int T = KH * KW;
  if (adjoint)
    pack_weight_adjoint_kernel<<<dim3((cin + 63) / 64, (cout_g + 3) / 4), 256, 0, st>>>(wp, w, cout_g, cin, T, flip, scale);
  else
    pack_weight_kernel<<<dim3((cout_g + 31) / 32, (cin + 31) / 32, G), 256, 0, st>>>(wp, w, cout_g, cin, T, flip, scale);
  return vsp::check_launch("pack_weight");
}

size_t vsp_winograd_weight_floats(int G, int cin, int cout_g) {
  const int ck = vspconv::wino_chunk(), mb = vspconv::wino_mbw(cout_g);
  const int64_t nch = (cin + ck - 1) / ck, nct = (cout_g + 16 * mb - 1) / (16 * mb);
  return (size_t)(G * nct * nch * 1024 * mb);
}

int vsp_winograd_weight_f32(float* U, const float* wp, int G, int cin, int cout_g, vsp_stream_t stream) {
  VSP_REQUIRE(G >= 1 && cin >= 1 && cout_g >= 1, "winograd_weight: bad dims");
  VSP_REQUIRE(U && wp, "winograd_weight: null pointer");
  VSP_REQUIRE(vspconv::wino_chunk() == 4, "winograd_weight: the fragment layout assumes one k-step per chunk");
  const int mb = vspconv::wino_mbw(cout_g);
  const int nch = (cin + 3) / 4, nct = (cout_g + 16 * mb - 1) / (16 * mb);
  const int64_t units = (int64_t)G * nct * nch;
  const int blocks = (int)((units + 3) / 4);
  hipStream_t st = vsp::as_stream(stream);
  switch (mb) {
    case 1: winograd_weight_kernel<1><<<blocks, 256, 0, st>>>(U, wp, cin, cout_g, nct, nch, units); break;
    case 2: winograd_weight_kernel<2><<<blocks, 256, 0, st>>>(U, wp, cin, cout_g, nct, nch, units); break;
    case 4: winograd_weight_kernel<4><<<blocks, 256, 0, st>>>(U, wp, cin, cout_g, nct, nch, units); break;
    default: return vsp::fail(VSP_EINVAL, "winograd_weight: unexpected block width");
  }
  return vsp::check_launch("winograd_weight");
}

size_t vsp_modulate_weight_bf16_bytes(int G, int cin, int cout_g) {
  const int64_t nch = (cin + 15) / 16, co_pad = (cout_g + 31) / 32 * 32;
  return (size_t)(G * nch * 9 * 2 * co_pad * 16);
}

int vsp_modulate_weight_bf16(uint16_t* out, const float* wp, const float* style, int B, int64_t style_bstride, int G, int cin, int cout_g,
                             vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && G >= 1 && cin >= 1 && cout_g >= 1, "modulate_weight: bad dims");
  if (B == 0) return VSP_OK;
  VSP_REQUIRE(out && wp && style && vsp::aligned16(out), "modulate_weight: null / unaligned pointer");
  const int nch = (cin + 15) / 16, co_pad = (cout_g + 31) / 32 * 32;
  const int64_t units = (int64_t)G * nch * 9 * 2 * co_pad;   // 16-byte units per image
  VSP_REQUIRE(units < ((int64_t)1 << 31) && B <= 65535, "modulate_weight: weight too large");
  modulate_weight_bf16_kernel<<<dim3((unsigned)((units + 255) / 256), (unsigned)B), 256, 0, vsp::as_stream(stream)>>>(
      reinterpret_cast<uint4*>(out), wp, style, style_bstride, G, cin, cout_g, nch, co_pad, (int)units);
  return vsp::check_launch("modulate_weight");
}

}  // extern "C"

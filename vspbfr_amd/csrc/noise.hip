// Keyed random tensors for gfx950: every random draw of one batch in ONE launch, invariant to how the batch is sharded.
//
// The reference draws from torch's global RNG stream, one `randn` per NoiseInjection layer (models/RestoreNet.py:564-569,
// e4e/models/stylegan2/model.py:287-292), one for x_T (ldm/ddpm.py:423) and one for z (restoration_test.py:77-82): 48
// launches per batch whose values depend on the draw order and on the batch split.  Here a value is a pure function of
// (seed, global image index, segment id, element index):
//   Philox4x32-10 (Salmon et al., SC'11; the generator behind curand / torch.cuda as well), key = (seed_lo, seed_hi),
//   counter = (element / 4, segment id, image_lo, image_hi) -> four 32-bit words -> four values:
//     normal : Box-Muller on pairs, u = (word >> 8 + 0.5) * 2^-24 in (0, 1): (r cos t, r sin t), r = sqrt(-2 ln u0), t = 2 pi u1
//     uniform: 2 u - 1 in (-1, 1)
// so rank r of W ranks that owns global images [lo, hi) produces exactly the tensors a single GPU would.
// Output layout: the segments back to back, segment s = [B][n_s] floats (a dense (B,1,H,W) / (B,18,512) / (B,512) tensor).
// HBM-bound: 4 B written per value, one float4 store per Philox call; oracle/device_rng.py restates it in numpy.
#include "vsp_common.h"

namespace {

struct SegTable {
  uint32_t quad_end[VSP_NOISE_MAX_SEGMENTS];  // running sum of B * ceil(n_s / 4)
  uint32_t n[VSP_NOISE_MAX_SEGMENTS];         // elements per image
  uint32_t id[VSP_NOISE_MAX_SEGMENTS];        // segment id that enters the counter
  uint64_t off[VSP_NOISE_MAX_SEGMENTS];       // float offset of the segment in `out`
  int nseg;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t w) { return ((float)(w >> 8) + 0.5f) * 5.9604644775390625e-08f; }

__global__ __launch_bounds__(256) void keyed_fill_kernel(float* __restrict__ out, SegTable tab, uint32_t total_quads,
                                                          uint32_t seed_lo, uint32_t seed_hi, uint64_t image0,
                                                          const int64_t* __restrict__ image0_dev, int dist) {
  if (image0_dev) image0 += (uint64_t)*image0_dev;  // graph replays: the base index lives in device memory
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total_quads; q += stride) {
    int s = 0;
    while (q >= tab.quad_end[s]) ++s;  // <= 64 entries in scalar registers; a wave almost always sits inside one segment
    const uint32_t base = s ? tab.quad_end[s - 1] : 0u;
    const uint32_t n = tab.n[s], qpi = (n + 3u) >> 2;  // quads per image
    const uint32_t local = q - base, b = local / qpi, e4 = local - b * qpi;
    const uint64_t img = image0 + b;
    uint32_t w[4];
    philox4x32_10(e4, tab.id[s], (uint32_t)img, (uint32_t)(img >> 32), seed_lo, seed_hi, w);
    float v[4];
    if (dist == 0) {
      const float r0 = sqrtf(-2.0f * logf(u01(w[0]))), r1 = sqrtf(-2.0f * logf(u01(w[2])));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u01(w[1]), &s0, &c0);
      sincosf(6.283185307179586f * u01(w[3]), &s1, &c1);
      v[0] = r0 * c0; v[1] = r0 * s0; v[2] = r1 * c1; v[3] = r1 * s1;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = 2.0f * u01(w[i]) - 1.0f;
    }
    float* dst = out + tab.off[s] + (uint64_t)b * n + (uint64_t)e4 * 4u;
    if ((n & 3u) == 0) {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      const uint32_t left = n - e4 * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((uint32_t)i < left) dst[i] = v[i];
    }
  }
}

}  // namespace

extern "C" int vsp_keyed_fill_f32(float* out, int B, const int64_t* seg_elems, const int32_t* seg_ids, int n_seg, uint64_t seed,
                                   int64_t image_index0, const int64_t* image_index0_dev, int dist, vsp_stream_t stream) {
  VSP_REQUIRE(B >= 0 && n_seg >= 0, "keyed_fill: negative batch / segment count");
  if (B == 0 || n_seg == 0) return VSP_OK;
  VSP_REQUIRE(out && seg_elems && seg_ids, "keyed_fill: null pointer");
  VSP_REQUIRE(n_seg <= VSP_NOISE_MAX_SEGMENTS, "keyed_fill: %d segments (max %d)", n_seg, VSP_NOISE_MAX_SEGMENTS);
  VSP_REQUIRE(dist == 0 || dist == 1, "keyed_fill: dist must be 0 (normal) or 1 (uniform(-1,1))");
  VSP_REQUIRE(image_index0 >= 0, "keyed_fill: negative image index");
  VSP_REQUIRE(vsp::aligned16(out), "keyed_fill: output must be 16-byte aligned");
  SegTable tab;
  uint64_t quads = 0, off = 0;
  for (int s = 0; s < n_seg; ++s) {
    VSP_REQUIRE(seg_elems[s] > 0 && seg_elems[s] < (1ll << 31), "keyed_fill: segment %d has %lld elements", s,
                (long long)seg_elems[s]);
    // a segment whose per-image size is not a multiple of 4 keeps float4 stores off (scalar tail), but its successor must
    // still start 16-byte aligned for the vector path: require that only of segments that use it
    VSP_REQUIRE((seg_elems[s] % 4 != 0) || (off % 4 == 0), "keyed_fill: segment %d is not 16-byte aligned (a preceding segment has an odd size)", s);
    quads += (uint64_t)B * (uint64_t)((seg_elems[s] + 3) / 4);
    VSP_REQUIRE(quads < (1ull << 32), "keyed_fill: too many elements for one launch");
    tab.quad_end[s] = (uint32_t)quads;
    tab.n[s] = (uint32_t)seg_elems[s];
    tab.id[s] = (uint32_t)seg_ids[s];
    tab.off[s] = off;
    off += (uint64_t)B * (uint64_t)seg_elems[s];
  }
  tab.nseg = n_seg;
  uint64_t blocks = (quads + 255) / 256;
  if (blocks > (uint64_t)vsp::kMaxStreamBlocks) blocks = vsp::kMaxStreamBlocks;
  keyed_fill_kernel<<<(int)blocks, 256, 0, vsp::as_stream(stream)>>>(out, tab, (uint32_t)quads, (uint32_t)seed,
                                                                     (uint32_t)(seed >> 32), (uint64_t)image_index0, image_index0_dev,
                                                                     dist);
  return vsp::check_launch("keyed_fill");
}

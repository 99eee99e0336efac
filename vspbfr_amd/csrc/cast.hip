// f32 <-> bf16 tensor conversion for gfx950 (boundary of the bf16-activation configuration: tensors that cross between a
// bf16-I/O kernel and an fp32 one -- the 16^2 and smaller maps, a few MB per batch).  HBM stream: 6 B per element.
#include "vsp_common.h"
#include "vsp_bf16.h"

namespace {

template <typename TO, typename TI>
__global__ __launch_bounds__(256) void cast_kernel(TO* __restrict__ out, const TI* __restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    vsp::Elem<TO>::store4(out + 4 * i, vsp::Elem<TI>::load4(x + 4 * i));
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    vsp::Elem<TO>::store1(out + i, vsp::Elem<TI>::load1(x + i));
}

template <typename TO, typename TI>
int launch_cast(TO* out, const TI* x, int64_t n, vsp_stream_t stream, const char* what) {
  VSP_REQUIRE(n >= 0, "%s: negative element count", what);
  if (n == 0) return VSP_OK;
  VSP_REQUIRE(out && x, "%s: null pointer", what);
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > vsp::kMaxStreamBlocks) blocks = vsp::kMaxStreamBlocks;
  cast_kernel<TO, TI><<<(int)blocks, 256, 0, vsp::as_stream(stream)>>>(out, x, n);
  return vsp::check_launch(what);
}

}  // namespace

extern "C" int vsp_convert_f32_to_bf16(uint16_t* out, const float* x, int64_t n, vsp_stream_t stream) {
  return launch_cast<vsp::bf16_t, float>(out, x, n, stream, "convert_f32_to_bf16");
}

extern "C" int vsp_convert_bf16_to_f32(float* out, const uint16_t* x, int64_t n, vsp_stream_t stream) {
  return launch_cast<float, vsp::bf16_t>(out, x, n, stream, "convert_bf16_to_f32");
}

/*
 * vspbfr_hip.h -- C ABI of libvspbfr_hip.so: the MI355X (gfx950) device kernels behind the VSPBFR
 * restoration inference path.
 *
 * Boundary contract
 * -----------------
 *  - plain C: device pointers, sizes, scalars and an opaque stream handle (a hipStream_t); no torch types.
 *  - every entry point ENQUEUES work on `stream` and returns immediately (no host sync), exactly like the
 *    reference's two native ops which launch on at::cuda::getCurrentCUDAStream()
 *    (reference op/fused_bias_act_kernel.cu:73, op/upfirdn2d_kernel.cu:215).
 *  - tensors are dense, row-major, fp32 (VSP_F32).  Pointers are borrowed; outputs are caller-allocated
 *    (the reference allocates with torch::empty_like / at::empty inside the op -- the Python mirror in
 *    vspbfr_amd/op does that allocation so that the ABI stays torch-free).
 *  - return value: 0 on success, a negative VSP_E* code otherwise; vsp_last_error() returns a
 *    thread-local human-readable message.  The Python mirror raises RuntimeError (what TORCH_CHECK raises
 *    in the reference: op/fused_bias_act.cpp:10-16, op/upfirdn2d.cpp:9-15).
 *
 * What each entry point replaces in the reference is cited on the declaration.
 */
#ifndef VSPBFR_HIP_H
#define VSPBFR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSP_ABI_VERSION 4

#define VSP_OK 0
#define VSP_EINVAL (-1)   /* bad argument (shape / null pointer / unsupported combination) */
#define VSP_ELAUNCH (-2)  /* hipLaunchKernel / HIP runtime failure */
#define VSP_ENOTSUP (-3)  /* valid request that this build has no kernel for */

typedef void* vsp_stream_t; /* hipStream_t; NULL = the default stream */

int vsp_abi_version(void);
const char* vsp_last_error(void);
/* number of HIP devices visible, or a negative VSP_E* code (used by the loader's self-check). */
int vsp_device_count(void);
/* sizeof of an ABI struct (0 = vsp_fir_epilogue, 1 = vsp_conv_params, 2 = vsp_gemm_params,
 * 3 = vsp_tacc_block, 4 = vsp_tacc_chain_params, 5 = vsp_conv_wgrad_params): lets a binding in
 * another language check its own struct layout when it loads the library. */
int vsp_struct_size(int which);

/* ------------------------------------------------------------------------------------------------
 * fused bias + activation  -- replaces `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)`
 * (reference op/fused_bias_act.cpp:18-31, kernel op/fused_bias_act_kernel.cu:19-65).
 *   out[i] = f(x[i] + bias[(i / step_b) % size_b]) * scale
 *   act*10+grad: 10,11 -> f = identity; 30 -> leaky-relu(alpha); 31 -> (ref[i] > 0 ? v : v*alpha);
 *   12, 32 -> 0.  bias == NULL means "no bias" (reference: empty tensor), ref likewise.
 * n = number of elements; step_b = product of dims after dim 1 (1 for 2-D input).
 * ---------------------------------------------------------------------------------------------- */
int vsp_fused_bias_act_f32(float* out, const float* x, const float* bias, const float* ref, int64_t n,
                           int step_b, int size_b, int act, int grad, float alpha, float scale,
                           vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d -- replaces `upfirdn2d_op.upfirdn2d(input[major,H,W,minor], kernel[kh,kw], up_x, up_y,
 * down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)` (reference op/upfirdn2d.cpp:17-31, kernels
 * op/upfirdn2d_kernel.cu:49-207, CPU definition op/upfirdn2d.py:365-406).
 * out dims: out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y (likewise out_w); the caller
 * allocates out[major, out_h, out_w, minor].  Negative pads crop, as in the reference.
 *
 * The optional epilogue (all pointers may be NULL) fuses what the reference runs as separate ops right
 * after the blur of an up-sampling StyledConv (models/RestoreNet.py:599-603, e4e stylegan2/model.py:337-341):
 *   v = fir(x)
 *   v = v * plane_scale[plane]                 (per (b,c) demodulation coefficient)
 *   v = v + noise[b, oy, ox] * noise_w[0]      (NoiseInjection; noise is [B,1,out_h,out_w])
 *   v = lrelu(v + act_bias[c], slope) * gain   (FusedLeakyReLU; only if act != 0)
 *   v = v + res1[...] + res2[...]              (same layout as out)
 * `channels` gives c = plane % channels, b = plane / channels (minor must be 1 when an epilogue is used).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vsp_fir_epilogue {
  const float* plane_scale; /* [major] or NULL */
  const float* noise;       /* [major/channels, out_h, out_w] or NULL */
  const float* noise_w;     /* device scalar, required iff noise != NULL */
  const float* act_bias;    /* [channels] or NULL (treated as 0) */
  const void* res1;         /* [major, out_h, out_w] or NULL; element type of out (float, or bf16 with vsp_upfirdn2d_bf16) */
  const void* res2;         /* likewise */
  int channels;             /* C (>=1) */
  int act;                  /* 0 none, 1 leaky-relu */
  float slope;              /* 0.2 */
  float gain;               /* sqrt(2) */
  int flags;                /* VSP_FIR_SEPARABLE: kernel[ky][kx] == kernel[ky][0] * kernel[0][kx] / kernel[0][0] (an outer product, as every
                             * blur of the path is: make_kernel([1,3,3,1]), models/RestoreNet.py:38-48) -- the caller's promise; the blur
                             * kernels may then run a row pass and a column pass (8 multiply-adds per output instead of 16); the result
                             * differs from the 2-D form by rounding only.  0 = general taps. */
} vsp_fir_epilogue;
#define VSP_FIR_SEPARABLE 1

int vsp_upfirdn2d_f32(float* out, const float* x, const float* kernel, int major, int in_h, int in_w,
                      int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                      int pad_x1, int pad_y0, int pad_y1, const vsp_fir_epilogue* epi /* may be NULL */,
                      vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * conv2d as an fp32-MFMA implicit GEMM -- replaces the conv2d_gradfix.conv2d / conv_transpose2d / F.conv2d
 * calls of the path (reference op/conv2d_gradfix.py:22-92, models/RestoreNet.py:125-131,373-416,510-553,
 * e4e/models/stylegan2/model.py:116,259-274, e4e/models/encoders/helpers.py:98-113) together with the
 * element-wise work the reference wraps around them (modulation, demodulation, bias, noise, activation,
 * residual adds, eval-mode BatchNorm, PReLU).
 *
 *   acc[b,co,oy,ox] = sum_{ci,ky,kx} Wp[g][ky*KW+kx][ci][co_g] *
 *                     xin(b, ci, oy*stride_y + ky*dil[g] - pad[g], ox*stride_x + kx*dil[g] - pad[g])
 *   xin(b,ci,iy,ix) = 0 outside the image, else x[b,ci,iy,ix] * in_scale[b*in_scale_bstride + ci] + in_shift[ci]
 * with g = co / cout_g (output-channel groups: either groups that differ only in dilation/padding and share the
 * input -- the four dilated branches of SMART_layer in one launch -- or true convolution groups with their own
 * input-channel slice, see x_group_stride), then per output element, in this order:
 *   v = acc * out_scale[b*Cout + co]                     (demodulation, models/RestoreNet.py:376-379)
 *   v = v * ch_scale[co] + ch_bias[co]                   (conv bias / folded eval BatchNorm)
 *   act1: v = lrelu(v + bias1[co], slope1) * gain1       (FusedLeakyReLU of `fusion`, RestoreNet.py:1176-1177)
 *   v = v + noise[b,oy,ox] * noise_w[0]                  (NoiseInjection, RestoreNet.py:564-569)
 *   act2: 1: v = lrelu(v + bias2[co], slope2) * gain2    (FusedLeakyReLU `activate`)
 *         2: v = v >= 0 ? v : v * prelu[co]              (nn.PReLU)
 *   v = v + res1[b,co,oy,ox] + res2[b,co,oy,ox]
 *   y[b, y_coff + co, oy*osy + ooy, ox*osx + oox] = v    (y has y_ch channels, y_h x y_w pixels)
 * The output stride/offset (osy, osx, ooy, oox) lets a stride-2 transposed conv run as four sub-pixel phase
 * convolutions writing one (2H+1)x(2W+1) tensor (same MACs as conv_transpose2d).
 * Wp is the launch's weight, packed by the host once at model-load time:
 *   Wp[g][tap][ci][co_g], co_g contiguous (see hip_ops.pack_weight in vspbfr_amd/hip_ops.py).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vsp_conv_params {
  const float* x;  /* [B, Cin, H, W] */
  const float* w;  /* packed [G][KH*KW][Cin][cout_g] */
  float* y;        /* [B, y_ch, y_h, y_w] */
  int B, Cin, H, W;
  int G, cout_g;   /* Cout = G * cout_g */
  int OH, OW;      /* number of output positions computed per image */
  int KH, KW;
  int stride_y, stride_x;
  int dil[4];      /* per group */
  int pad_y[4];    /* per group */
  int pad_x[4];    /* per group */
  int y_ch, y_coff, y_h, y_w;
  int osy, osx, ooy, oox;
  /* prologue */
  const float* in_scale; /* NULL or [.., Cin]; grouped input (x_group_stride > 0): [.., x_ch], group g reads [g * x_group_stride, + Cin) */
  int in_scale_bstride;  /* Cin for per-sample styles, 0 for per-channel constants */
  const float* in_shift; /* NULL or [Cin] */
  /* epilogue */
  const float* out_scale; /* NULL or [B, Cout] */
  const float* ch_scale;  /* NULL or [Cout] */
  const float* ch_bias;   /* NULL or [Cout] */
  int act1;               /* 0 none, 1 lrelu */
  const float* bias1;     /* NULL or [Cout] */
  float slope1, gain1;
  const float* noise;     /* NULL or [B, OH, OW] */
  const float* noise_w;   /* device scalar */
  int act2;               /* 0 none, 1 lrelu, 2 prelu */
  const float* bias2;     /* NULL or [Cout] */
  const float* prelu;     /* [Cout] when act2 == 2 */
  float slope2, gain2;
  const float* res1;      /* NULL or same layout as the y region written (see res_* below) */
  const float* res2;
  int res_ch, res_coff;   /* residual tensors are [B, res_ch, y_h, y_w], read at channel res_coff + co */
  int tile_hint;          /* 0 = let the library choose; n > 0 = configuration n-1, an error if it does not fit (tests / tuning);
                             n < 0 = prefer configuration -n-1, fall back to the library's choice if it does not fit */
  /* grouped input (true grouped convolution, e.g. the 18 map2style heads of the e4e encoder run as one launch per stage):
   * x has x_ch channels per image (0 = Cin) and group g reads channels [g*x_group_stride, g*x_group_stride + Cin).
   * x_group_stride = 0: all groups read the same Cin channels (the dilation groups of SMART_layer).  With G > 4 every
   * group uses dil[0] / pad_y[0] / pad_x[0]. */
  int x_ch, x_group_stride;
  /* transposed = 1: y = conv_transpose2d(xin, W, stride 2, padding 0) for a 3x3 kernel in ONE launch (all four
   * sub-pixel phases; reference models/RestoreNet.py:530-532, e4e stylegan2/model.py:259).  w is the ordinary packed
   * [1][9][Cin][Cout] weight (W[co][ci][ky][kx], not flipped); y must be [B, y_ch, 2H+1, 2W+1]; KH = KW = 3, G = 1;
   * stride / dilation / padding / OH / OW / os* / oo* fields are ignored; prologue scaling and the epilogue's
   * per-channel terms apply, noise and residuals are not available (they follow the blur in the reference). */
  int transposed;
  /* io_bf16 = 1 (vsp_conv2d_bf16 only): x, y, res1 and res2 hold bf16 elements (raw 16-bit words, same dense NCHW shapes) --
   * the bf16-ACTIVATION configuration of BASELINE configs[2]: 2 B per element through HBM instead of 4.  Every other operand
   * (noise, scales, biases) and the accumulation stay fp32; y is rounded to nearest even once, after the epilogue chain. */
  int io_bf16;
  /* dil_by_input_quarter = 1 (vsp_conv2d_f32 only): the DATA GRADIENT of four dilated branches over one input (the SMART / LargeConv
   * layers, reference models/RestoreNet.py:205-215, 742-750) in one pass:  y = sum_q conv(x[:, q Cin/4 : (q+1) Cin/4], W_q, dil[q]),
   * pad = dil.  G = 4 and x_group_stride = 0 as in the forward dilation-group call, but the four "groups" are the four blocks of
   * cout_g output channels of ONE convolution over all Cin input channels (w = [4][9][Cin][cout_g], cout_g <= 16 per launch row),
   * and dil[q] / pad[q] belong to input-channel quarter q.  3x3, stride 1, Cin a multiple of 16; served by the pipelined kernels
   * only (VSP_ENOTSUP when no configuration fits). */
  int dil_by_input_quarter;
  /* w_bstride != 0 (vsp_conv2d_bf16 with io_bf16 = 1 only; every other entry requires 0): PER-IMAGE weights -- image b reads the weight set at
   * (char*)w + b * w_bstride (bytes, a multiple of 16), each in the layout of vsp_conv2d_bf16.  This is the reference's own "fused" modulated
   * convolution (models/RestoreNet.py:381-416: weight * style per sample): vsp_modulate_weight_bf16 builds bf16(W * style[b]) once per layer and
   * batch, in_scale / in_shift must then be NULL, and the kernel stages bf16 pixels into LDS with no arithmetic at all. */
  int64_t w_bstride;
} vsp_conv_params;

int vsp_conv2d_f32(const vsp_conv_params* p, vsp_stream_t stream);
/* Number of tile configurations compiled in, and a description of configuration i ("64x256 ..."). */
/* The same convolution contract for 3x3 / stride 1 / padding = dilation layers (G = 1, or up to four dilation groups over
 * one shared input: x_group_stride = 0) through Winograd F(2x2,3x3); a dilation d is served as d*d polyphase sub-images.
 * `w` must hold the TRANSFORMED weights U = G g G^T (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]) in the kernel's
 * fragment order.  With CK = vsp_conv2d_winograd_chunk() input channels per chunk, MB = vsp_conv2d_winograd_mbw(cout_g)
 * 16-channel blocks per workgroup, Cin zero-padded to a multiple of CK and cout_g to a multiple of 16*MB:
 *     w[((((((g * ntile + tile) * nchunk + chunk) * 8 + wave) * 2 + pp) * 64 + lane) * MB + mb]
 *         = U_g[position 2*wave + pp][ci = CK*chunk + (lane >> 4)][co = 16*MB*tile + 16*mb + (lane & 15)]
 * (a wave owns two Winograd positions; its A fragments of one position and chunk are one contiguous run: a wave-wide load reads
 * consecutive memory).
 * Every prologue / epilogue field of vsp_conv_params keeps its meaning.  tile_hint names the kernel FORM (tests / tuning): 0 = the
 * library chooses; 1 = task list (conv_wino.hip), 2 = row owner (conv_wino_ro.hip / conv_wino_rod.hip), 3 = register-resident U with
 * polyphase staging (conv_wino_rs.hip: Cin <= 64, W % 4 == 0, no in_shift -- the 64 -> 4 x 16 dilation groups of the SMART layers,
 * reference models/RestoreNet.py:179-244); a named form that does not serve the launch returns VSP_ENOTSUP.  16 multiplies per 2x2
 * output tile instead of 36; fp32 error ~1e-6 relative on top of the direct kernel's summation-order noise. */
int vsp_conv2d_winograd_f32(const vsp_conv_params* p, vsp_stream_t stream);
int vsp_conv2d_winograd_chunk(void);
int vsp_conv2d_winograd_mbw(int cout_g);
/* The same contract through Winograd F(4x4,3x3) for the DEEP stride-1 3x3 layers (one group, dilation 1, padding 1, no in_shift;
 * Cin a multiple of 4, H and W multiples of 4, dense 16-byte aligned output / noise / residual planes: VSP_ENOTSUP otherwise --
 * reference layers models/RestoreNet.py:213-244,410-416, e4e/models/stylegan2/model.py:268-276 at 256 / 512 channels): 36 multiplies
 * per 4x4 output tile instead of 144.  Two launches: the input transform V = B^T d B (in_scale applied) into `work`
 * (vsp_conv2d_winograd4_work_floats(p) floats, fragment order of the GEMM), then a barrier-free GEMM with both operands fetched from
 * L2 straight into MFMA fragments and the fused epilogue of vsp_conv2d_f32.  Interpolation points 0, +-3/4, +-3/2, infinity: all
 * transform constants are dyadic, fp32 error = the direct kernel's (rms 1.5e-6 at unit scale).  `w` = vsp_winograd4_weight_f32:
 *     w[((((tile * nchunk + chunk) * 12 + wave) * 3 + q) * 64 + lane) * 4 + mb]
 *         = U[position 3*wave + q][ci = 4*chunk + (lane >> 4)][co = 64*tile + 16*mb + (lane & 15)],   U = G g G^T
 * (one wave-wide 16-byte load = 1 KiB of consecutive memory). */
int vsp_conv2d_winograd4_f32(const vsp_conv_params* p, float* work, size_t work_floats, vsp_stream_t stream);
/* Winograd F(4x4,3x3), FUSED form (round 5, conv_wino4f.hip): the layer class of vsp_conv2d_winograd4_f32 (3x3, stride 1, fp32, style scale
 * `in_scale` but no `in_shift`, dense same-size output) without a work buffer -- the input transform runs in registers in the MFMA's
 * fragment layout, the transformed input never exists in memory.  Serves Cin % 8 == 0 (<= 512), W >= 16 and
 *   - one group with padding = dilation in {1, 2, 4, 8}, H and W multiples of 4 x dilation: dilation 1 = the 32 / 64-channel layers on maps
 *     >= 128^2 (reference e4e/models/stylegan2/model.py:268-276, models/RestoreNet.py:421-555), where the two-kernel form's V round trip
 *     costs more than it saves; a dilated layer runs on its polyphase sub-images (tiles of 4 x 4 outputs `dilation` apart, windows read
 *     from an LDS-staged region);
 *   - G = 2..4 dilation groups over ONE shared input (x_group_stride = 0; the SMART branches models/RestoreNet.py:179-244), each group's
 *     geometry as above, in one launch: group g writes channels y_coff + g * cout_g ..., and every per-channel operand (out_scale, ch_scale,
 *     biases, prelu) and the residuals are indexed g * cout_g + channel -- the convention of vsp_conv2d_f32 for G > 1.
 * `w` = vsp_winograd4f_weight_f32's output, per group and the groups one after the other: U = G g G^T (the same 36 values per (ci, co) as
 * vsp_winograd4_weight_f32) in the order
 * [co / 32][ci / 4][position pair 18][lane 64][4]: lane = 16 (ci % 4) + (co % 16), element e = position 2 pp + (e >> 1), co block (e & 1).
 * VSP_ENOTSUP for a launch it does not serve (the caller falls back to vsp_conv2d_winograd_f32). */
size_t vsp_winograd4f_weight_floats(int cin, int cout);
int vsp_winograd4f_weight_f32(float* U, const float* wp, int cin, int cout, vsp_stream_t stream);
int vsp_conv2d_winograd4f_f32(const vsp_conv_params* params, vsp_stream_t stream);

size_t vsp_conv2d_winograd4_work_floats(const vsp_conv_params* p);
size_t vsp_winograd4_weight_floats(int cin, int cout);
int vsp_winograd4_weight_f32(float* U, const float* wp, int cin, int cout, vsp_stream_t stream);
/* 1x1 convolution, stride 1, on small maps as one GEMM (the bottleneck 1x1 layers of the identity loss network at 7x7 / 4x4,
 * Loss/id_loss.py:13,27-41): y[b][co][p] = act(sum_ci w[co][ci] x[b][ci][p] + bias[co]); w row-major (Cout, Cin) = the OIHW
 * weight as it lies; Cin a multiple of 16; bias may be NULL; act: 0 none, 1 leaky relu (slope) times gain. */
int vsp_conv1x1_small_f32(float* y, const float* w, const float* x, const float* bias, int B, int Cout, int Cin, int P, int act,
                          float slope, float gain, vsp_stream_t stream);
/* Weight re-layout on the device (a trained weight is re-packed every iteration: restoration_train.py:123-131, 207-212 step the
 * optimizers between passes).  vsp_pack_weight_f32: OIHW weight (G*cout_g, cin, KH, KW) -> the layout of vsp_conv_params.w,
 * wp[g][tap][ci][co_g] = scale * w[g*cout_g + co_g][ci][tap'] with tap' = KH*KW-1-tap when `flip`.  `adjoint` (G = 1) packs the
 * weight of the data gradient instead: the roles of the channels exchanged, wp[tap][i = co][o = ci] = scale * w[co][ci][tap']
 * (replaces weight.transpose(0,1).flip(2,3) + packing in the reference's conv2d_gradfix.py:152-190 backward).
 * vsp_winograd_weight_f32: packed weight -> the transformed weight of vsp_conv2d_winograd_f32 in its fragment order
 * (vsp_winograd_weight_floats() floats); sums in fp64, rounded once. */
int vsp_pack_weight_f32(float* wp, const float* w, int G, int cout_g, int cin, int KH, int KW, int adjoint, int flip, float scale,
                        vsp_stream_t stream);
size_t vsp_winograd_weight_floats(int G, int cin, int cout_g);
int vsp_winograd_weight_f32(float* U, const float* wp, int G, int cin, int cout_g, vsp_stream_t stream);
int vsp_conv2d_num_configs(void);
const char* vsp_conv2d_config_name(int i);

/* The same convolution contract on the BF16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the "bf16 kernels"
 * configuration of the path (BASELINE configs[2]).  Served layers, all 3x3:
 *   - stride 1, padding = dilation: G = 1, up to four dilation groups over one shared input, or true groups (x_group_stride);
 *   - stride 2, dilation 1, padding 0 or 1: G = 1 or true groups (StyledConv_down after its blur, IR-SE down-convs, the
 *     batched e4e style heads);
 *   - transposed = 1: the stride-2 transposed conv of the up-sampling StyledConvs in one pass (fields as for vsp_conv2d_f32).
 * x, y and every prologue / epilogue operand stay fp32: the kernel scales the input by the style in fp32, rounds to bf16 (RNE)
 * while staging and runs the fp32 epilogue chain of vsp_conv2d_f32 on the fp32 accumulators.  `w` must hold the weights
 * rounded to bf16 in the kernel's LDS image order; with Cin zero-padded to a multiple of 16, co_pad = cout_g rounded up to 32
 * and nchunk = ceil(Cin / 16):
 *     w[(((((g * nchunk + chunk) * 9 + tap) * 2 + octet) * co_pad + co) * 8 + j]      (uint16 bf16 bit patterns)
 *         = bf16( W_g[tap][ci = 16*chunk + 8*octet + j][co] )
 * (one 16-byte row per (tap, channel octet, co): what one lane feeds v_mfma as its A fragment; a workgroup copies its rows
 * global -> LDS with global_load_lds_dwordx4).  tile_hint selects the tile variant (0 = automatic, see conv_bf16.hip).
 * Error vs the fp32 kernels: 2^-9 relative per operand, ~2.5e-3 of the output range per layer on random data -- this entry is
 * a throughput configuration, not the parity path. */
int vsp_conv2d_bf16(const vsp_conv_params* p, vsp_stream_t stream);

/* The split-precision form of vsp_conv2d_bf16 ("bf16x3") for the stride-1 and stride-2 layers: both operands are carried as
 * hi + lo bf16 pairs (hi = bf16(v), lo = bf16(v - hi): 16 significant bits) and every product is evaluated as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi into the same fp32 accumulator -- three MFMAs per tile pair on the bf16 pipe, relative
 * error ~2^-16 per product instead of 2^-8: fp32-grade results for layers that the fp32 matrix pipe (1/16 of the bf16 rate)
 * bounds.  Same contract and operands as vsp_conv2d_bf16; `w` holds BOTH weight parts, chunk by chunk:
 *     w[((((((g * nchunk + chunk) * 2 + part) * 9 + tap) * 2 + octet) * co_pad + co) * 8 + j]
 *         part 0 = bf16(W), part 1 = bf16(W - float(bf16(W)))   (hip_ops.bf16x3_weight). */
int vsp_conv2d_bf16x3(const vsp_conv_params* p, vsp_stream_t stream);

/* The "row-vector K" form of vsp_conv2d_bf16 for the low-channel, large-map stride-1 layers of the bf16-activation
 * configuration (32 -> 32 at 1024^2, 64 -> 64 at 512^2, 128 -> 128 at 256^2 ...; conv_bf16_rv.hip): a lane's eight k-values are two
 * channels x four consecutive pixels of an input row (three horizontal taps + one zero weight), so the NCHW bf16 image is
 * staged as it lies in HBM and the output leaves the accumulators as packed pixel pairs.  Serves io_bf16 = 1, G = 1, 3x3,
 * stride 1, dilation = padding = 1, Cin % 8 == 0, Cout % 32 == 0, W % 64 == 0, 16-byte aligned x, images below 1 GiB; returns
 * VSP_ENOTSUP for any other launch (the caller falls back to vsp_conv2d_bf16).  Same operands and epilogue as vsp_conv2d_bf16;
 * `w` in this kernel's order (hip_ops.bf16rv_weight), nchunk = Cin / 8:
 *     w[(((((chunk * 3 + ky) * 2 + quad) * 2 + half) * Cout + co) * 8 + e]              (uint16 bf16 bit patterns)
 *         = bf16( W[tap = 3*ky + kx][ci = 8*chunk + 4*quad + 2*half + (e >> 2)][co] ),  kx = e & 3;  0 for kx = 3.
 * tile_hint: 0 = automatic, 1 = 32-channel tiles (16 rows x 64 pixels), 2 = 64-channel tiles (8 rows x 64 pixels). */
int vsp_conv2d_bf16rv(const vsp_conv_params* p, vsp_stream_t stream);
/* The dilation groups of a SMART branch launch with bf16 activations (io_bf16 = 1; reference models/RestoreNet.py:179-244, 270-418): up to four
 * groups (dilation = padding = 1, 2, 4, 8) of at most 16 output channels over ONE shared input of at most 64 channels (a multiple of 8), W a
 * multiple of 8, no in_shift.  `w` is the PACKED FP32 weight of vsp_conv2d_f32 (the kernel rounds w * in_scale to bf16 once per image and keeps
 * it in registers); patch channel-last and polyphase in LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulation and epilogue chain, bf16 output.
 * VSP_ENOTSUP when the launch is not one it serves (conv_bf16_dg.hip). */
int vsp_conv2d_bf16dg(const vsp_conv_params* p, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Strided, batched small GEMM on fp32 MFMA -- replaces F.linear / torch.matmul of the path
 * (EqualLinear: models/RestoreNet.py:161-171; TACC_block / spatial_attention: models/CodeDiffuser.py:35-47,
 * 86-116).
 *   C[z][m][n] = epi( alpha * sum_k A[z][m][k] * B[z][n][k] )
 *   element addresses: A + z*a_zs + m*a_ms + k*a_ks   (same for B with n), C + z*c_zs + m*c_ms + n
 *   epi(v): v += bias[z*bias_zs + n] * bias_scale (bias may be NULL); act==1: v = lrelu(v, slope) * gain;
 *           act==2: v = sigmoid(v)
 * ---------------------------------------------------------------------------------------------- */
typedef struct vsp_gemm_params {
  const float* A;
  const float* Bm;
  float* C;
  int Z, M, N, K;
  int64_t a_zs, a_ms, a_ks;
  int64_t b_zs, b_ns, b_ks;
  int64_t c_zs, c_ms;
  float alpha;
  const float* bias;
  float bias_scale;
  int act;
  float slope, gain;
  int64_t bias_zs; /* bias stride between batches (0 = one bias vector for all z) */
} vsp_gemm_params;

int vsp_gemm_f32(const vsp_gemm_params* p, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise helpers of Code_diffuser / the style MLPs (reference models/CodeDiffuser.py:7-12,41,101,31,73;
 * models/RestoreNet.py:24-29).  x and out are [rows, cols] row-major unless stated otherwise.
 * ---------------------------------------------------------------------------------------------- */
/* out[z, r, c] = x[z,r,c] * rsqrt(mean_r(x[z,:,c]^2) + eps): PixelNorm over dim=1 of a [Z,R,C] tensor. */
int vsp_pixelnorm_dim1_f32(float* out, const float* x, int Z, int R, int C, float eps, vsp_stream_t stream);
/* LayerNorm over the last dim (biased variance, eps inside sqrt); gamma/beta may be NULL;
 * optional fused pre-add: normalises (x + add).  post: 0 none, 1 lrelu(slope)*gain. */
int vsp_layernorm_f32(float* out, const float* x, const float* add, const float* gamma, const float* beta,
                      int rows, int cols, float eps, int post, float slope, float gain, vsp_stream_t stream);
/* softmax over the last dim of [rows, cols]. */
int vsp_softmax_lastdim_f32(float* out, const float* x, int rows, int cols, vsp_stream_t stream);
/* softmax over dim=1 of [Z, R, C] (column-wise within each z). */
int vsp_softmax_dim1_f32(float* out, const float* x, int Z, int R, int C, vsp_stream_t stream);
/* out = h * (1 + gamma) + beta (TACC_block tail, models/CodeDiffuser.py:114). */
int vsp_film_f32(float* out, const float* h, const float* gamma, const float* beta, int64_t n,
                 vsp_stream_t stream);
/* out = a * x + b * y with a, b read from device arrays at index idx (DDPM posterior mean,
 * ldm/ddpm.py:348-352: coef1[t]*x0 + coef2[t]*x_t). */
int vsp_axpby_idx_f32(float* out, const float* x, const float* y, const float* a, const float* b, int idx,
                      int64_t n, vsp_stream_t stream);
/* demodulation coefficients: out[b, co] = rsqrt(wscale^2 * sum_ci style[b,ci]^2 * wsq[co,ci] + eps)
 * (models/RestoreNet.py:376-379 with the tap sum hoisted: wsq[co,ci] = sum_taps W[co,ci,:,:]^2). */
int vsp_demod_f32(float* out, const float* style, const float* wsq, int B, int Cin, int Cout, float wscale,
                  float eps, vsp_stream_t stream);
/* Every style modulation of a network for one latent in two launches (round 5): per modulated layer l the modulation vector
 * mod_l[b, ci] = alpha_l * sum_k src[b * bstride + src_off_l + k] * w_l[ci, k] + bias_l[ci] * bias_scale_l   (EqualLinear,
 * models/RestoreNet.py:142-171, evaluated at models/RestoreNet.py:211,467 once per layer) and, where wsq_l is given, the demodulation
 * coefficients demod_l[b, co] = rsqrt(wscale2_l * sum_ci mod_l[b, ci]^2 * wsq_l[co, ci] + eps) (models/RestoreNet.py:376-379).  `table`
 * is a DEVICE array of L entries; src_off is the element offset of the layer's style row inside the latent tensor (row b of the view
 * latent[:, i] starts at src + b * bstride + src_off).  K a multiple of 256 (>= 512), B <= 16, 16-byte aligned rows.  Results are
 * bit-identical to vsp_gemm_f32 (few-row form) followed by vsp_demod_f32 per layer. */
typedef struct {
  const float* w;      /* (cin, K) EqualLinear weight */
  const float* bias;   /* (cin) or NULL */
  const float* wsq;    /* (cout, cin) sum of squared taps, or NULL: no demodulation */
  float* mod;          /* out (B, cin) */
  float* demod;        /* out (B, cout) or NULL */
  int64_t src_off;
  int cin, cout;
  float alpha, bias_scale, wscale2;
  int pad_;
} vsp_style_layer;
int vsp_style_plan_f32(const vsp_style_layer* table, int L, const float* src, int B, int64_t bstride, int K, int max_cin, int max_cout,
                       float eps, vsp_stream_t stream);
/* The same coefficients under autograd (training rows; reference models/RestoreNet.py:376-379 differentiated by torch): forward from
 * the weight w[Cout, Cin, K] (K taps) -- wsq[co, ci] = sum_k w^2 is written for the backward -- and the gradient of a loss through
 * `out` with respect to the style and the weight:  t = -0.5 wscale^2 out^3 g;  dstyle[b, ci] = 2 style[b, ci] sum_co t[b, co] wsq[co, ci];
 * dw[co, ci, k] = 2 w[co, ci, k] sum_b t[b, co] style[b, ci]^2 (overwritten, not accumulated; either output may be NULL).  B <= 16. */
int vsp_demod_weight_f32(float* out, float* wsq, const float* style, const float* w, int B, int Cin, int Cout, int K, float wscale,
                         float eps, vsp_stream_t stream);
int vsp_demod_weight_bwd_f32(float* dstyle, float* dw, const float* g, const float* out, const float* style, const float* wsq,
                             const float* w, int B, int Cin, int Cout, int K, float wscale, vsp_stream_t stream);
/* ... with a row pitch for g AND out (column windows of wider [B, g_stride] tensors: the four branches of a SMART layer share one
 * concatenated demodulation tensor) and `accumulate` = 1: dstyle / dw are ADDED to what the buffers hold (the convolution's own style and
 * weight gradients), so that one tensor per operand leaves the layer's backward. */
int vsp_demod_weight_bwd_acc_f32(float* dstyle, float* dw, const float* g, int g_stride, const float* out, const float* style,
                                 const float* wsq, const float* w, int B, int Cin, int Cout, int K, float wscale, int accumulate,
                                 vsp_stream_t stream);
/* 2x2 mean pooling of [planes, 2*OH, 2*OW] (F.interpolate bilinear 512->256 with align_corners=False is
 * exactly this, Loss/e4e_embedding.py:97; AdaptiveAvgPool2d 1024->512, e4e/models/psp.py:246). */
int vsp_avgpool2x2_f32(float* out, const float* x, int64_t planes, int OH, int OW, vsp_stream_t stream);
/* bilinear resize, align_corners=True, then add: out = resize(x -> [OH,OW]) + y
 * (_upsample_add, e4e/models/encoders/helpers.py:123-140). */
int vsp_upsample_add_f32(float* out, const float* x, const float* y, int64_t planes, int IH, int IW, int OH,
                         int OW, vsp_stream_t stream);
/* global average pool: out[plane] = mean(x[plane, :]) (SEModule, helpers.py:57-73). */
int vsp_plane_mean_f32(float* out, const float* x, int64_t planes, int hw, vsp_stream_t stream);
/* out = x * gate[plane] + y (SE excitation + shortcut add, helpers.py:72,112). y may be NULL. */
int vsp_scale_add_f32(float* out, const float* x, const float* gate, const float* y, int64_t planes, int hw,
                      vsp_stream_t stream);
/* strided gather of y[b,c,::s,::s] (MaxPool2d(1, stride) shortcut, helpers.py:101). */
int vsp_subsample_f32(float* out, const float* x, int64_t planes, int IH, int IW, int s, vsp_stream_t stream);
/* 8-bit image quantiser of torchvision.utils.save_image(normalize=True, value_range=(lo, hi)) fused with the
 * NCHW -> NHWC transpose a PNG encoder wants (reference restoration_test.py:138-157; torchvision 0.13 utils.py):
 *   t = (clamp(x, lo, hi) - lo) / max(hi - lo, 1e-5);  out[b,y,x,c] = (uint8) clamp(t * 255 + 0.5, 0, 255) */
int vsp_quantize_u8_nhwc(uint8_t* out, const float* x, int B, int C, int H, int W, float lo, float hi,
                         vsp_stream_t stream);
/* F.interpolate(x, (OH, OW), mode="bilinear", align_corners=False) on `planes` = B*C planes (the resize to 256^2 in front of the
 * e4e encoder, reference Loss/e4e_embedding.py:91-100, for inputs that are not 512^2 -- there it is the 2x2 mean above). */
int vsp_resize_bilinear_f32(float* out, const float* x, int64_t planes, int IH, int IW, int OH, int OW, vsp_stream_t stream);
/* Latent plumbing of a batch as single launches (the reference runs it as repeat / cat / flip / add on (B, 18, 512) tensors):
 *   e4e_codes:   codes[b,t,:] = heads[t,b,:] + (t > 0 ? heads[0,b,:] : 0) + latent_avg[t,:]  -- the 18 map2style outputs (T, B, D) to
 *                W+ codes (B, T, D) (psp_encoders.py:188-199 w0 + delta_i, psp.py:159-165 latent_avg; latent_avg may be NULL)
 *   rows_concat: out[b,t,:] = [seg_0 | seg_1 | seg_2], seg_k = src_k[b*batch_stride_k + tt*token_stride_k + 0..width_k), tt = t or
 *                T-1-t (flip_k); token_stride 0 broadcasts one row over the tokens (Restoration_net's latent = [W+ code | mapped z],
 *                its flipped copy and the decoder styles [latent | x_global], models/RestoreNet.py:1000-1025).  Host arrays. */
int vsp_e4e_codes_f32(float* out, const float* heads, const float* latent_avg, int B, int T, int D, vsp_stream_t stream);
int vsp_rows_concat_f32(float* out, int B, int T, int nseg, const float* const* src, const int* batch_stride, const int* token_stride,
                        const int* width, const int* flip, vsp_stream_t stream);
/* out = a + b + c (c may be NULL). */
int vsp_add3_f32(float* out, const float* a, const float* b, const float* c, int64_t n, vsp_stream_t stream);

/* 1x1 convolution with at most 4 channels on one side, as an HBM stream (no MFMA tile, no padding of the tiny side):
 *   Cout <= 4 (ToRGB: modulated 1x1 conv without demodulation + bias + FIR-upsampled skip, models/RestoreNet.py:647-666):
 *       y[b,co,p] = sum_ci x[b,ci,p] * w[co,ci] * in_scale[b,ci] + ch_bias[co] + res[b,co,p] + up(up_src)[b,co,p]
 *       up(.) = upfirdn2d(up_src (B,Cout,H/2,W/2), up_kernel (4x4), up = 2, pad = (2,1)), the `Upsample` of the RGB skip
 *       (models/RestoreNet.py:100-118), evaluated inside the kernel: the skip image is never materialised at full size
 *   Cin <= 4 (the 3 -> 64 input layer, two FusedLeakyReLUs in the epilogue, models/RestoreNet.py:725-787 with k = 1):
 *       y[b,co,p] = act2(act1(sum_ci x[b,ci,p] * w[co,ci] * in_scale[b,ci] + ch_bias[co])),  act_i(v) = lrelu(v + bias_i[co], 0.2) * sqrt2
 * x (B,Cin,HW), y / res (B,Cout,HW) dense, W = row length; in_scale, ch_bias, bias1, bias2, res, up_src may be NULL. */
int vsp_pointwise_f32(float* y, const float* x, const float* w, const float* in_scale, const float* ch_bias,
                      const float* bias1, int act1, const float* bias2, int act2, const float* res, const float* up_src,
                      const float* up_kernel, int W, int B, int Cin, int Cout, int64_t HW, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused TACC_block step of Code_diffuser (reference models/CodeDiffuser.py:86-116 and :35-47), 18 tokens x 512
 * channels.  P is the [B*18, ldp] projection buffer of one block: columns [k_off, v_off, q2_off, v2_off) + 512 hold
 * K, V, q (spatial_attention) and v (spatial_attention) of pixelnorm(x).  The Linear(513 -> 512) layers that read the
 * condition c = [embd, t/T] are supplied split: e* = embd @ W[:, :512]^T (step independent, [B*18, 512]) and the last
 * weight column w* (w*[d * stride]); tfrac = t/T is shared by the batch (the sampler's case).
 *   scores:    score[b,i,j] = softmax_j( K[b,i,:] . (eQ[b,j,:] + tfrac*wq) / sqrt(18) )
 *   chan_attn: A = softmax over rows of ((ek[b] + tfrac*wk)^T q2 / sqrt(512));  t[b] = v2[b] @ A
 *   tail:      (token attention folded in: score as above) y = LN(score @ V + LN(t)) * (1 + gamma) + beta;
 *              if xold: y = c1[idx]*y + c2[idx]*xold (DDPM posterior
 *              mean, ldm/ddpm.py:348-352);  pn = PixelNorm over the 18 tokens of y (may be NULL)
 *   head_pre:  out[s,m,:] = lrelu(LN(e[m,:] + (s/t_div)*wcol) * ln_w + ln_b, 0.2) * sqrt(2) for s = 0..S-1
 *              (first half of the gamma_/beta_ heads for every step of the chain at once)
 * ---------------------------------------------------------------------------------------------- */
int vsp_tacc_scores_f32(float* score, const float* P, int ldp, int k_off, const float* eQ, const float* wq,
                        int wq_stride, float tfrac, int B, int n_tok, int dim, vsp_stream_t stream);
int vsp_tacc_chan_attn_f32(float* t, const float* P, int ldp, int q2_off, int v2_off, const float* ek,
                           const float* wk, int wk_stride, float tfrac, int B, int n_tok, int dim,
                           vsp_stream_t stream);
int vsp_tacc_tail_f32(float* y, float* pn, const float* P, int ldp, int k_off, int v_off, const float* eQ,
                      const float* wq /* contiguous [512] */, float tfrac, const float* t, const float* gamma,
                      const float* beta, const float* xold, const float* c1, const float* c2, int idx, int B, int n_tok,
                      int dim, vsp_stream_t stream);
int vsp_tacc_head_pre_f32(float* out, const float* e, const float* wcol, int w_stride, const float* ln_w,
                          const float* ln_b, int S, int M, int dim, float t_div, vsp_stream_t stream);

/* The whole sampler chain of Code_diffuser behind one call (replaces the Python loops ldm/ddpm.py:400-429 p_sample_loop /
 * ldm/ddim.py:130-180 ddim_sampling around models/CodeDiffuser.py:118-140): for s in 0..n_steps-1, with t = step[s]:
 *   x0 = denoiser(x, cond, t)  (n_blocks TACC blocks);   x <- c1[k] * x0 + c2[k] * x,  k = coef_idx[s] (or t when NULL)
 * (c1 == c2 == NULL: x <- x0).  `blocks`, `step`, `coef_idx` are HOST arrays; every pointer inside is a device pointer.
 * Per block: wcat = [Wk; Wv; Wq2; Wv2] rows (2048 x 512), eQ / ek = condition halves of the two Linear(513) layers
 * ((B*18) x 512), wq / wk = their contiguous t-columns (512), gamma / beta = FiLM heads of steps 0..head_steps-1
 * (head_steps x (B*18) x 512, from vsp_tacc_head_pre_f32 + vsp_gemm_f32).  x is updated in place; `work` holds
 * vsp_tacc_chain_work_floats(B) floats of scratch.  3 launches per block and step, nothing synchronises. */
typedef struct vsp_tacc_block {
  const float* wcat;
  const float* eQ;
  const float* ek;
  const float* wq;
  const float* wk;
  const float* gamma;
  const float* beta;
  /* optional (NULL: read wcat): the same [4D, D] matrix in MFMA FRAGMENT order -- [n / 16][k / 16][lane = 16 (k % 16 / 4) + n % 16][k % 4]
   * -- so that a wavefront's 16-byte-per-lane load is 1 KiB of consecutive memory (read row-major, the 16 lanes of a quarter wave hit
   * 16 different cache lines for 16 bytes each: the projection was bound by the texture-address path, not by its MFMAs) */
  const float* wcat_frag;
} vsp_tacc_block;

typedef struct vsp_tacc_chain_params {
  int B, n_tok, dim, n_blocks;
  const vsp_tacc_block* blocks;
  float* x;
  float* work;
  size_t work_floats;
  int n_steps;
  const int* step;
  const int* coef_idx;
  const float* c1;
  const float* c2;
  float t_div;     /* t enters the condition as t / t_div (Code_diffuser.max_period) */
  int head_steps;
} vsp_tacc_chain_params;

size_t vsp_tacc_chain_work_floats(int B);
int vsp_tacc_chain_f32(const vsp_tacc_chain_params* p, vsp_stream_t stream);
/* (A persistent single-launch form of this chain -- clusters of 1 ... 16 workgroups per image, two cluster barriers per block --
 * was built and parity-tested in round 3 and measured SLOWER than the three launches per block (46 vs 27.5 us per block at batch 8):
 * retired to the branch `experiments/persistent-chain` in round 4; DESIGN section 4 keeps the accounting.) */

/* ------------------------------------------------------------------------------------------------
 * bf16 activations in HBM (BASELINE configs[2] "bf16 kernels"; the fp32 path above is the parity path).  A bf16 tensor is
 * the raw 16-bit words of its fp32 twin rounded to nearest even, same dense NCHW shape.  Arithmetic stays fp32.
 *   vsp_convert_*      tensor conversion at the boundary between bf16-I/O kernels and fp32 ones (maps of 16^2 and smaller)
 *   vsp_upfirdn2d_bf16 the `Blur` form of vsp_upfirdn2d_f32 (up = down = 1, minor = 1, 2x2 / 3x3 / 4x4 taps, out_w >= 16) with
 *                      x, out and the epilogue's res1 / res2 in bf16 (kernel taps, noise, scales, biases fp32); VSP_ENOTSUP otherwise
 *   vsp_pointwise_bf16 vsp_pointwise_f32 with the WIDE side in bf16: x when Cout <= 4 (ToRGB reads bf16 features, writes the
 *                      fp32 image), y when Cin <= 4 (the 3 -> 64 input layer reads the fp32 image, writes bf16 features)
 * ---------------------------------------------------------------------------------------------- */
int vsp_convert_f32_to_bf16(uint16_t* out, const float* x, int64_t n, vsp_stream_t stream);
int vsp_convert_bf16_to_f32(float* out, const uint16_t* x, int64_t n, vsp_stream_t stream);
int vsp_upfirdn2d_bf16(uint16_t* out, const uint16_t* x, const float* kernel, int major, int in_h, int in_w, int minor,
                       int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                       int pad_y1, const vsp_fir_epilogue* epilogue, vsp_stream_t stream);
/* Per-image modulated weights for vsp_conv2d_bf16 (w_bstride): out[b] = bf16(wp * style[b][ci]) in the kernel's LDS-image order
 * [group][chunk][tap][octet 2][co_pad][8] (ci = 16 chunk + 8 octet + j; Cin zero-padded to 16, cout_g to 32), wp = the packed fp32 weights
 * [G][9][Cin][cout_g] of vsp_conv2d_f32, style = (B, Cin) rows `style_bstride` floats apart (shared by the groups of a dilation-group launch).
 * One rounding per weight (the kernel's own path rounds W and x * style separately).  Returns the byte size of one image's set through
 * vsp_modulate_weight_bf16_bytes.  Reference: models/RestoreNet.py:381-383 (weight = scale * weight * style), without the demodulation,
 * which stays an fp32 (B, Cout) vector in the epilogue. */
size_t vsp_modulate_weight_bf16_bytes(int G, int cin, int cout_g);
int vsp_modulate_weight_bf16(uint16_t* out, const float* wp, const float* style, int B, int64_t style_bstride, int G, int cin, int cout_g,
                             vsp_stream_t stream);
int vsp_pointwise_bf16(void* y, const void* x, const float* w, const float* in_scale, const float* ch_bias,
                       const float* bias1, int act1, const float* bias2, int act2, const float* res, const float* up_src,
                       const float* up_kernel, int W, int B, int Cin, int Cout, int64_t HW, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution backward (the training step; reference op/conv2d_gradfix.py:104-227, which on current torch is
 * autograd of F.conv2d / F.conv_transpose2d, :78-92).
 *   data gradient   = the forward kernels: conv with the flipped, channel-transposed weight (stride 1), the one-pass
 *                     transposed conv (adjoint of the stride-2 conv) and the stride-2 conv (adjoint of the transposed conv)
 *   weight gradient = vsp_conv2d_wgrad_f32:
 *     dw[g][co][ci][ky][kx] = sum_{b,oy,ox} dy[b, g*Cout_g+co, oy, ox] * dy_scale[b, .] * x[b, g*Cin_g+ci, oy*stride+ky*dil-pad, ox*stride+kx*dil-pad] * x_scale[b, .]
 *     dw is (G*Cout_g, Cin_g, KH, KW) dense (torch's weight layout), overwritten; x (B, G*Cin_g, H, W), dy (B, G*Cout_g, OH, OW);
 *     x_scale (B, G*Cin_g) / dy_scale (B, G*Cout_g) may be NULL (the per-sample style / demodulation vectors of a
 *     modulated layer in its modulate-input / demodulate-output form); 3x3 or 1x1, stride 1 or 2.  G = B with batch 1 is
 *     the reference's groups=batch form (models/RestoreNet.py:373-383).  A transposed conv's weight gradient is the same
 *     call with x and dy exchanged (and the result read as (ci, co)).
 *   vsp_plane_dot_f32: out[p] = sum_i a[p,i]*b[p,i] -- the gradients of the two per-sample scale vectors.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vsp_conv_wgrad_params {
  const float* x;
  const float* dy;
  float* dw;
  const float* x_scale;
  const float* dy_scale;
  int B, Cin_g, H, W, G, Cout_g, OH, OW, KH, KW, stride, dil, pad;
  /* ABI 3 (all zero = the plain call above): channel windows into larger tensors, a shared input with per-group geometry (the four
   * dilated SMART branches: G = 4, x_shared = 1, per_group_geometry = 1, dil_g = pad_g = {1, 2, 4, 8}), accumulation into dw */
  int x_ch, x_coff;        /* channels of the x tensor (0: G*Cin_g, or Cin_g when shared) and first channel used */
  int dy_ch, dy_coff;      /* channels of the dy tensor (0: G*Cout_g) and first channel used */
  int x_shared;            /* 1: every group reads the same Cin_g input channels */
  int per_group_geometry;  /* 1: group g uses dil_g[g] / pad_g[g] (G <= 4) instead of dil / pad */
  int dil_g[4], pad_g[4];
  int accumulate;          /* 1: add to dw instead of overwriting it */
  float* work;             /* optional workspace: the split-K partial sums go to private copies of dw (plain stores) and a second
                            * kernel adds them up, instead of fp32 atomics into dw (25 % of the kernel's time at 512 channels);  */
  size_t work_floats;      /* floats in `work`: at least one copy of dw, vsp_conv2d_wgrad_work_floats() for the full split */
  float dw_scale;          /* 0 = 1: the sum is multiplied by it before it is stored / added (the layer's equalized-lr factor, so that dw is
                            * the gradient of the PARAMETER: reference models/RestoreNet.py:131, 150 `weight * self.scale`) */
} vsp_conv_wgrad_params;
/* workspace size (floats) with which vsp_conv2d_wgrad_f32 runs its preferred split for these parameters */
size_t vsp_conv2d_wgrad_work_floats(const vsp_conv_wgrad_params* p);
int vsp_conv2d_wgrad_f32(const vsp_conv_wgrad_params* p, vsp_stream_t stream);
int vsp_plane_dot_f32(float* out, const float* a, const float* b, int64_t planes, int64_t n, vsp_stream_t stream);
/* out[p] = sum_i a[p,i]*b[p,i] and, in the same pass, a[p,:] *= scale[p] (d/ds and d/dx of a modulated layer from d/d(x s)) */
int vsp_plane_dot_scale_f32(float* out, float* a, const float* b, const float* scale, int64_t planes, int64_t n, vsp_stream_t stream);
/* NoiseInjection + FusedLeakyReLU of a styled layer as one stream (training forward; reference models/RestoreNet.py:558-569 then
 * op/fused_act.py:199-233):  y[b,c,p] = lrelu(x[b,c,p] + noise_w[0] * noise[b,p] + bias[c], slope) * gain;  bias may be NULL.
 * vsp_noise_dot_f32: out[0] = sum_{b,c,p} gx[b,c,p] * noise[b,p] -- the gradient of the scalar noise weight. */
int vsp_noise_bias_act_f32(float* y, const float* x, const float* noise, const float* noise_w, const float* bias, int B, int C,
                           int64_t hw, float slope, float gain, vsp_stream_t stream);
int vsp_noise_dot_f32(float* out, const float* gx, const float* noise, int B, int C, int64_t hw, vsp_stream_t stream);
/* Backward of the tail of a SMART layer (reference models/RestoreNet.py:220-244: FusedLeakyReLU(bias1) -> NoiseInjection ->
 * FusedLeakyReLU(bias2); the forward is the fusion conv's epilogue: act1 / bias1 / noise / act2 / bias2 of vsp_conv_params) from the
 * final output y alone: g1 = gradient entering the conv, db1[c], db2[c], dnw[0] (the three are zeroed here, then accumulated). */
int vsp_smart_tail_bwd_f32(float* g1, float* db1, float* db2, float* dnw, const float* g, const float* y, const float* noise,
                           const float* noise_w, const float* bias2, int B, int C, int64_t hw, float slope, float gain,
                           vsp_stream_t stream);
/* out[c] = sum over b and the plane of x[b, c, :] (x (B, C, hw) dense): bias gradients of the training step */
int vsp_channel_sum_f32(float* out, const float* x, int B, int C, int64_t hw, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ADA augmentation primitives (reference non_leaking.py:857-934).
 *   vsp_affine_sample_f32      out = F.grid_sample(x, F.affine_grid(theta, (B,C,OH,OW), align_corners=False), "bilinear", "zeros",
 *                              align_corners=False); theta (B, 2, 3) fp32 on the device; the grid is never materialised
 *   vsp_affine_sample_bwd_f32  gx = adjoint of the above w.r.t. x applied to gout (gx (B,C,IH,IW) is overwritten)
 *   vsp_color_affine_f32       y[b,c,p] = sum_k M[b,c,k] x[b,k,p] + t[b,c] on 3-channel images (apply_color, :910-918); M (B,3,3),
 *                              t (B,3) or NULL
 * ---------------------------------------------------------------------------------------------- */
int vsp_affine_sample_f32(float* out, const float* x, const float* theta, int B, int C, int IH, int IW, int OH, int OW,
                          vsp_stream_t stream);
int vsp_affine_sample_bwd_f32(float* gx, const float* gout, const float* theta, int B, int C, int IH, int IW, int OH, int OW,
                              vsp_stream_t stream);
int vsp_color_affine_f32(float* y, const float* x, const float* M, const float* t, int B, int64_t HW, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Operators of the training losses (BASELINE configs[4]: LPIPS-VGG + ArcFace identity; reference restoration_train.py:236-245).
 *   vsp_maxpool2d_f32 / _bwd   F.max_pool2d(x, k, s, p) in floor mode on `planes` = B*C planes (torchvision 0.13 vgg16.features
 *                              2x2/2, resnet101 3x3/2 pad 1); the backward recomputes each window's arg-max (first maximum in
 *                              row-major order, ATen's rule) from x; dx is overwritten
 *   vsp_lpips_layer_f32        out[b] = mean_p sum_c w[c] (f0/(|f0|_c + 1e-10) - f1/(|f1|_c + 1e-10))^2  -- normalize_tensor,
 *                              squared difference, the 1x1 `lin` layer and spatial_average of one LPIPS level in one stream
 *                              (my_lpips/networks_basic.py:73-83, my_lpips/__init__.py:44-46); f0, f1 (B,C,HW), w (C), out (B)
 *   vsp_lpips_layer_bwd_f32    df1 = gout[b] * d out[b] / d f1   (the reference calls forward(target, pred): my_lpips/__init__.py:42)
 *   vsp_resize_bilinear_bwd_f32  adjoint of vsp_resize_bilinear_f32 (F.interpolate(size=112) in Loss/id_loss.py:37-41); dx overwritten
 * ---------------------------------------------------------------------------------------------- */
int vsp_maxpool2d_f32(float* out, const float* x, int64_t planes, int H, int W, int OH, int OW, int k, int s, int p,
                      vsp_stream_t stream);
int vsp_maxpool2d_bwd_f32(float* dx, const float* dy, const float* x, int64_t planes, int H, int W, int OH, int OW, int k, int s,
                          int p, vsp_stream_t stream);
int vsp_lpips_layer_f32(float* out, const float* f0, const float* f1, const float* w, int B, int C, int HW, vsp_stream_t stream);
int vsp_lpips_layer_bwd_f32(float* df1, const float* f0, const float* f1, const float* w, const float* gout, int B, int C, int HW,
                            vsp_stream_t stream);
int vsp_resize_bilinear_bwd_f32(float* dx, const float* dy, int64_t planes, int IH, int IW, int OH, int OW, vsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Keyed random tensors -- replaces the path's global-RNG draws: one `image.new_empty(B,1,H,W).normal_()` per NoiseInjection
 * (reference models/RestoreNet.py:564-569, e4e/models/stylegan2/model.py:287-292), `torch.randn(shape)` for x_T
 * (ldm/ddpm.py:423), `torch.randn(batch, latent_dim)` for z (restoration_test.py:77-82) and the synthetic LQ batch of the
 * benchmark.  ONE launch fills n_seg tensors laid out back to back in `out`, segment s = [B][seg_elems[s]] floats;
 *   value(s, b, e) = f( Philox4x32-10( key = seed, counter = (e / 4, seg_ids[s], image_index0 + b) ) word e % 4 ),
 * dist 0: standard normal (Box-Muller), dist 1: uniform(-1, 1).  The value depends on the GLOBAL image index only, so a
 * batch sharded over ranks draws what a single GPU would (SURVEY 8e "per-rank RNG streams derived from (seed, global image
 * index)").  seg_elems / seg_ids are HOST arrays; n_seg <= VSP_NOISE_MAX_SEGMENTS. */
#define VSP_NOISE_MAX_SEGMENTS 64
int vsp_keyed_fill_f32(float* out, int B, const int64_t* seg_elems, const int32_t* seg_ids, int n_seg, uint64_t seed,
                       int64_t image_index0, const int64_t* image_index0_dev, int dist, vsp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VSPBFR_HIP_H */

// Host side of vsp_conv2d_f32: tile-configuration table (assembled from the instantiation units conv_inst_*.hip),
// cost model, argument checks and launch.  The kernel itself is the template in conv_kernel.h.
#include "conv_kernel.h"
#include <cstdlib>
#include <cstring>

namespace vspconv {
extern const Cfg kCfgsA[];
extern const int kNumA;
extern const Cfg kCfgsB[];
extern const int kNumB;
extern const Cfg kCfgsC[];
extern const int kNumC;
extern const Cfg kCfgsD[];
extern const int kNumD;
extern const Cfg kCfgsE[];
extern const int kNumE;
extern const Cfg kCfgsF[];
extern const int kNumF;
extern const Cfg kCfgsG[];
extern const int kNumG;
extern const Cfg kCfgsP[];  // conv_pipe.hip: the double-buffered single-pipeline kernels (PF field 3)
extern const int kNumP;
}  // namespace vspconv

namespace {

using vspconv::Cfg;
using vspconv::ConvK;

// neutral operands for absent epilogue inputs (read through a zero stride): [0] = 1, [1] = 0, [4] = 0.2, [5] = 0.01
__device__ float kConst[8] = {1.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};

constexpr int kMaxCfgs = 192;
static Cfg kCfgs[kMaxCfgs];
static int kNumCfgs = 0;

static void build_table() {
  if (kNumCfgs) return;
  int n = 0;
  for (int i = 0; i < vspconv::kNumA && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsA[i];
  for (int i = 0; i < vspconv::kNumB && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsB[i];
  for (int i = 0; i < vspconv::kNumC && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsC[i];
  for (int i = 0; i < vspconv::kNumD && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsD[i];
  for (int i = 0; i < vspconv::kNumE && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsE[i];
  for (int i = 0; i < vspconv::kNumF && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsF[i];
  for (int i = 0; i < vspconv::kNumG && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsG[i];
  for (int i = 0; i < vspconv::kNumP && n < kMaxCfgs; ++i) kCfgs[n++] = vspconv::kCfgsP[i];
  kNumCfgs = n;
}

constexpr size_t kMaxLds = 100 * 1024;  // > 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize (set once per kernel)
constexpr size_t kMaxLdsPipe = 160 * 1024;  // conv_pipe.hip: one workgroup may own the whole LDS of a CU

struct Plan {
  int cfg;
  int tw_log2, th, tiles_x, tiles_y, co_tiles;
  bool dg;
  int strip_col, strip_row;  // transposed mode edge strips (strip_col < 0: ragged tiles instead)
  size_t lds;
  int ps;                    // conv_pipe.hip: plane pitch of the staged patch (floats)
};

static int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static int host_round_pitch(int n, int odd) {
  if (odd) return n | 1;
  int p = (n & ~31) + 16;
  return p >= n ? p : p + 32;
}

// Fill the geometry of configuration c for problem p; returns false if it does not fit (LDS / index limits).
static bool is_tc(const Cfg& k) { return k.name[strlen(k.name) - 1] == 't'; }
static bool is_dg(const Cfg& k) { return k.name[strlen(k.name) - 1] == 'd'; }
static bool is_kg(const Cfg& k) { return k.name[strlen(k.name) - 1] == 'a'; }   // conv_pipe.hip: data gradient of the dilation groups
static bool is_fg(const Cfg& k) { const size_t n = strlen(k.name); return n >= 2 && k.name[n - 2] == 'f'; }   // conv_pipe.hip fixed geometry

static bool make_plan(const vsp_conv_params& p, int c, Plan* out) {
  const Cfg& k = kCfgs[c];
  const bool tc = is_tc(k);
  if (tc != (p.transposed != 0)) return false;
  const bool kg = is_kg(k);
  if (kg != (p.dil_by_input_quarter != 0)) return false;
  const bool dg = is_dg(k) || kg;   // (the data-gradient form shares the staging geometry and the weight image)
  if (dg) {  // four dilated branches over one input, padding = dilation (SMART / LargeConv layers)
    if (p.G != 4 || p.x_group_stride != 0 || p.stride_y != 1 || p.stride_x != 1 || p.KH != 3 || p.KW != 3) return false;
    for (int g = 0; g < 4; ++g)
      if (p.pad_y[g] != p.dil[g] || p.pad_x[g] != p.dil[g]) return false;
  }
  const int CO_T = 16 * k.MB * k.WM, NPIX = tc ? 16 * k.WN * (k.NB / 4) : 16 * k.NB * k.WN;
  const int WS = (CO_T % 32 == 0) ? CO_T + 16 : CO_T;
  int twl = ilog2_ceil(p.OW);
  // cap the tile width: 16-pixel MFMA column blocks want >= 16 contiguous pixels; wider tiles cut halo re-reads
  int cap = NPIX >= 256 ? 5 : 4;  // 32 or 16
  if (k.PF == 3 && dg) cap = NPIX >= 512 ? 5 : 4;  // shared patch with a halo of 8: square tiles stage the fewest words
  if (twl > cap) twl = cap;
  if ((1 << twl) > NPIX) twl = ilog2_ceil(NPIX);
  const bool fg = k.PF == 3 && is_fg(k);
  if (fg) {  // conv_pipe.hip fixed geometry (dilation-group mode): 16 x 16 pixels, dilations exactly 1, 2, 4, 8
    twl = 4;
    if (!dg || p.dil[0] != 1 || p.dil[1] != 2 || p.dil[2] != 4 || p.dil[3] != 8) return false;
  }
  const int TW = 1 << twl, TH = NPIX / TW;
  int dmax = 1;
  for (int g = 0; g < (p.G > 4 ? 1 : p.G); ++g) dmax = p.dil[g] > dmax ? p.dil[g] : dmax;
  const int PH = tc ? TH + 1 : (TH - 1) * p.stride_y + (p.KH - 1) * dmax + 1;
  const int PW = tc ? TW + 1 : (TW - 1) * p.stride_x + (p.KW - 1) * dmax + 1;
  if (PW > 256 || PH * PW >= 65536) return false;
  int plane = PH * PW;
  out->strip_col = -1;
  out->strip_row = 0;
  out->tiles_x = (p.OW + TW - 1) / TW;
  out->tiles_y = (p.OH + TH - 1) / TH;
  if (tc && (NPIX <= 128 || k.PF == 3) && p.H % TH == 0 && p.W % TW == 0) {
    out->tiles_x = p.W / TW;
    out->tiles_y = p.H / TH;
    out->strip_col = (p.H + NPIX - 1) / NPIX;
    out->strip_row = (p.W + 1 + NPIX - 1) / NPIX;
    if ((NPIX + 1) * 2 > plane) plane = (NPIX + 1) * 2;
  }
  if (k.PF == 3) {  // conv_pipe.hip: 3x3 only, vector weight rows, whole 64-word patch rows per wave, two LDS buffers
    if (p.KH != 3 || p.KW != 3 || p.cout_g % 4 != 0 || !vsp::aligned16(p.w) || p.in_shift || p.Cin % k.CK != 0) return false;
    if (kg && (p.Cin % 16 != 0 || (p.Cin / 4) % k.CK != 0)) return false;   // a chunk lies inside one input-channel quarter
    const int NW = k.WM * k.WN, wpc = k.CK >= NW ? 1 : NW / k.CK;
    const int nrow = (plane + 63) / 64;
    // (PMAX field = PROWS: 64-word rows of a channel plane per wave; checked below once the pitch is known)
    const int ps = host_round_pitch(k.PMAX * wpc * 64, !tc && p.stride_x != 1);   // every staged row lands inside its own plane
    if (nrow > k.PMAX * wpc) return false;
    const size_t lds = 2 * ((size_t)9 * k.CK * WS + (size_t)k.CK * ps) * sizeof(float);
    if (lds > kMaxLdsPipe) return false;
    out->cfg = c; out->tw_log2 = twl; out->th = TH;
    out->co_tiles = dg ? (p.cout_g + 15) / 16 : (p.cout_g + CO_T - 1) / CO_T;
    out->dg = dg; out->lds = lds; out->ps = ps;
    return true;
  }
  const int PS = host_round_pitch(plane, !tc && p.stride_x != 1);
  // LDS-DMA staging: no input shift; 1x1 / 3x3 kernels only (the tap walk of the DMA variants was found wrong on a 4x4 kernel by the
  // result check of tools/autotune_train.py; tests/test_hip_ops.py::test_conv_every_config_agrees keeps every variant honest)
  if (k.PF == 2 && (p.in_shift || p.cout_g % 4 != 0 || !vsp::aligned16(p.w) || p.KH * p.KW > 9)) return false;
  size_t lds = ((size_t)p.KH * p.KW * k.CK * WS + (size_t)(k.PF == 2 ? 2 : 1) * k.CK * PS) * sizeof(float);
  const size_t red = (size_t)(k.WK - 1) * k.WM * k.WN * k.MB * k.NB * 4 * 64 * sizeof(float);
  if (red > lds) lds = red;
  if (lds > kMaxLds) return false;
  out->cfg = c;
  out->tw_log2 = twl;
  out->th = TH;
  out->co_tiles = dg ? (p.cout_g + 15) / 16 : (p.cout_g + CO_T - 1) / CO_T;
  out->dg = dg;
  out->lds = lds;
  return true;
}

// Cost model: MFMA slots issued (padded tile work), in waves of resident blocks over 256 CUs, plus a per-chunk
// staging term.  Only relative order matters.
static double plan_cost(const vsp_conv_params& p, const Plan& pl) {
  const Cfg& k = kCfgs[pl.cfg];
  const int CO_T = 16 * k.MB * k.WM, NPIX = 16 * k.NB * k.WN;
  const int waves = k.WM * k.WN * k.WK;
  const double blocks = ((double)pl.tiles_x * pl.tiles_y + (pl.strip_col > 0 ? pl.strip_col + pl.strip_row : 0)) * pl.co_tiles * (pl.dg ? 1 : p.G) * p.B;
  const int cin_pad = (p.Cin + k.CK - 1) / k.CK * k.CK;
  // cycles one block needs on one SIMD-set: each wave issues MB*NB MFMAs (32 cyc) per k-step
  const double ksteps = (double)p.KH * p.KW * cin_pad / 4.0;
  const double mfma_cyc = ksteps * k.MB * k.NB * 32.0 / k.WK;  // per wave
  const double reads = ksteps * (k.MB + k.NB) * 8.0 / k.WK;  // LDS issue cost per wave, overlappable: small weight
  const double chunks = (double)cin_pad / k.CK;
  const double stage = chunks * 2500.0;                  // barrier + global latency + LDS writes per chunk
  int per_cu = (int)(160 * 1024 / (pl.lds + 1024));
  const int by_waves = 8 / waves * 4 / 4;  // keep <= 2 waves per SIMD per block set: 8 waves/CU when 4-wave blocks
  (void)by_waves;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  // waves per SIMD when the CU is full
  const double wps = per_cu * waves / 4.0;
  // block time when co-resident with (per_cu-1) others: MFMA pipe shared, staging hidden if wps >= 2
  double t_block = mfma_cyc * (wps < 1.0 ? 1.0 : wps) + (wps >= 2.0 ? 0.15 : 1.0) * stage + 0.05 * reads;
  const double slots = 256.0 * per_cu;
  const double rounds = blocks / slots;
  const double full = (rounds < 1.0) ? 1.0 : rounds;  // a partially filled chip still takes one block time
  // when the chip is under-filled the blocks do not share SIMDs: undo the sharing factor
  if (rounds < 1.0) {
    const double occ = blocks / 256.0;  // blocks per CU
    const double w = occ * waves / 4.0;
    t_block = mfma_cyc * (w < 1.0 ? 1.0 : w) + stage + 0.05 * reads;
  }
  (void)CO_T;
  (void)NPIX;
  return full * t_block;
}

// Address of kConst on the current device (resolved once per device; the first conv call of a process must not
// happen inside a stream capture -- the Python loader makes a warm-up call at import).
static const float* device_consts() {
  static const float* cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    vsp::set_error("hipGetDevice failed");
    return nullptr;
  }
  if (!cache[dev]) {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(kConst));
    if (e != hipSuccess) {
      vsp::set_error("hipGetSymbolAddress(kConst): %s", hipGetErrorString(e));
      return nullptr;
    }
    // slots: [0]=1 [1]=0 [2]=1 [3]=0 [4]=0.2 [5]=0.01 (constant slopes)
    const float init[8] = {1.f, 0.f, 1.f, 0.f, 0.2f, 0.01f, 0.f, 0.f};
    e = hipMemcpy(ptr, init, sizeof(init), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      vsp::set_error("hipMemcpy(kConst): %s", hipGetErrorString(e));
      return nullptr;
    }
    cache[dev] = static_cast<const float*>(ptr);
  }
  return cache[dev];
}

static const float* slope_slot(const float* kc, float slope) {
  if (slope == 0.2f) return kc + 4;
  if (slope == 0.01f) return kc + 5;
  if (slope == 1.f) return kc;
  if (slope == 0.f) return kc + 1;
  return nullptr;
}

}  // namespace

// Argument checks shared by vsp_conv2d_f32 and vsp_conv2d_winograd_f32; *empty = nothing to do.
static int validate_conv(const vsp_conv_params& p, int* x_ch_out, bool* empty, bool allow_w_bstride = false) {
  *empty = false;
  VSP_REQUIRE(p.x && p.w && p.y, "conv2d: null tensor pointer");
  VSP_REQUIRE(p.B >= 0 && p.Cin >= 1 && p.H >= 1 && p.W >= 1, "conv2d: bad input dims B=%d Cin=%d H=%d W=%d", p.B,
              p.Cin, p.H, p.W);
  VSP_REQUIRE(p.G >= 1 && p.G <= 1024 && p.cout_g >= 1, "conv2d: bad group spec G=%d cout_g=%d", p.G, p.cout_g);
  VSP_REQUIRE(p.x_group_stride >= 0 && p.x_ch >= 0, "conv2d: negative x_ch / x_group_stride");
  VSP_REQUIRE(p.w_bstride == 0 || allow_w_bstride, "conv2d: per-image weights (w_bstride) are served by vsp_conv2d_bf16 with io_bf16 = 1 only");
  const int x_ch = p.x_ch > 0 ? p.x_ch : p.Cin;
  VSP_REQUIRE((int64_t)(p.G - 1) * p.x_group_stride + p.Cin <= x_ch, "conv2d: group input channels exceed x_ch=%d", x_ch);
  VSP_REQUIRE(p.x_group_stride == 0 || !p.in_shift, "conv2d: an input shift is not supported with grouped input");
  VSP_REQUIRE(p.KH >= 1 && p.KW >= 1 && p.KH * p.KW <= 49, "conv2d: unsupported kernel %dx%d", p.KH, p.KW);
  VSP_REQUIRE(p.stride_y >= 1 && p.stride_x >= 1 && p.stride_x <= 2 && p.stride_y <= 2, "conv2d: stride must be 1 or 2");
  VSP_REQUIRE(p.dil_by_input_quarter == 0 || (p.dil_by_input_quarter == 1 && p.G == 4 && p.x_group_stride == 0 && p.Cin % 16 == 0 && !p.transposed),
              "conv2d: dil_by_input_quarter needs G = 4 blocks of output channels over one shared input with Cin %% 16 == 0");
  VSP_REQUIRE(p.OH >= 0 && p.OW >= 0, "conv2d: negative output size");
  VSP_REQUIRE(p.osy >= 1 && p.osx >= 1 && p.ooy >= 0 && p.oox >= 0, "conv2d: bad output stride/offset");
  VSP_REQUIRE(!p.noise || p.noise_w, "conv2d: noise given without noise_w");
  VSP_REQUIRE(p.act2 != 2 || p.prelu, "conv2d: act2=prelu without slopes");
  for (int g = 0; g < (p.G > 4 ? 1 : p.G); ++g) VSP_REQUIRE(p.dil[g] >= 1, "conv2d: dilation must be >= 1");
  if (p.B == 0 || p.OH == 0 || p.OW == 0) {
    *empty = true;
    return VSP_OK;
  }
  const int Cout = p.G * p.cout_g;
  VSP_REQUIRE(p.y_coff >= 0 && p.y_coff + Cout <= p.y_ch, "conv2d: output channel window [%d,%d) outside %d", p.y_coff,
              p.y_coff + Cout, p.y_ch);
  VSP_REQUIRE((p.OH - 1) * p.osy + p.ooy < p.y_h && (p.OW - 1) * p.osx + p.oox < p.y_w,
              "conv2d: output positions exceed the %dx%d output tensor", p.y_h, p.y_w);
  VSP_REQUIRE((int64_t)p.y_ch * p.y_h * p.y_w < ((int64_t)1 << 31) && (int64_t)x_ch * p.H * p.W < ((int64_t)1 << 31),
              "conv2d: one image must hold fewer than 2^31 elements");
  if (p.res1 || p.res2)
    VSP_REQUIRE(p.res_coff >= 0 && p.res_coff + Cout <= p.res_ch, "conv2d: residual channel window out of range");

  *x_ch_out = x_ch;
  return VSP_OK;
}

// Everything of the kernel argument block that does not depend on the tile plan (operands resolved to pointer + stride).
static int fill_convk(const vsp_conv_params& p, int x_ch, ConvK& q) {
  q.x = p.x; q.w = p.w; q.y = p.y;
  q.B = p.B; q.Cin = p.Cin; q.H = p.H; q.W = p.W; q.G = p.G; q.cout_g = p.cout_g;
  q.OH = p.OH; q.OW = p.OW; q.KH = p.KH; q.KW = p.KW; q.sy = p.stride_y; q.sx = p.stride_x;
  for (int g = 0; g < 4; ++g) { q.dil[g] = p.dil[g]; q.pady[g] = p.pad_y[g]; q.padx[g] = p.pad_x[g]; }
  q.y_ch = p.y_ch; q.y_coff = p.y_coff; q.y_h = p.y_h; q.y_w = p.y_w;
  q.osy = p.osy; q.osx = p.osx; q.ooy = p.ooy; q.oox = p.oox;
  q.in_scale = p.in_scale; q.in_scale_bstride = p.in_scale_bstride; q.in_shift = p.in_shift;
  const float* kc = device_consts();
  VSP_REQUIRE(kc != nullptr, "conv2d: cannot resolve device constants: %s", vsp_last_error());
  const float* kOne = kc;
  const float* kZero = kc + 1;
  auto sel = [](const float* ptr, const float* dflt, const float** outp, int* outs) {
    *outp = ptr ? ptr : dflt;
    *outs = ptr ? 1 : 0;
  };
  sel(p.out_scale, kOne, &q.osp, &q.oss);
  sel(p.ch_scale, kOne, &q.csp, &q.css);
  sel(p.ch_bias, kZero, &q.cbp, &q.cbs);
  sel(p.act1 ? p.bias1 : nullptr, kZero, &q.b1p, &q.b1s);
  q.s1 = p.act1 ? p.slope1 : 1.f;
  q.g1 = p.act1 ? p.gain1 : 1.f;
  sel(p.noise, kZero, &q.nzp, &q.nzs);
  q.nwp = p.noise ? p.noise_w : kZero;
  sel(p.act2 == 1 ? p.bias2 : nullptr, kZero, &q.b2p, &q.b2s);
  if (p.act2 == 2) {
    q.s2p = p.prelu; q.s2s = 1; q.g2 = 1.f;
  } else if (p.act2 == 1) {
    // constant slope: staged in the per-launch slot of the constant block is not possible (async launches share
    // it), so the two slopes used by the path live in fixed slots: 0.2 (FusedLeakyReLU) and 0.01 (nn.LeakyReLU)
    const float* slot = slope_slot(kc, p.slope2);
    VSP_REQUIRE(slot != nullptr, "conv2d: unsupported act2 slope %g (supported: 0.2, 0.01, 0, 1)", p.slope2);
    q.s2p = slot; q.s2s = 0; q.g2 = p.gain2;
  } else {
    q.s2p = kOne; q.s2s = 0; q.g2 = 1.f;
  }
  sel(p.res1, kZero, &q.r1p, &q.r1s);
  sel(p.res2, kZero, &q.r2p, &q.r2s);
  q.res_ch = p.res_ch; q.res_coff = p.res_coff;
  q.x_ch = x_ch;
  q.x_gs = p.x_group_stride;
  return VSP_OK;
}

// configuration index kNumCfgs (tile_hint kNumCfgs + 1) names the small-map kernel of conv_smallmap.hip
extern "C" int vsp_conv2d_num_configs(void) {
  build_table();
  return kNumCfgs + 1;
}
extern "C" const char* vsp_conv2d_config_name(int i) {
  build_table();
  return (i >= 0 && i < kNumCfgs) ? kCfgs[i].name : (i == kNumCfgs ? "smallmap" : "");
}

extern "C" int vsp_conv2d_winograd_chunk(void) { return vspconv::wino_chunk(); }
extern "C" int vsp_conv2d_winograd_mbw(int cout_g) { return vspconv::wino_mbw(cout_g); }

extern "C" int vsp_conv2d_winograd_f32(const vsp_conv_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d_winograd: null params");
  const vsp_conv_params& p = *pp;
  VSP_REQUIRE(!p.transposed && p.G >= 1 && p.G <= 4 && p.KH == 3 && p.KW == 3 && p.stride_y == 1 && p.stride_x == 1 &&
                  p.x_group_stride == 0,
              "conv2d_winograd: only 3x3, stride 1, at most four groups over one shared input");
  for (int g = 0; g < p.G; ++g)
    VSP_REQUIRE((p.dil[g] == 1 || p.dil[g] == 2 || p.dil[g] == 4 || p.dil[g] == 8) && p.pad_y[g] == p.dil[g] && p.pad_x[g] == p.dil[g],
                "conv2d_winograd: group %d needs dilation 1, 2, 4 or 8 and padding = dilation (got dilation %d, padding %d/%d)", g,
                p.dil[g], p.pad_y[g], p.pad_x[g]);
  VSP_REQUIRE(p.io_bf16 == 0, "conv2d_winograd: fp32 activations only (io_bf16 is served by vsp_conv2d_bf16)");
  VSP_REQUIRE(p.dil_by_input_quarter == 0, "conv2d_winograd: dil_by_input_quarter is served by vsp_conv2d_f32");
  VSP_REQUIRE(p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0, "conv2d_winograd: dense output only");
  VSP_REQUIRE(p.OH == p.H && p.OW == p.W, "conv2d_winograd: output size must equal the input size");
  VSP_REQUIRE(vsp::aligned16(p.w), "conv2d_winograd: transformed weights must be 16-byte aligned");
  VSP_REQUIRE((int64_t)16 * p.Cin * p.cout_g < ((int64_t)1 << 31), "conv2d_winograd: weight tensor too large");
  VSP_REQUIRE((int64_t)(p.x_ch > 0 ? p.x_ch : p.Cin) * p.H * p.W * 4 < ((int64_t)1 << 31),
              "conv2d_winograd: one input image must be smaller than 2 GiB (32-bit buffer offsets)");
  int x_ch = 0;
  bool empty = false;
  if (int rc = validate_conv(p, &x_ch, &empty)) return rc;
  if (empty) return VSP_OK;
  ConvK q{};
  if (int rc = fill_convk(p, x_ch, q)) return rc;
  {
    static const int dbg = vsp::tune_env("VSP_CONV_DBG") ? atoi(vsp::tune_env("VSP_CONV_DBG")) : 0;
    q.dbg = dbg;
  }
  {
    const float* kc = device_consts();
    const bool affine = p.in_shift != nullptr;
    q.wtp = (p.in_scale && !affine) ? p.in_scale : kc;             q.wt_cs = (p.in_scale && !affine) ? 1 : 0;
    q.wt_bs = (p.in_scale && !affine) ? p.in_scale_bstride : 0;
    q.wcp = (p.in_scale && affine) ? p.in_scale : kc;              q.wc_cs = (p.in_scale && affine) ? 1 : 0;
    q.wc_bs = (p.in_scale && affine) ? p.in_scale_bstride : 0;
    q.wshp = affine ? p.in_shift : kc + 1;                         q.wsh_cs = affine ? 1 : 0;
  }
  VSP_REQUIRE(p.tile_hint >= 0 && p.tile_hint <= 3, "conv2d_winograd: tile_hint names the kernel form: 0 automatic, 1 task list, 2 row owner, 3 register-resident U");
  if (int rc = vspconv::wino_launch(q, p.tile_hint, vsp::as_stream(stream))) return rc;
  return vsp::check_launch("conv2d_winograd");
}

extern "C" size_t vsp_winograd4_weight_floats(int cin, int cout) { return vspconv::wino4_weight_floats(cin, cout); }

extern "C" int vsp_winograd4_weight_f32(float* U, const float* wp, int cin, int cout, vsp_stream_t stream) {
  VSP_REQUIRE(U && wp && cin >= 1 && cout >= 1, "winograd4_weight: bad arguments");
  if (int rc = vspconv::wino4_weight_launch(U, wp, cin, cout, vsp::as_stream(stream))) return rc;
  return vsp::check_launch("winograd4_weight");
}

extern "C" size_t vsp_conv2d_winograd4_work_floats(const vsp_conv_params* pp) {
  return pp ? vspconv::wino4_work_floats(pp->B, pp->Cin, pp->H, pp->W) : 0;
}

extern "C" int vsp_conv2d_winograd4_f32(const vsp_conv_params* pp, float* work, size_t work_floats, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr && work != nullptr, "conv2d_winograd4: null params / work buffer");
  const vsp_conv_params& p = *pp;
  VSP_REQUIRE(!p.transposed && p.G == 1 && p.KH == 3 && p.KW == 3 && p.stride_y == 1 && p.stride_x == 1 && p.dil[0] == 1 &&
                  p.pad_y[0] == 1 && p.pad_x[0] == 1,
              "conv2d_winograd4: one group, 3x3, stride 1, dilation 1, padding 1");
  VSP_REQUIRE(p.io_bf16 == 0 && p.dil_by_input_quarter == 0 && p.in_shift == nullptr, "conv2d_winograd4: fp32, no affine input shift");
  VSP_REQUIRE(p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.OH == p.H && p.OW == p.W, "conv2d_winograd4: dense same-size output");
  VSP_REQUIRE(vsp::aligned16(p.w) && vsp::aligned16(work), "conv2d_winograd4: weights and work buffer must be 16-byte aligned");
  VSP_REQUIRE(work_floats >= vspconv::wino4_work_floats(p.B, p.Cin, p.H, p.W), "conv2d_winograd4: work buffer too small");
  VSP_REQUIRE((int64_t)vspconv::wino4_work_floats(1, p.Cin, p.H, p.W) * 4 < ((int64_t)1 << 40), "conv2d_winograd4: image too large");
  int x_ch = 0;
  bool empty = false;
  if (int rc = validate_conv(p, &x_ch, &empty)) return rc;
  if (empty) return VSP_OK;
  ConvK q{};
  if (int rc = fill_convk(p, x_ch, q)) return rc;
  {
    static const int dbg = vsp::tune_env("VSP_CONV_DBG") ? atoi(vsp::tune_env("VSP_CONV_DBG")) : 0;
    q.dbg = dbg;
  }
  if (!vspconv::wino4_eligible(q))
    return vsp::fail(VSP_ENOTSUP, "conv2d_winograd4: needs Cin %% 4 == 0, H, W %% 4 == 0, dense 16-byte aligned output / noise / residual planes");
  if (int rc = vspconv::wino4_launch(q, work, p.in_scale, p.in_scale ? p.in_scale_bstride : 0, vsp::as_stream(stream))) return rc;
  return vsp::check_launch("conv2d_winograd4");
}

extern "C" size_t vsp_winograd4f_weight_floats(int cin, int cout) { return vspconv::wino4f_weight_floats(cin, cout); }

extern "C" int vsp_winograd4f_weight_f32(float* U, const float* wp, int cin, int cout, vsp_stream_t stream) {
  VSP_REQUIRE(U && wp && cin >= 1 && cout >= 1, "winograd4f_weight: bad arguments");
  if (int rc = vspconv::wino4f_weight_launch(U, wp, cin, cout, vsp::as_stream(stream))) return rc;
  return vsp::check_launch("winograd4f_weight");
}

extern "C" int vsp_conv2d_winograd4f_f32(const vsp_conv_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d_winograd4f: null params");
  const vsp_conv_params& p = *pp;
  VSP_REQUIRE(!p.transposed && p.G >= 1 && p.G <= 4 && p.KH == 3 && p.KW == 3 && p.stride_y == 1 && p.stride_x == 1,
              "conv2d_winograd4f: 3x3, stride 1, one group or up to four dilation groups over one shared input");
  for (int g = 0; g < p.G; ++g)
    VSP_REQUIRE((p.dil[g] == 1 || p.dil[g] == 2 || p.dil[g] == 4 || p.dil[g] == 8) && p.pad_y[g] == p.dil[g] && p.pad_x[g] == p.dil[g],
                "conv2d_winograd4f: dilation 1 / 2 / 4 / 8, padding = dilation");
  VSP_REQUIRE(p.io_bf16 == 0 && p.dil_by_input_quarter == 0 && p.in_shift == nullptr, "conv2d_winograd4f: fp32, no affine input shift");
  VSP_REQUIRE(p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.OH == p.H && p.OW == p.W, "conv2d_winograd4f: dense same-size output");
  VSP_REQUIRE(vsp::aligned16(p.w), "conv2d_winograd4f: weights must be 16-byte aligned");
  int x_ch = 0;
  bool empty = false;
  if (int rc = validate_conv(p, &x_ch, &empty)) return rc;
  if (empty) return VSP_OK;
  ConvK q{};
  if (int rc = fill_convk(p, x_ch, q)) return rc;
  {
    static const int dbg = vsp::tune_env("VSP_CONV_DBG") ? atoi(vsp::tune_env("VSP_CONV_DBG")) : 0;
    q.dbg = dbg;
    const float* kc = device_consts();
    q.wtp = p.in_scale ? p.in_scale : kc;
    q.wt_cs = p.in_scale ? 1 : 0;
    q.wt_bs = p.in_scale ? p.in_scale_bstride : 0;
    q.wshp = kc + 1;
    q.wsh_cs = 0;
  }
  if (!vspconv::wino4f_eligible(q))
    return vsp::fail(VSP_ENOTSUP, "conv2d_winograd4f: needs Cin %% 8 == 0 (<= 512), H, W %% (4 x dilation) == 0, W >= 16, groups over one shared input, dense 16-byte aligned input / output / noise / residual planes < 2 GiB per image");
  if (int rc = vspconv::wino4f_launch(q, vsp::as_stream(stream))) return rc;
  return vsp::check_launch("conv2d_winograd4f");
}

static int conv2d_bf16_impl(const vsp_conv_params* pp, vsp_stream_t stream, bool split, int rv = 0);   // rv: 1 = conv_bf16_rv.hip, 2 = conv_bf16_dg.hip
extern "C" int vsp_conv2d_bf16(const vsp_conv_params* pp, vsp_stream_t stream) { return conv2d_bf16_impl(pp, stream, false); }
extern "C" int vsp_conv2d_bf16x3(const vsp_conv_params* pp, vsp_stream_t stream) { return conv2d_bf16_impl(pp, stream, true); }
// the row-vector-K kernel (conv_bf16_rv.hip): same operands as vsp_conv2d_bf16 with io_bf16 = 1, its own weight order; VSP_ENOTSUP when the launch is not one it serves
extern "C" int vsp_conv2d_bf16rv(const vsp_conv_params* pp, vsp_stream_t stream) { return conv2d_bf16_impl(pp, stream, false, 1); }
// the dilation-group kernel (conv_bf16_dg.hip): bf16 activations, `w` = the PACKED FP32 weights of vsp_conv2d_f32 (converted in the kernel); VSP_ENOTSUP when it does not serve the launch
extern "C" int vsp_conv2d_bf16dg(const vsp_conv_params* pp, vsp_stream_t stream) { return conv2d_bf16_impl(pp, stream, false, 2); }

static int conv2d_bf16_impl(const vsp_conv_params* pp, vsp_stream_t stream, bool split, int rv) {
  VSP_REQUIRE(pp != nullptr, "conv2d_bf16: null params");
  vsp_conv_params pcopy = *pp;
  int mode = 0;
  VSP_REQUIRE(pcopy.KH == 3 && pcopy.KW == 3 && pcopy.G >= 1, "conv2d_bf16: 3x3 kernels only");
  if (pcopy.transposed) {  // same normalisation as vsp_conv2d_f32: the grid runs over input positions m = 0..H, n = 0..W
    mode = 2;
    VSP_REQUIRE(pcopy.G == 1, "conv2d_bf16: transposed mode needs G = 1");
    VSP_REQUIRE(!pcopy.noise && !pcopy.res1 && !pcopy.res2, "conv2d_bf16: transposed mode has no noise / residual epilogue");
    VSP_REQUIRE(pcopy.y_h == 2 * pcopy.H + 1 && pcopy.y_w == 2 * pcopy.W + 1, "conv2d_bf16: transposed output must be (2H+1)x(2W+1)");
    pcopy.stride_y = pcopy.stride_x = 1;
    pcopy.dil[0] = 1;
    pcopy.pad_y[0] = pcopy.pad_x[0] = 1;
    pcopy.OH = pcopy.H + 1;
    pcopy.OW = pcopy.W + 1;
    pcopy.osy = pcopy.osx = 2;
    pcopy.ooy = pcopy.oox = 0;
  } else if (pcopy.stride_y == 2 && pcopy.stride_x == 2) {
    mode = 1;
    VSP_REQUIRE(pcopy.dil[0] == 1 && pcopy.pad_y[0] == pcopy.pad_x[0] && (pcopy.pad_y[0] == 0 || pcopy.pad_y[0] == 1),
                "conv2d_bf16: stride 2 needs dilation 1 and padding 0 or 1");
    VSP_REQUIRE(pcopy.G == 1 || pcopy.x_group_stride > 0, "conv2d_bf16: stride-2 groups need their own input slices");
    VSP_REQUIRE(pcopy.OH <= (pcopy.H + 2 * pcopy.pad_y[0] - 3) / 2 + 1 && pcopy.OW <= (pcopy.W + 2 * pcopy.pad_x[0] - 3) / 2 + 1,
                "conv2d_bf16: output larger than the stride-2 convolution of the input");
  } else {
    VSP_REQUIRE(pcopy.stride_y == 1 && pcopy.stride_x == 1, "conv2d_bf16: stride must be 1 or 2");
    VSP_REQUIRE(pcopy.G <= 4 || pcopy.x_group_stride > 0, "conv2d_bf16: more than four groups need their own input slices");
    for (int g = 0; g < (pcopy.G > 4 ? 1 : pcopy.G); ++g)
      VSP_REQUIRE(pcopy.dil[g] >= 1 && pcopy.dil[g] <= 64 && pcopy.pad_y[g] == pcopy.dil[g] && pcopy.pad_x[g] == pcopy.dil[g],
                  "conv2d_bf16: group %d needs padding = dilation (got dilation %d, padding %d/%d)", g, pcopy.dil[g],
                  pcopy.pad_y[g], pcopy.pad_x[g]);
    VSP_REQUIRE(pcopy.OH == pcopy.H && pcopy.OW == pcopy.W, "conv2d_bf16: output size must equal the input size");
  }
  const vsp_conv_params& p = pcopy;
  if (mode != 2) VSP_REQUIRE(p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0, "conv2d_bf16: dense output only");
  VSP_REQUIRE(vsp::aligned16(p.w), "conv2d_bf16: packed weights must be 16-byte aligned");
  VSP_REQUIRE(p.Cin % 8 == 0, "conv2d_bf16: Cin must be a multiple of 8 (got %d)", p.Cin);
  VSP_REQUIRE((int64_t)(p.x_ch > 0 ? p.x_ch : p.Cin) * p.H * p.W * 4 < ((int64_t)1 << 31),
              "conv2d_bf16: one input image must be smaller than 2 GiB (32-bit buffer offsets)");
  int x_ch = 0;
  bool empty = false;
  const bool per_image_w = p.w_bstride != 0;
  if (per_image_w) {
    VSP_REQUIRE(!split && rv == 0 && p.io_bf16 == 1, "conv2d_bf16: per-image weights need the general bf16 kernel with bf16 activations");
    VSP_REQUIRE(p.w_bstride > 0 && p.w_bstride % 16 == 0, "conv2d_bf16: w_bstride must be a positive multiple of 16 bytes");
    VSP_REQUIRE(!p.in_scale && !p.in_shift, "conv2d_bf16: per-image weights carry the style: in_scale / in_shift must be NULL");
    VSP_REQUIRE((p.W & 1) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 3) == 0, "conv2d_bf16: per-image weights need even rows (pixel-pair staging)");
  }
  if (int rc = validate_conv(p, &x_ch, &empty, true)) return rc;
  if (empty) return VSP_OK;
  VSP_REQUIRE((int64_t)p.G * p.cout_g <= 65535 && p.B <= 65535, "conv2d_bf16: grid too large");
  ConvK q{};
  if (int rc = fill_convk(p, x_ch, q)) return rc;
  {  // the bf16 kernel reads its prologue operands unconditionally: absent ones are device constants through stride 0
    const float* kc = device_consts();
    q.bf_isc_s = q.in_scale ? 1 : 0;
    q.bf_ish_s = q.in_shift ? 1 : 0;
    if (!q.in_scale) { q.in_scale = kc; q.in_scale_bstride = 0; }
    if (!q.in_shift) q.in_shift = kc + 1;
  }
  VSP_REQUIRE(!(split && p.io_bf16), "conv2d_bf16x3: the split-precision form keeps fp32 activations (io_bf16 = 0)");
  VSP_REQUIRE(p.io_bf16 == 0 || p.io_bf16 == 1, "conv2d_bf16: io_bf16 must be 0 or 1");
  VSP_REQUIRE(p.dil_by_input_quarter == 0, "conv2d_bf16: dil_by_input_quarter is served by vsp_conv2d_f32");
  q.io_bf16 = p.io_bf16;
  q.w_bs = p.w_bstride / 16;
  {
    static const int dbg = vsp::tune_env("VSP_CONV_DBG") ? atoi(vsp::tune_env("VSP_CONV_DBG")) : 0;  // ablation builds only (VSP_BF16_ABLATE)
    q.dbg = dbg;
  }
  if (rv == 2) {
    if (mode != 0 || !vspconv::bf16dg_eligible(q)) return VSP_ENOTSUP;
    if (int rc = vspconv::bf16dg_launch(q, p.act1 != 0 || p.act2 != 0 || p.noise != nullptr, vsp::as_stream(stream))) return rc;
    return vsp::check_launch("conv2d_bf16dg");
  }
  if (rv) {
    if (mode != 0 || !vspconv::bf16rv_eligible(q)) return VSP_ENOTSUP;
    if (int rc = vspconv::bf16rv_launch(q, p.tile_hint, vsp::as_stream(stream))) return rc;
    return vsp::check_launch("conv2d_bf16rv");
  }
  if (split) {
    if (int rc = vspconv::bf16_launch_split(q, mode, p.tile_hint, vsp::as_stream(stream))) return rc;
    return vsp::check_launch("conv2d_bf16x3");
  }
  if (int rc = vspconv::bf16_launch(q, mode, p.tile_hint, vsp::as_stream(stream))) return rc;
  return vsp::check_launch("conv2d_bf16");
}

extern "C" int vsp_conv2d_f32(const vsp_conv_params* pp, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "conv2d: null params");
  VSP_REQUIRE(pp->io_bf16 == 0, "conv2d: fp32 activations only (io_bf16 is served by vsp_conv2d_bf16)");
  build_table();
  vsp_conv_params pcopy = *pp;
  if (pcopy.transposed) {  // normalise the ignored fields: the launch grid runs over input positions m = 0..H, n = 0..W
    VSP_REQUIRE(pcopy.KH == 3 && pcopy.KW == 3 && pcopy.G == 1, "conv2d: transposed mode needs a 3x3 kernel and G = 1");
    VSP_REQUIRE(!pcopy.noise && !pcopy.res1 && !pcopy.res2, "conv2d: transposed mode has no noise / residual epilogue");
    // (a larger output plane is allowed: the adjoint of a stride-2 conv over an even-sized input is one row / column wider; the
    //  kernel only writes the (2H+1) x (2W+1) positions, the caller zeroes the margin)
    VSP_REQUIRE(pcopy.y_h >= 2 * pcopy.H + 1 && pcopy.y_w >= 2 * pcopy.W + 1, "conv2d: transposed output must hold (2H+1)x(2W+1)");
    pcopy.stride_y = pcopy.stride_x = 1;
    pcopy.dil[0] = 1;
    pcopy.pad_y[0] = pcopy.pad_x[0] = 1;
    pcopy.OH = pcopy.H + 1;
    pcopy.OW = pcopy.W + 1;
    pcopy.osy = pcopy.osx = 2;
    pcopy.ooy = pcopy.oox = 0;
  }
  const vsp_conv_params& p = pcopy;
  int x_ch = 0;
  bool empty = false;
  if (int rc = validate_conv(p, &x_ch, &empty)) return rc;
  if (empty) return VSP_OK;
  VSP_REQUIRE((int64_t)x_ch * p.H * p.W * 4 < ((int64_t)1 << 31) && (int64_t)p.KH * p.KW * p.Cin * p.cout_g * 4 < ((int64_t)1 << 31),
              "conv2d: one input image and one group's weights must each be smaller than 2 GiB (32-bit buffer offsets)");
  const int Cout = p.G * p.cout_g;
  (void)Cout;
  {
    // the small-map kernel: named (tile_hint kNumCfgs + 1), preferred by the tuned table (negative), or -- for a shape the table
    // does not know -- when the whole batch has at most 256 output positions against K >= 1024
    const int sm = kNumCfgs + 1;
    const bool named = p.tile_hint == sm, preferred = p.tile_hint == -sm;
    const bool guess = p.tile_hint == 0 && (int64_t)p.B * p.OH * p.OW <= 256 && (int64_t)p.Cin * p.KH * p.KW >= 1024;
    if ((named || preferred || guess) && !p.dil_by_input_quarter) {
      ConvK q{};
      if (int rc = fill_convk(p, x_ch, q)) return rc;
      {
        static const int dbg = vsp::tune_env("VSP_CONV_DBG") ? atoi(vsp::tune_env("VSP_CONV_DBG")) : 0;   // tuning switches (0 in production)
        q.dbg = dbg;
      }
      if (vspconv::smallmap_eligible(q, p.transposed != 0)) return vspconv::smallmap_launch(q, vsp::as_stream(stream));
      VSP_REQUIRE(!named, "conv2d: configuration smallmap does not fit this problem (Cin %% 16 == 0, at most 8192 output positions, not transposed)");
    }
  }
  Plan best{};
  bool found = false;
  if (p.tile_hint > 0) {
    VSP_REQUIRE(p.tile_hint <= kNumCfgs, "conv2d: tile_hint %d out of range", p.tile_hint);
    found = make_plan(p, p.tile_hint - 1, &best);
    VSP_REQUIRE(found, "conv2d: configuration %s does not fit this problem", kCfgs[p.tile_hint - 1].name);
  } else if (p.tile_hint < 0 && -p.tile_hint <= kNumCfgs) {  // (-(kNumCfgs + 1), not eligible: the cost model decides)
    // a PREFERENCE (tuned table): the table is keyed by geometry only, the same shape may come with an operand this
    // configuration cannot serve (e.g. an input shift with the LDS-DMA staging) -- then the cost model decides
    found = make_plan(p, -p.tile_hint - 1, &best);
  }
  if (!found) {
    double best_cost = 0.0;
    for (int c = 0; c < kNumCfgs; ++c) {
      Plan pl{};
      // the fused dilation-group and the pipelined kernels are only used when named (tile_hint / tuned table); the data gradient of
      // the dilation groups exists only there: its configurations are tried in table order
      if (p.dil_by_input_quarter ? !is_kg(kCfgs[c]) : (is_dg(kCfgs[c]) || kCfgs[c].PF == 3)) continue;
      if (!make_plan(p, c, &pl)) continue;
      const double cost = p.dil_by_input_quarter ? (double)c : plan_cost(p, pl);
      if (!found || cost < best_cost) {
        best = pl;
        best_cost = cost;
        found = true;
      }
    }
    if (!found) return vsp::fail(VSP_ENOTSUP, "conv2d: no tile configuration fits (KH=%d dil=%d)", p.KH, p.dil[0]);
  }
  const Cfg& k = kCfgs[best.cfg];
  const int CO_T = 16 * k.MB * k.WM;
  const int64_t gy = best.dg ? best.co_tiles : (int64_t)best.co_tiles * p.G;
  VSP_REQUIRE(gy <= 65535 && p.B <= 65535, "conv2d: grid too large");

  ConvK q{};
  if (int rc = fill_convk(p, x_ch, q)) return rc;
  q.tw_log2 = best.tw_log2; q.th = best.th; q.tiles_x = best.tiles_x; q.tiles_y = best.tiles_y;
  q.strip_col = best.strip_col;
  q.co_tiles = best.co_tiles;
  q.w_vec4 = (p.cout_g % 4 == 0) && (CO_T % 4 == 0) && vsp::aligned16(p.w) ? 1 : 0;
  q.ps_odd = !p.transposed && p.stride_x != 1;
  {
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = vsp::tune_env("VSP_CONV_DBG");
      dbg = e ? atoi(e) : 0;
    }
    q.dbg = dbg;
  }

  if (k.PF == 3) {  // conv_pipe.hip: plane pitch from the plan, per-channel input scale / shift as pointer + stride
    const float* kc = device_consts();
    q.bf_plane = best.ps;
    q.wcp = p.in_scale ? p.in_scale : kc;      q.wc_cs = p.in_scale ? 1 : 0;  q.wc_bs = p.in_scale ? p.in_scale_bstride : 0;
  }
  if (best.lds > 64 * 1024) {
    static vsp::LdsAttrOnce raised[kMaxCfgs];   // per configuration and device
    if (int rc = raised[best.cfg].ensure(reinterpret_cast<const void*>(k.kern), (int)(k.PF == 3 ? kMaxLdsPipe : kMaxLds), "conv2d")) return rc;
  }
  q.wg_order = (q.dbg & 0x1000000) ? 0 : 1;   // XCD-aware order (conv_kernel.h); VSP_CONV_DBG = 16777216 keeps the dispatch order
  if (k.PF == 3 && p.G == 1 && q.wg_order == 1 && !(q.dbg & 0x20000)) {   // conv_pipe.hip: channel-tile groups on weight-heavy layers
    const int64_t wtile = (int64_t)p.Cin * p.KH * p.KW * CO_T * 4, wall = wtile * best.co_tiles;
    const int64_t xall = (int64_t)p.B * p.Cin * p.H * p.W * 4;
    if (wall > 16 * 1024 * 1024 && best.co_tiles >= 4 && wall * 2 > xall / 8) {
      int cgs = (int)((int64_t)(2 * 1024 * 1024) / wtile);
      q.wg_cgs = cgs < 1 ? 1 : (cgs > best.co_tiles ? best.co_tiles : cgs);
      q.wg_order = 2;
    }
  }
  dim3 grid((unsigned)(best.tiles_x * best.tiles_y + (best.strip_col > 0 ? best.strip_col + best.strip_row : 0)), (unsigned)gy,
            (unsigned)p.B);
  dim3 block(64 * k.WM * k.WN * k.WK);
  hipLaunchKernelGGL(k.kern, grid, block, best.lds, vsp::as_stream(stream), q);
  return vsp::check_launch("conv2d");
}

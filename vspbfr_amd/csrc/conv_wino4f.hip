// Winograd F(4x4, 3x3) on fp32 MFMA, FUSED "lane-owner" form for the shallow wide layers (round 5): 64 -> 64 at 512^2, 32 -> 32 at 1024^2,
// 64 -> 64 at 128^2 / 256^2 (reference e4e/models/stylegan2/model.py:268-276, models/RestoreNet.py:421-555).
//
// Why a third F(4x4)/F(2x2) variant.  The two-kernel F(4x4) pair (conv_wino4.hip) writes V = B^T d B to HBM: 2.25 x the input out and back
// in, which only pays where >= 4 channel tiles share one transform -- on a 64-channel layer V is 1.2 GB for 0.5 GB of input.  The F(2x2)
// row-owner kernel serves these layers at 53-57 % pipe busy with 16/36 of the direct count to execute; F(4x4) executes 36/144.
// Here the transform runs in REGISTERS, in exactly the fragment layout the MFMA wants, with no exchange at all:
//   * v_mfma_f32_16x16x4_f32 with A = U (M = 16 output channels), B = V (N = 16 tiles, K = 4 input channels): the B operand of lane
//     (tile n = lane & 15, channel kq = lane >> 4) is V[position][ci = 4 ks + kq][tile n].  So a lane loads the 6 x 6 window of ITS tile and
//     ITS channel (one aligned 16-byte quad per row; the two outer columns come from the neighbour lanes by DPP row shifts, the N-block's own
//     outer columns by one masked 4-byte load), runs B^T d B on it (144 FMAs, every constant dyadic: points 0, +-3/4, +-3/2, inf as in
//     conv_wino4.hip) and holds all 36 B operands of k-step ks.  No V image, neither in HBM nor in LDS.
//   * the accumulators of a lane are (tile n, 4 output channels) x 36 positions: A^T M A is in-lane as well, and the 4 x 4 outputs of a
//     tile row are 16-byte stores that 16 neighbouring lanes join to 256-byte runs.
//   * price: 36 positions x 2 channel blocks x 4 = 288 accumulator registers per wave -> ONE wave per SIMD (4-wave workgroup, one per CU,
//     persistent).  Nothing hides a stall, so everything is software-pipelined by hand: window loads two k-steps ahead, the transform of
//     k-step ks + 1 between the MFMAs of k-step ks (2 VALU per MFMA), U through a three-slab LDS ring (one barrier per k-step of 72 MFMAs).
//   * U (36 x 32 x 4 floats per k-step and 32-channel half = 18 KB) is shared by the four waves (four N-blocks of 16 tiles, stacked
//     vertically) through LDS; the style scale of the input channel is folded into U while it is staged (the copy goes through registers
//     anyway), so the window path carries no multiply.
// Operand traffic per CU and k-step (2304 pipe cycles): U 18 KB + windows 4 x 6.1 KB = 18.5 B/clk from L2 (the pair's GEMM: 24).
//
//   U4F [co half (32)][k-step][pp 18][lane 64][4]      lane = (kq, lr): ci = 4 ks + kq; element e: position 2 pp + (e >> 1), co = 32 half + 16 (e & 1) + lr
#include "conv_kernel.h"
#include <type_traits>
#include <cstdio>

namespace vspconv {

namespace {

typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef unsigned u32x4f __attribute__((__vector_size__(4 * sizeof(unsigned))));
typedef float f32x2f __attribute__((ext_vector_type(2)));

constexpr float FA = 0.75f, FB = 1.5f, FA2 = 0.5625f, FB2 = 2.25f, FA2B2 = 1.265625f, FSUM2 = 2.8125f, FA3 = 0.421875f, FB3 = 3.375f;
constexpr int F4_THR = 256;
constexpr int F4_SLAB = 18 * 64 * 4;          // floats of one (half, k-step) slab in HBM
constexpr int F4_SLABL = 5 * F4_THR * 4;      // floats of one LDS ring slot: five 16-byte chunks per thread (the last half-round lands in the slot's tail)
constexpr int F4_RING = 3;
constexpr int F4_MAXC = 512;                  // input channels of the scale table
constexpr int F4_LDS = F4_RING * F4_SLABL + F4_MAXC + 32 * 4;
constexpr int F4_OOB = 0x7ffffff0;

struct F4Plan {
  int nwg, items, J, nbx, nbyg, nco2, nks;      // nwg: workgroups that share these items (the launch, or one group's partition of it)
};
struct F4PlanG {
  F4Plan g[4];
};

// ------------------------------------------------------------------------------------------------------------ weights: U = G g G^T
__global__ __launch_bounds__(256) void wino4f_weight_kernel(float* __restrict__ U, const float* __restrict__ wp, int cin, int cout, int nks, int64_t units) {
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= units) return;
  const int lane = threadIdx.x & 63, kq = lane >> 4, lr = lane & 15;
  const int ks = (int)(unit % nks);
  const int half = (int)(unit / nks);
  const int ci = 4 * ks + kq;
  const double Gm[6][3] = {{64.0 / 81.0, 0.0, 0.0},           {-128.0 / 243.0, -32.0 / 81.0, -8.0 / 27.0}, {-128.0 / 243.0, 32.0 / 81.0, -8.0 / 27.0},
                           {32.0 / 243.0, 16.0 / 81.0, 8.0 / 27.0}, {32.0 / 243.0, -16.0 / 81.0, 8.0 / 27.0},  {0.0, 0.0, 1.0}};
  float u[2][36];
  for (int cb = 0; cb < 2; ++cb) {
    const int co = 32 * half + 16 * cb + lr;
    const bool in = ci < cin && co < cout;
    double gk[3][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) gk[tap / 3][tap % 3] = in ? (double)wp[((int64_t)tap * cin + ci) * cout + co] : 0.0;
    double r[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int x = 0; x < 3; ++x) r[i][x] = Gm[i][0] * gk[0][x] + Gm[i][1] * gk[1][x] + Gm[i][2] * gk[2][x];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) u[cb][6 * i + j] = (float)(r[i][0] * Gm[j][0] + r[i][1] * Gm[j][1] + r[i][2] * Gm[j][2]);
  }
  f32x4f* dst = reinterpret_cast<f32x4f*>(U + unit * (int64_t)F4_SLAB) + lane;
#pragma unroll
  for (int pp = 0; pp < 18; ++pp) dst[pp * 64] = f32x4f{u[0][2 * pp], u[1][2 * pp], u[0][2 * pp + 1], u[1][2 * pp + 1]};
}

// The 288 accumulator registers of a wave do not fit the 256 AccVGPRs: positions 0..31 live there, positions 32..35 in architectural
// VGPRs.  hipcc picks ONE register file for every MFMA of a kernel and then shuffles the overflow through v_accvgpr moves and scratch
// (first build: 668 moves + 60 scratch accesses per two k-steps), so the MFMAs are written out with explicit constraints.  The compiler
// does not see an MFMA here: the only hazards left to cover by hand are the reads of the accumulators in the epilogue (s_nop below);
// the A / B operands are produced a k-step (V) or two units (U, LDS) ahead and waited for like any other inline-asm operand.
#define F4_MFMA_A(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define F4_MFMA_V(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// LDS hand-over barrier: the LDS queue drained, NOT the vector-memory queue (__syncthreads waits for every window load in flight)
#define F4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- LDS window loader (template D >= 1; D = 0 is the register loader above).  The window stream of a workgroup goes global -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds: no register, no vector instruction -- beside fp32 MFMAs every vector instruction costs pipe time)
// and the lanes read their 6 x 6 window from there (LDS reads are free beside the MFMAs).  That also opens the DILATED layers: with
// dilation D the convolution is D x D independent dense convolutions on the polyphase sub-images, a lane's tile = 4 x 4 outputs D apart,
// its window = 6 x 6 inputs D apart -- strided in HBM (4-byte loads cost twice the vector-memory time of 16-byte ones and the kernel
// would be bound there), but any stride is one ds_read_b32 in LDS.  A workgroup item is still a 16-row x 64-column output region of
// one image: its D^2 phases hold (16 / D / 4) x (64 / D / 4) tiles each, 64 in all = 4 waves x 16 lanes.
//   region in LDS per k-step (4 channels): [channel][row -D .. 16 + D)[column -HQ .. 64 + HQ)  (HQ = max(4, D): whole 16-byte quads),
//   written in that linear order by NI DMA instructions per wave (lane l of instruction n of wave w = quad (4 n + w) 64 + l; quads outside
//   the image or the region have an out-of-range offset: the DMA writes zeros there -- tools/ubench/lds_dma_oob.hip -- the padding for free).
//   Three slots: window ks + 4 is requested behind MFMA 53 of k-step ks, awaited (counted vmcnt) before the barrier of k-step ks + 1, read
//   behind MFMAs 37 .. 55 of k-step ks + 2, a row at a time.  (Dilation 8, two slots: requested behind the barrier of k-step ks instead -- into the slot read in that k-step.)
template <int D>
struct F4G {
  static constexpr int DD = D == 0 ? 1 : D;
  static constexpr int HQ = DD < 4 ? 4 : DD;
  static constexpr int RH = DD == 8 ? 32 : 16, RW = DD == 8 ? 32 : 64;   // the item's output region (dilation 8: a tile spans 32 rows; 64 phases x one tile)
  static constexpr int NR = RH + 2 * DD, RWP = RW + 2 * HQ, NQ = RWP / 4;
  static constexpr int PLANE = (NR * RWP + 15) / 16 * 16 + (DD == 8 ? 8 : 4);   // floats per channel; the four channels of a k-step start 4 (8) banks apart
  static constexpr int PLANEQ = PLANE / 4;
  static constexpr int NI = (4 * PLANEQ + 255) / 256;          // DMA instructions per wave and k-step
  static constexpr int SLOT = NI * 256 * 4;                     // floats of one window slot
  static constexpr int WS = DD == 8 ? 2 : 3;                    // window slots (dilation 8: 40 KB each -- two, and the item's third window is awaited in the prologue)
  static constexpr int LDS = D == 0 ? F4_LDS : F4_LDS + WS * SLOT;
};
// one DMA instruction: 64 lanes x 16 bytes from (descriptor, lane byte offset voff + scalar offset soff) to LDS byte address ldsb + 16 lane.
// hipcc does not see it (it would order every LDS read of the k-step behind a DMA it knows of): its completion is counted by hand below.
#define F4_DMA(voff, rsrc, soff, ldsb)                                                                                              \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"   \
               : "=&s"(m0keep) : "v"(voff), "s"(rsrc), "s"(ldsb), "s"(soff) : "memory")

#ifdef VSP_F4_TRACE   // tuning only: shader-clock stamps of one workgroup's waves inside its third item (tools/build_abl.sh conv_wino4f.hip VSP_F4_TRACE f4trace)
__device__ unsigned long long f4_trace_buf[4 * 64];
// (stamps are kept in scalar registers and written once at the end of the item: a store behind a branch per stamp is a join, and a join
//  waits for every store in flight -- the first trace measured the store round trip, not the epilogue)
#define F4_STAMP(idx) do { stv[idx] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define F4_STAMP(idx) do {} while (0)
#endif

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, i>) for i in [B, E): every index below is a compile-time constant
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

#ifdef VSP_F4_ABL   // tuning builds only (tools/build_abl.sh conv_wino4f.hip VSP_F4_ABL=<bits> <name>): 1 window loads from one 16 KB region,
constexpr int f4ab = VSP_F4_ABL;   // 2 no U staging after the prologue, 4 no MFMAs, 8 no transform, 16 no epilogue, 32 no barrier in the k-step, 64 epilogue without its stores
#else
constexpr int f4ab = 0;
#endif

// Packed fp32 VALU with explicit half selection (asm: the stream beside the MFMAs is laid out by hand, and hipcc has no builtin for the
// op_sel forms).  D = A * B + C on register PAIRS; each source's low-lane / high-lane operand is its pair's half `lo` / `hi` (0 = .x, 1 = .y).
#define F4_PK(d, a, b, c, MODS) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define F4_SEL_AX " op_sel:[0,0,0] op_sel_hi:[0,1,1]"        // A broadcast .x, B and C as they lie
#define F4_SEL_AY " op_sel:[1,0,0] op_sel_hi:[1,1,1]"        // A broadcast .y
#define F4_SEL_AXN " op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"   // A broadcast .x, negated
#define F4_SEL_AYN " op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"

// RES: residual operands present; ACT1: first activation present (compile-time: a branch in the epilogue is a join, and hipcc waits for
// every store in flight at a join -- the first build spent 43 % of its time there)
// g: the workgroup's index among the pl.nwg that share the plan's items; grp: the dilation group (channels grp * cout_g ... of every per-channel
// operand, of y and of the residuals; weight slabs of that group) -- 0 for a one-group layer
template <bool RES, bool ACT1, int D>
__device__ __forceinline__ void f4_body(const ConvK& p, const F4Plan& pl, float* smem, const int g, const int grp) {
  using GE = F4G<D>;
  constexpr int DD = GE::DD;
  float* Sc = smem + F4_RING * F4_SLABL;
  float* Et = Sc + F4_MAXC;
  float* Wr = Et + 32 * 4;               // (D >= 1) window ring: three slots of GE::SLOT floats
  typedef __attribute__((address_space(3))) float lds_f;
  typedef __attribute__((address_space(3))) f32x4f lds_f4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chw = p.H * p.W;
  const int nks = pl.nks;
  const int Cout = p.cout_g, cgo = grp * p.cout_g, Ctot = p.G * p.cout_g;
  const int y_plane = p.y_h * p.y_w;
  const float nw = p.nwp[0];
  const int slot = (g & 7) * (pl.nwg >> 3) + (g >> 3);      // workgroups of one XCD (g % 8) walk neighbouring items
  const int it0 = slot * pl.J, it1 = min(pl.items, it0 + pl.J);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w) + (int64_t)grp * pl.nco2 * nks * F4_SLAB, 0, pl.nco2 * nks * F4_SLAB * 4, 0x00020000);
  lds_f4* const Ul4 = (lds_f4*)(lds_f*)smem;
  // constant pairs of B^T (conv_wino4.hip header): C1 = (-b2, -a2), C2 = (-(a2 + b2), a2 b2), C3 = (a, b)
  const f32x2f C1 = {-FB2, -FA2}, C2 = {-FSUM2, FA2B2}, C3 = {FA, FB};

  for (int it = it0; it < it1; ++it) {
#ifdef VSP_F4_TRACE
    const bool trace_on = g == 40 && it == it0 + 2;
    unsigned stv[52];
#endif
    F4_STAMP(0);
    // (lane-dependent values are re-derived per item from an opaque copy of the lane index: computed once at kernel entry they are spilled
    //  around the k-loop, and a reload -- a scratch load -- waits for every store of the previous item)
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i));
    const int lr = lane_i & 15, kq = lane_i >> 4;
    const int half = it % pl.nco2;
    int t = it / pl.nco2;
    const int bx = t % pl.nbx;
    t /= pl.nbx;
    const int byg = t % pl.nbyg;
    const int b = t / pl.nbyg;
    // the lane's tile: phase (py, px) of the region, tile (ty, tx) inside the phase -- first output (y0, xq), outputs DD apart
    const int ry = GE::RH * byg, x0 = GE::RW * bx;
    const int y0 = DD == 8 ? ry + 2 * wave + (lr >> 3) : ry + (wave & (DD - 1)) + 4 * DD * (wave / DD);
    const bool wave_ok = DD == 8 ? ry < p.H : y0 < p.H;       // (dilation 8: H is a multiple of the 32-row region; otherwise y0 is the wave's)
    const int xq = DD == 8 ? x0 + (lr & 7) : x0 + (lr & (DD - 1)) + 4 * DD * (lr / DD);
    const float* xb = p.x + (int64_t)b * p.x_ch * chw;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.Cin * chw * 4, 0x00020000);
    // Window loads: one aligned 16-byte quad per row (window columns 1..4 = the tile's own pixels) + one 4-byte load per row that only
    // lanes 0 / 15 of a 16-lane row perform (the N-block's outer columns; every other lane gets column 0 / 5 from its neighbour by DPP).
    // Rows 1..4 take their row through the scalar offset; rows 0 and 5 (one up / four down) have their own lane offsets, out of range at
    // the image border.
    const int rowb = (f4ab & 1) ? 256 : p.W * 4;
    const bool colok = xq < p.W;
    const int vq = (wave_ok && colok) ? ((f4ab & 1) ? (kq * 1024 + 8 * 64 + lr * 4) * 4 : (kq * chw + y0 * p.W + xq) * 4) : F4_OOB;
    const int vq0 = (wave_ok && colok && y0 > 0) ? vq - rowb : F4_OOB;
    const int vq5 = (wave_ok && colok && y0 + 4 < p.H) ? vq + 4 * rowb : F4_OOB;
    const int xh = lr == 0 ? x0 - 1 : x0 + 64;
    const bool hok = wave_ok && (lr == 0 ? x0 > 0 : (lr == 15 && xh < p.W));
    const int vh = hok ? ((f4ab & 1) ? (kq * 1024 + 8 * 64 + lr) * 4 : (kq * chw + y0 * p.W + xh) * 4) : F4_OOB;
    const int vh0 = (hok && y0 > 0) ? vh - rowb : F4_OOB;
    const int vh5 = (hok && y0 + 4 < p.H) ? vh + 4 * rowb : F4_OOB;

    f32x4f Rq[6];          // window rows: columns 1..4
    float Rh[6];           //              the N-block's outer column (lanes 0 / 15 of a 16-lane row)
    f32x2f Wp[6][3];       // (D >= 1)     column pairs (0, 1), (2, 3), (4, 5)
    // ---- LDS loader: the lane's DMA source offsets (constant over the k-steps), its window base in a slot, the descriptor in scalar registers
    int dmo[GE::NI];
    const lds_f* Wl = nullptr;
    u32x4f xdesc = {0u, 0u, 0u, 0u};
    unsigned wr_base = 0;
    if constexpr (D >= 1) {
#pragma unroll
      for (int n = 0; n < GE::NI; ++n) {
        const int L = (4 * n + wave) * 64 + lane_i;
        const int ch = L / GE::PLANEQ, rem = L - ch * GE::PLANEQ, row = rem / GE::NQ, cq = rem - row * GE::NQ;
        const int gy = ry - DD + row, gx = x0 - GE::HQ + 4 * cq;
        const bool ok = ch < 4 && row < GE::NR && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        dmo[n] = ok ? (ch * chw + gy * p.W + gx) * 4 : F4_OOB;
      }
      Wl = (const lds_f*)Wr + kq * GE::PLANE + (y0 - ry) * GE::RWP + (xq - x0) + GE::HQ - DD;
      const uint64_t xa = reinterpret_cast<uint64_t>(xb);
      xdesc = u32x4f{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(xa >> 32) & 0xffffu)),
                     (unsigned)__builtin_amdgcn_readfirstlane(p.Cin * chw * 4), 0x00020000u};
      wr_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_f*)Wr);
    }
    auto dma_w = [&](int ksw) {       // window ksw -> ring slot ksw % WS (k-steps past the end re-request the last one)
      const int so = __builtin_amdgcn_readfirstlane(min(ksw, nks - 1) * 16 * chw);
      const unsigned lb = (unsigned)__builtin_amdgcn_readfirstlane((int)(wr_base + ((ksw % GE::WS) * GE::SLOT + wave * 256) * 4));
      static_for<0, GE::NI>([&](auto Nc) {
        constexpr int n = decltype(Nc)::value;
        const int vo = dmo[n], so_l = so;
        const u32x4f rd = xdesc;
        const unsigned lbn = lb + n * 4096;
        unsigned m0keep;
        F4_DMA(vo, rd, so_l, lbn);
      });
    };
    auto read_row = [&](int ksw, auto Ic) {      // row i of the lane's window of k-step ksw from its slot
      constexpr int i = decltype(Ic)::value;
      const lds_f* s = Wl + (ksw % GE::WS) * GE::SLOT;
      // (column pairs (0, 1), (2, 3), (4, 5): what hipcc's ds_read2_b32 merging yields anyway -- and the column pass below is six packed FMAs on them)
      const lds_f* r = s + i * DD * GE::RWP;
      Wp[i][0] = f32x2f{r[0], r[DD]};
      Wp[i][1] = f32x2f{r[2 * DD], r[3 * DD]};
      Wp[i][2] = f32x2f{r[4 * DD], r[5 * DD]};
    };
    auto read_w = [&](int ksw) { static_for<0, 6>([&](auto Ic) { read_row(ksw, Ic); }); };
    auto load_w = [&](int so_k) {
      Rq[0] = __builtin_bit_cast(f32x4f, __builtin_amdgcn_raw_buffer_load_b128(xrs, vq0, so_k, 0));
      Rh[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vh0, so_k, 0));
#pragma unroll
      for (int r = 1; r < 5; ++r) {
        Rq[r] = __builtin_bit_cast(f32x4f, __builtin_amdgcn_raw_buffer_load_b128(xrs, vq, so_k + (r - 1) * rowb, 0));
        Rh[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vh, so_k + (r - 1) * rowb, 0));
      }
      Rq[5] = __builtin_bit_cast(f32x4f, __builtin_amdgcn_raw_buffer_load_b128(xrs, vq5, so_k, 0));
      Rh[5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vh5, so_k, 0));
    };
    // The transform V = B^T d B in PACKED arithmetic.  First B^T down the window rows, on column PAIRS as the quads hold them:
    // X = 0: (c1, c2) = q.xy, 1: (c3, c4) = q.zw, 2: (c0, c5) built by two DPP moves per row.  12 packed FMAs per pair -> Xt[X][0..5].
    f32x2f Xt[3][6];
    auto rowpass = [&](auto Xc) {
      constexpr int X = decltype(Xc)::value;
      const f32x2f c1 = C1, c2 = C2, c3 = C3;       // (locals: clang rejects an asm operand that names a captured variable inside a generic lambda)
      f32x2f w[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const f32x4f q = Rq[i];
        if constexpr (D >= 1) w[i] = Wp[i][X];
        if constexpr (D == 0 && X == 0) w[i] = f32x2f{q[0], q[1]};
        if constexpr (D == 0 && X == 1) w[i] = f32x2f{q[2], q[3]};
        if constexpr (D == 0 && X == 2) {
          const float hh = Rh[i];
          int lo = __builtin_bit_cast(int, hh), hi = lo;
          const float qwf = q[3], qxf = q[0];      // (hipcc: a bit_cast of a vector ELEMENT expression reads element 0 -- scalars first)
          const int qw = __builtin_bit_cast(int, qwf), qx = __builtin_bit_cast(int, qxf);
          asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(lo) : "v"(qw));   // column 0 <- neighbour's column 4 (lane 0 of a row keeps the loaded one)
          asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(hi) : "v"(qx));   // column 5 <- neighbour's column 1 (lane 15 keeps the loaded one)
          w[i] = f32x2f{__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)};
        }
      }
      const f32x2f w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5];
      f32x2f e1, o1, e2, o2, n0, n5, t0, t1, t2, t3, t4, t5;
      F4_PK(e1, c1, w2, w4, F4_SEL_AX);
      F4_PK(o1, c1, w1, w3, F4_SEL_AX);
      F4_PK(e2, c1, w2, w4, F4_SEL_AY);
      F4_PK(o2, c1, w1, w3, F4_SEL_AY);
      F4_PK(n0, c2, w2, w4, F4_SEL_AX);
      F4_PK(n5, c2, w3, w5, F4_SEL_AX);
      F4_PK(t0, c2, w0, n0, F4_SEL_AY);
      F4_PK(t5, c2, w1, n5, F4_SEL_AY);
      F4_PK(t1, c3, o1, e1, F4_SEL_AX);
      F4_PK(t2, c3, o1, e1, F4_SEL_AXN);
      F4_PK(t3, c3, o2, e2, F4_SEL_AY);
      F4_PK(t4, c3, o2, e2, F4_SEL_AYN);
      Xt[X][0] = t0; Xt[X][1] = t1; Xt[X][2] = t2; Xt[X][3] = t3; Xt[X][4] = t4; Xt[X][5] = t5;
    };
    // Then B^T along transformed row i: d0 = H.x, (d1, d2) = A, (d3, d4) = B, d5 = H.y.  Four packed FMAs with half selection give
    // (e1, e2), (o1, o2), (V1, V2), (V3, V4); V0 and V5 mix halves of different pairs: four scalar FMAs.  -> T[i]: {V0, V5}, {V1, V2}, {V3, V4}
    auto colpass = [&](auto Ic, f32x2f (&T)[6][3]) {
      constexpr int i = decltype(Ic)::value;
      const f32x2f c1 = C1, c3 = C3;
      const float fs = -FSUM2, fab = FA2B2;
      f32x2f e12, o12, v12, v34;
      if constexpr (D >= 1) {
        // LDS loader: the pairs are (d0, d1), (d2, d3), (d4, d5) -- (V0, V5) = a2b2 (d0, d1) - (a2 + b2) (d2, d3) + (d4, d5) is two packed FMAs as well
        const f32x2f c2 = C2;
        const f32x2f P01 = Xt[0][i], P23 = Xt[1][i], P45 = Xt[2][i];
        f32x2f n05, v05;
        F4_PK(n05, c2, P23, P45, F4_SEL_AX);
        F4_PK(e12, c1, P23, P45, " op_sel:[0,0,0] op_sel_hi:[1,0,0]");         // (-b2, -a2) d2 + d4
        F4_PK(o12, c1, P01, P23, " op_sel:[0,1,1] op_sel_hi:[1,1,1]");         // (-b2, -a2) d1 + d3
        F4_PK(v05, c2, P01, n05, F4_SEL_AY);
        F4_PK(v12, c3, o12, e12, " op_sel:[0,0,0] op_sel_hi:[0,0,0] neg_hi:[1,0,0]");
        F4_PK(v34, c3, o12, e12, " op_sel:[1,1,1] op_sel_hi:[1,1,1] neg_hi:[1,0,0]");
        T[i][0] = v05; T[i][1] = v12; T[i][2] = v34;
        return;
      }
      const f32x2f A = Xt[0][i], B = Xt[1][i], H = Xt[2][i];
      const float d0 = H[0], d1 = A[0], d2 = A[1], d3 = B[0], d4 = B[1], d5 = H[1];
      float n0, n5, v0, v5;
      F4_PK(e12, c1, A, B, " op_sel:[0,1,1] op_sel_hi:[1,1,1]");          // (-b2, -a2) d2 + d4
      F4_PK(o12, c1, A, B, " op_sel:[0,0,0] op_sel_hi:[1,0,0]");          // (-b2, -a2) d1 + d3
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(n0) : "v"(fs), "v"(d2), "v"(d4));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(n5) : "v"(fs), "v"(d3), "v"(d5));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v0) : "v"(fab), "v"(d0), "v"(n0));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v5) : "v"(fab), "v"(d1), "v"(n5));
      F4_PK(v12, c3, o12, e12, " op_sel:[0,0,0] op_sel_hi:[0,0,0] neg_hi:[1,0,0]");   // e1 +- a o1
      F4_PK(v34, c3, o12, e12, " op_sel:[1,1,1] op_sel_hi:[1,1,1] neg_hi:[1,0,0]");   // e2 +- b o2
      T[i][0] = f32x2f{v0, v5}; T[i][1] = v12; T[i][2] = v34;
    };
    // the B operand of Winograd position 6 i + j
    auto vpos = [&](f32x2f (&T)[6][3], int pos) -> float {
      const int i = pos / 6, j = pos % 6;
      return j == 0 ? T[i][0][0] : (j == 5 ? T[i][0][1] : (j == 1 ? T[i][1][0] : (j == 2 ? T[i][1][1] : (j == 3 ? T[i][2][0] : T[i][2][1]))));
    };

    // ---- per-item tables: style scale per input channel, epilogue operands of this half's 32 output channels
    __syncthreads();   // (the previous item's epilogue read Et)
    for (int c = tid; c < p.Cin; c += F4_THR) Sc[c] = p.wtp[(int64_t)b * p.wt_bs + (int64_t)c * p.wt_cs];
    if (tid >= 64 && tid < 96) {
      const int j = tid - 64;
      const int cgi = 32 * half + j;
      const int cg = cgo + (cgi < Cout ? cgi : Cout - 1);
      const float os = p.osp[((int64_t)b * Ctot + cg) * p.oss], cs = p.csp[cg * p.css];
      *reinterpret_cast<f32x4f*>(Et + 4 * j) = f32x4f{os * cs, p.cbp[cg * p.cbs] + p.b1p[cg * p.b1s], p.b2p[cg * p.b2s], p.s2p[cg * p.s2s]};
    }
    __syncthreads();

    // ---- U staging: slab (half, ks) -> ring slot ks % 3, five 16-byte chunks per thread, scaled by the channel's style factor
    f32x4f ust[5];
    auto u_load = [&](int ks) {
      const int so = __builtin_amdgcn_readfirstlane((half * nks + ks) * (F4_SLAB * 4));
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int c = tid + F4_THR * r;
        ust[r] = __builtin_bit_cast(f32x4f, __builtin_amdgcn_raw_buffer_load_b128(urs, c < F4_SLAB / 4 ? c * 16 : F4_OOB, so, 0));
      }
    };
    auto u_scale = [&](auto Cc, float sc) {     // (two packed multiplies per chunk)
      constexpr int c = decltype(Cc)::value;
      const f32x4f u = ust[c];
      f32x2f lo = {u[0], u[1]}, hi = {u[2], u[3]};
      const f32x2f s2 = {sc, sc};
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(s2));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(s2));
      ust[c] = f32x4f{lo[0], lo[1], hi[0], hi[1]};
    };

    f32x4f accA[32][2], accV[4][2];
#pragma unroll
    for (int q = 0; q < 32; ++q) accA[q][0] = accA[q][1] = f32x4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) accV[q][0] = accV[q][1] = f32x4f{0.f, 0.f, 0.f, 0.f};
    f32x2f Ta[6][3], Tb[6][3];
    f32x4f uf[3];

    // ---- prologue: window 0 -> Ta, window 1 in flight, U(0) in slot 0, U(1) in the staging registers
    F4_STAMP(1);
    if constexpr (D >= 1) {
      dma_w(0);
      dma_w(1);
      if constexpr (GE::WS == 3) dma_w(2);
    } else {
      load_w(0);
    }
    u_load(0);
    {
      const float sc = ((lds_f*)Sc)[kq];
      static_for<0, 5>([&](auto Cc) { u_scale(Cc, sc); });
      lds_f4* dst = Ul4 + tid;
#pragma unroll
      for (int r = 0; r < 5; ++r) dst[F4_THR * r] = ust[r];
    }
    u_load(nks > 1 ? 1 : 0);
    if constexpr (D >= 1) {
      // (the staging of U(0) above waited for its loads: the DMA requests are older, hence landed; the barrier publishes all three windows)
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      F4_BARRIER();
      read_w(0);
    }
    static_for<0, 3>([&](auto Xc) { rowpass(Xc); });
    if constexpr (D >= 1) read_w(1);
    else load_w((f4ab & 1) ? 0 : __builtin_amdgcn_readfirstlane((nks > 1 ? 1 : 0) * 16 * chw));
    static_for<0, 6>([&](auto Ic) { colpass(Ic, Ta); });
    F4_BARRIER();
    if constexpr (D >= 1 && GE::WS == 3) dma_w(3);      // (slot 0: every wave has read window 0)
    if constexpr (D >= 1 && GE::WS == 2) {              // (both slots are free now: window 2 is read in the first k-step -- awaited here)
      dma_w(2);
      dma_w(3);
      asm volatile("s_waitcnt vmcnt(%0)" :: "i"(GE::NI) : "memory");
      F4_BARRIER();
    }
    {
      const lds_f4* Us = Ul4 + lane;
      uf[0] = Us[0];
      uf[1] = Us[64];
    }

    // ---- one k-step: 72 MFMAs (position pair pp x 2 channel blocks).  fp32 MFMAs and fp32 VALU share the SIMD's FMA lanes: beside a
    //      v_mfma_f32_16x16x4_f32 NO vector instruction is hidden (tools/ubench/mfma_valu_gap.hip, one wave per SIMD: + 5.5 cycles per VALU
    //      instruction in a run, + 13 for a lone one, packed or not; loads, LDS and scalar instructions + 0.3), so the k-step is
    //      72 x 32.5 cycles + what its VALU instructions cost.  Hence: the transform as 72 PACKED FMAs (144 scalar ones + 12 DPP moves in the
    //      first form): 36 packed FMAs + 12 DPP moves for B^T down the rows, 24 packed + 24 scalar FMAs for B^T along them, in six runs
    //      behind every twelfth MFMA; the U scaling as ten packed multiplies in three of the runs.
    //        run 0..2: row pass of the column pairs (c0, c5), (c1, c2), (c3, c4) of window ks + 1; after run 2 the window registers are
    //                  re-loaded with window ks + 2
    //        run 3..5: column pass of transformed rows (0,1), (2,3), (4,5) -> Vn
    //        U: chunks scaled in runs 1..3 (packed multiplies), written two MFMAs later, their registers re-loaded with U(ks + 2)
    //        U fragments: one 16-byte LDS read per four MFMAs, two position pairs ahead; barrier after MFMA 56
    auto kstep = [&](int ks, f32x2f (&Vc)[6][3], f32x2f (&Vn)[6][3]) {
      const lds_f4* Us = Ul4 + (ks % F4_RING) * (F4_SLABL / 4) + lane;
      const lds_f4* Un = Ul4 + ((ks + 1) % F4_RING) * (F4_SLABL / 4) + lane;
      const int k1 = min(ks + 1, nks - 1), k2 = min(ks + 2, nks - 1);
      const int so_r = (f4ab & 1) ? 0 : __builtin_amdgcn_readfirstlane(k2 * 16 * chw);
      const int so_u = __builtin_amdgcn_readfirstlane((half * nks + k2) * (F4_SLAB * 4));
      lds_f4* ud = Ul4 + (k1 % F4_RING) * (F4_SLABL / 4) + tid;
      float sc = 1.f;
      static_for<0, 72>([&](auto SL) {
        constexpr int sl = decltype(SL)::value;
        constexpr int pp = sl >> 2, e = sl & 3;
        if constexpr (e == 0) {
          if constexpr (pp < 16) uf[(pp + 2) % 3] = Us[(pp + 2) * 64];
          if constexpr (pp == 16) uf[0] = Un[0];          // (after the barrier: the next k-step's first two fragments)
          if constexpr (pp == 17) uf[1] = Un[64];
        }
        {
          const float ua = uf[pp % 3][e];
          const float vb = vpos(Vc, 2 * pp + (e >> 1));
          if constexpr ((f4ab & 4) != 0) {
          } else if constexpr (pp < 16) {
            F4_MFMA_A(accA[2 * pp + (e >> 1)][e & 1], ua, vb);
          } else {
            F4_MFMA_V(accV[2 * pp + (e >> 1) - 32][e & 1], ua, vb);
          }
        }
        if constexpr (sl % 12 == 11) {
          constexpr int run = sl / 12;
          if constexpr (!(f4ab & 8)) {
            if constexpr (run == 0) rowpass(std::integral_constant<int, 2>{});        // (the DPP pairs first: their registers are the first to come free)
            if constexpr (run == 1) rowpass(std::integral_constant<int, 0>{});
            if constexpr (run == 2) rowpass(std::integral_constant<int, 1>{});
            if constexpr (run >= 3) {
              colpass(std::integral_constant<int, 2 * (run - 3)>{}, Vn);
              colpass(std::integral_constant<int, 2 * (run - 3) + 1>{}, Vn);
            }
          }
          if constexpr (run == 2 && D == 0) load_w(so_r);                             // window ks + 2 (the row passes are through with the registers)
          if constexpr (!(f4ab & 2)) {
            if constexpr (run == 0) sc = ((lds_f*)Sc)[4 * k1 + kq];
            if constexpr (run == 1) { u_scale(std::integral_constant<int, 0>{}, sc); u_scale(std::integral_constant<int, 1>{}, sc); }
            if constexpr (run == 2) { u_scale(std::integral_constant<int, 2>{}, sc); }
            if constexpr (run == 3) { u_scale(std::integral_constant<int, 3>{}, sc); u_scale(std::integral_constant<int, 4>{}, sc); }
          }
        }
        if constexpr (!(f4ab & 2)) {
          if constexpr (sl == 26 || sl == 28 || sl == 38 || sl == 50 || sl == 52) {      // write the scaled chunk, re-load its registers with U(ks + 2)
            constexpr int c = sl == 26 ? 0 : (sl == 28 ? 1 : (sl == 38 ? 2 : (sl == 50 ? 3 : 4)));
            ud[F4_THR * c] = ust[c];
            const int ci = tid + F4_THR * c;
            ust[c] = __builtin_bit_cast(f32x4f, __builtin_amdgcn_raw_buffer_load_b128(urs, ci < F4_SLAB / 4 ? ci * 16 : F4_OOB, so_u, 0));
          }
        }
        // (LDS loader) window ks + 2, one row at a time: LDS returns in order -- a U fragment read behind a burst of 36 reads from four waves
        // at once waits for all of them; all rows before the barrier (two-slot ring: the request behind it overwrites this slot)
        if constexpr (D >= 1 && (sl == 37 || sl == 41 || sl == 45 || sl == 49 || sl == 53 || sl == 55))
          read_row(ks + 2, std::integral_constant<int, sl == 55 ? 5 : (sl - 37) / 4>{});
        if constexpr (D >= 1 && GE::WS == 3 && sl == 53) dma_w(ks + 4);
        if constexpr (sl == 56 && !(f4ab & 32)) {
          // (in flight, oldest first: window ks + 3, the five U chunks of this k-step, [three slots: window ks + 4] -- the first must have landed)
          if constexpr (D >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(GE::WS == 3 ? 5 + GE::NI : 5) : "memory");
          F4_BARRIER();
        }
        if constexpr (D >= 1 && GE::WS == 2 && sl == 57) dma_w(ks + 4);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    F4_STAMP(2);
    for (int ks = 0; ks < nks; ks += 2) {
      kstep(ks, Ta, Tb);
      F4_STAMP(3 + ks);
      kstep(ks + 1, Tb, Ta);
      F4_STAMP(4 + ks);
    }

    // ---- epilogue: A^T M A per (channel block, register) in-lane, the fused operand chain, 16-byte stores
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results: up to 18 wait states before a VALU / accvgpr read)
    auto accr = [&](int pos, int cb, int r) -> float { return pos < 32 ? accA[pos][cb][r] : accV[pos - 32][cb][r]; };
    // (every lane-dependent value of the epilogue is derived from an opaque copy of the lane index: computed from `lane` they are
    //  loop-invariant, get hoisted above the k-loop, spilled across it, and each reload -- a scratch load -- waits for every store in flight)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lr_e = lane_e & 15, kq_e = lane_e >> 4;
    const int xq_e = DD == 8 ? x0 + (lr_e & 7) : x0 + (lr_e & (DD - 1)) + 4 * DD * (lr_e / DD);
    if (wave_ok && xq_e < p.W && !(f4ab & 16)) {
      const int ybytes = Cout * y_plane * 4;
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + ((int64_t)b * p.y_ch + p.y_coff + cgo) * y_plane, 0, ybytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t r1rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.r1p + ((int64_t)b * p.res_ch + p.res_coff + cgo) * y_plane * p.r1s), 0, p.r1s ? ybytes : 16, 0x00020000);
      const __amdgpu_buffer_rsrc_t r2rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.r2p + ((int64_t)b * p.res_ch + p.res_coff + cgo) * y_plane * p.r2s), 0, p.r2s ? ybytes : 16, 0x00020000);
      const __amdgpu_buffer_rsrc_t nzrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.nzp + (int64_t)b * p.OH * p.OW * p.nzs), 0,
                                                                            p.nzs ? p.OH * p.OW * 4 : 16, 0x00020000);
      const int y0_e = DD == 8 ? ry + 2 * wave + (lr_e >> 3) : y0;
      const int pix = (y0_e * p.y_w + xq_e) * 4;      // byte offset of the tile's first output inside a channel plane (y_w == OW)
      F4_STAMP(40);
      // a tile row = four outputs DD apart: one 16-byte access at dilation 1, four 4-byte ones otherwise (`mul`: 0 / 1, the operand's stride)
      auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, int off, int mul) -> f32x4f {
        if constexpr (DD == 1) {
          return __builtin_bit_cast(f32x4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off * mul, 0, 0));
        } else {
          f32x4f v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (off + e * DD * 4) * mul, 0, 0));
          return v;
        }
      };
      const int rowb_y = DD * p.y_w * 4;       // bytes between two output rows of a tile
      f32x4f nz[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) nz[i] = ld4(nzrs, pix + i * rowb_y, p.nzs) * nw;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * cb + 4 * kq_e + r;
          const int cg = 32 * half + j;
          const f32x4f et = *reinterpret_cast<const f32x4f*>(Et + 4 * j);
          // (with residuals: their rows are requested before the arithmetic of the pass -- loaded where they are used, each row's wait also
          //  covered the previous row's store, four round trips per pass)
          const int cbase = cg < Cout ? cg * y_plane * 4 : F4_OOB;      // (a channel past Cout: every access of the lane out of range)
          f32x4f rs1[4], rs2[4];
          if (RES) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ro = cbase + pix + i * rowb_y;
              rs1[i] = ld4(r1rs, ro, p.r1s);
              rs2[i] = ld4(r2rs, ro, p.r2s);
            }
          }
          float z[4][6];
#pragma unroll
          for (int nu = 0; nu < 6; ++nu) {
            const float m0 = accr(nu, cb, r), m1 = accr(6 + nu, cb, r), m2 = accr(12 + nu, cb, r), m3 = accr(18 + nu, cb, r), m4 = accr(24 + nu, cb, r),
                        m5 = accr(30 + nu, cb, r);
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            z[0][nu] = m0 + s1 + s2;
            z[1][nu] = fmaf(FB, d2, FA * d1);
            z[2][nu] = fmaf(FB2, s2, FA2 * s1);
            z[3][nu] = fmaf(FB3, d2, fmaf(FA3, d1, m5));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float s1 = z[i][1] + z[i][2], d1 = z[i][1] - z[i][2], s2 = z[i][3] + z[i][4], d2 = z[i][3] - z[i][4];
            f32x4f o4 = {z[i][0] + s1 + s2, fmaf(FB, d2, FA * d1), fmaf(FB2, s2, FA2 * s1), fmaf(FB3, d2, fmaf(FA3, d1, z[i][5]))};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = fmaf(o4[e], et[0], et[1]);
              if (ACT1) v = v * (v > 0.f ? p.g1 : p.s1 * p.g1);
              v += nz[i][e] + et[2];
              o4[e] = v * (v > 0.f ? p.g2 : et[3] * p.g2);
            }
            const int ro = cbase + pix + i * rowb_y;
            if (RES) {
              // (an absent residual is a stride-0 pointer at ONE constant zero: the 16-byte load reads its neighbours too, the factor drops them)
              o4 += rs1[i] * (float)p.r1s;
              o4 += rs2[i] * (float)p.r2s;
            }
            if constexpr ((f4ab & 64) != 0) {      // (tuning: keep the arithmetic alive without the store)
              asm volatile("" :: "v"(o4));
            } else if constexpr (DD == 1) {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4f, o4), yrs, ro, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float oe = o4[e];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oe), yrs, ro + e * DD * 4, 0, 0);
              }
            }
          }
          F4_STAMP(41 + 4 * cb + r);
        }
    }
#ifdef VSP_F4_TRACE
    if (trace_on && lane == 0) {
#pragma unroll
      for (int k = 0; k < 52; ++k) f4_trace_buf[wave * 64 + k] = stv[k];
    }
#endif
  }
}

template <bool RES, bool ACT1, int D>
__global__ __launch_bounds__(F4_THR, 1) void conv_wino4f_kernel(const ConvK p, const F4Plan pl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f4_body<RES, ACT1, D>(p, pl, smem, blockIdx.x, 0);
}

// Up to four dilation groups over ONE shared input (SMART branches, dilation 1 / 2 / 4 / 8) in one launch: the grid is cut into one partition
// of workgroups per group, each running the body of its group's dilation on its group's items (four launches of 128 items each leave half the
// chip idle on 512 -> 4 x 128 at 64^2; together they fill it).
template <bool RES, bool ACT1>
__global__ __launch_bounds__(F4_THR, 1) void conv_wino4f_groups_kernel(const ConvK p, const F4PlanG pg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nwgp = pg.g[0].nwg;
  const int grp = blockIdx.x / nwgp, g = blockIdx.x - grp * nwgp;
  const int d = p.dil[grp];
  if (d == 1) f4_body<RES, ACT1, 0>(p, pg.g[grp], smem, g, grp);
  else if (d == 2) f4_body<RES, ACT1, 2>(p, pg.g[grp], smem, g, grp);
  else if (d == 4) f4_body<RES, ACT1, 4>(p, pg.g[grp], smem, g, grp);
  else f4_body<RES, ACT1, 8>(p, pg.g[grp], smem, g, grp);
}

}  // namespace

// launches the fused F(4x4) form serves: one group or up to four dilation groups over ONE shared input, padding = dilation in {1, 2, 4, 8}, H and W
// multiples of 4 x dilation (whole tiles in every polyphase sub-image), Cin % 8 == 0 up to F4_MAXC (512), W >= 16, style scale only (no affine
// shift), 16-byte aligned operand planes, dense output
bool wino4f_eligible(const ConvK& q) {
  if (q.G < 1 || q.G > 4 || (q.G > 1 && q.x_gs != 0) || q.Cin % 8 != 0 || q.Cin > F4_MAXC || q.W < 16 || q.H < 4) return false;
  for (int g = 0; g < q.G; ++g) {
    const int d = q.dil[g];
    if ((d != 1 && d != 2 && d != 4 && d != 8) || q.pady[g] != d || q.padx[g] != d) return false;
    if (q.H % (4 * d) != 0 || q.W % (4 * d) != 0) return false;   // whole 4 x 4 tiles in every polyphase sub-image
  }
  if (q.y_w != q.OW || q.y_h != q.OH) return false;
  if (q.wshp != nullptr && q.wsh_cs != 0) return false;
  if (reinterpret_cast<uintptr_t>(q.x) & 15) return false;
  if ((reinterpret_cast<uintptr_t>(q.y) & 15) || (q.r1s > 1) || (q.r2s > 1) || (q.nzs > 1)) return false;
  if ((q.r1s && (reinterpret_cast<uintptr_t>(q.r1p) & 15)) || (q.r2s && (reinterpret_cast<uintptr_t>(q.r2p) & 15)) ||
      (q.nzs && (reinterpret_cast<uintptr_t>(q.nzp) & 15)))
    return false;
  if ((int64_t)q.Cin * q.H * q.W * 4 >= 0x7fffff00ll || (int64_t)q.cout_g * q.y_h * q.y_w * 4 >= 0x7fffff00ll) return false;
  if ((int64_t)q.G * ((q.cout_g + 31) / 32) * (q.Cin / 4) * F4_SLAB * 4 >= 0x7fffff00ll) return false;
  return true;
}

size_t wino4f_weight_floats(int cin, int cout) {
  const int64_t nks = (cin + 3) / 4, nco2 = (cout + 31) / 32;
  return (size_t)(nco2 * nks * F4_SLAB);
}

int wino4f_weight_launch(float* U, const float* wp, int cin, int cout, hipStream_t stream) {
  const int nks = (cin + 3) / 4, nco2 = (cout + 31) / 32;
  const int64_t units = (int64_t)nco2 * nks;
  wino4f_weight_kernel<<<(unsigned)((units + 3) / 4), 256, 0, stream>>>(U, wp, cin, cout, nks, units);
  return VSP_OK;
}

// q.w = U4F (wino4f_weight_launch)
static int f4_plan(const ConvK& q, int d, int nwg, F4Plan* pl) {
  pl->nks = q.Cin / 4;
  pl->nco2 = (q.cout_g + 31) / 32;
  pl->nbx = d == 8 ? (q.W + 31) / 32 : (q.W + 63) / 64;
  pl->nbyg = d == 8 ? (q.H + 31) / 32 : (q.H + 15) / 16;
  const int64_t items = (int64_t)q.B * pl->nbyg * pl->nbx * pl->nco2;
  if (items > 0x7fffffff) return vsp::fail(VSP_EINVAL, "conv2d_winograd4f: too many tiles");
  pl->items = (int)items;
  pl->J = (pl->items + nwg - 1) / nwg;
  pl->nwg = nwg;
  return VSP_OK;
}

int wino4f_launch(ConvK q, hipStream_t stream) {
  static const int wgs_env = vsp::tune_env("VSP_WINO4F_WGS") ? atoi(vsp::tune_env("VSP_WINO4F_WGS")) : 0;
  int nwg = wgs_env > 0 ? (wgs_env + 7) / 8 * 8 : vsp::kNumCU;
  const bool res = q.r1s || q.r2s, act1 = !(q.s1 == 1.f && q.g1 == 1.f);
  if (q.G > 1) {   // dilation groups: one partition of workgroups per group (a multiple of 8: the XCD walk of the body)
    F4PlanG pg;
    const int nwgp = nwg / q.G / 8 * 8;
    if (nwgp < 8) return vsp::fail(VSP_EINVAL, "conv2d_winograd4f: too few workgroups for %d groups", q.G);
    for (int g = 0; g < 4; ++g)
      if (int rc = f4_plan(q, q.dil[g < q.G ? g : 0], nwgp, &pg.g[g])) return rc;
    const size_t lds = (size_t)F4G<4>::LDS * sizeof(float);       // (the largest of the four bodies)
    static_assert(F4G<4>::LDS >= F4G<8>::LDS && F4G<4>::LDS >= F4G<2>::LDS && F4G<4>::LDS >= F4G<0>::LDS, "LDS of the group launch");
#define F4_LAUNCH_G(RES_, ACT_)                                                                                                   \
  do {                                                                                                                            \
    static vsp::LdsAttrOnce attr;                                                                                                 \
    if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wino4f_groups_kernel<RES_, ACT_>), (int)lds, "conv2d_winograd4f")) return rc; \
    conv_wino4f_groups_kernel<RES_, ACT_><<<nwgp * q.G, F4_THR, lds, stream>>>(q, pg);                                            \
  } while (0)
    if (res) {
      if (act1) F4_LAUNCH_G(true, true); else F4_LAUNCH_G(true, false);
    } else {
      if (act1) F4_LAUNCH_G(false, true); else F4_LAUNCH_G(false, false);
    }
#undef F4_LAUNCH_G
    return VSP_OK;
  }
  F4Plan pl;
  if (int rc = f4_plan(q, q.dil[0], nwg, &pl)) return rc;
  // loader: dilation 1 = window loads into registers (D = 0) unless VSP_WINO4F_LDS=1 asks for the LDS loader (D = 1); dilation 2 / 4 / 8 = LDS loader
  static const int lds_env = vsp::tune_env("VSP_WINO4F_LDS") ? atoi(vsp::tune_env("VSP_WINO4F_LDS")) : 0;
  const int dsel = q.dil[0] == 1 ? (lds_env ? 1 : 0) : q.dil[0];
#define F4_LAUNCH(RES_, ACT_, D_)                                                                                                 \
  do {                                                                                                                            \
    const size_t lds = (size_t)F4G<D_>::LDS * sizeof(float);                                                                      \
    static vsp::LdsAttrOnce attr;                                                                                                 \
    if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wino4f_kernel<RES_, ACT_, D_>), (int)lds, "conv2d_winograd4f")) return rc; \
    conv_wino4f_kernel<RES_, ACT_, D_><<<nwg, F4_THR, lds, stream>>>(q, pl);                                                      \
  } while (0)
#define F4_LAUNCH_D(D_)                                                                                                           \
  do {                                                                                                                            \
    if (res) {                                                                                                                    \
      if (act1) F4_LAUNCH(true, true, D_); else F4_LAUNCH(true, false, D_);                                                       \
    } else {                                                                                                                      \
      if (act1) F4_LAUNCH(false, true, D_); else F4_LAUNCH(false, false, D_);                                                     \
    }                                                                                                                             \
  } while (0)
  switch (dsel) {
    case 0: F4_LAUNCH_D(0); break;
    case 1: F4_LAUNCH_D(1); break;
#ifndef VSP_F4_NODIL
    case 2: F4_LAUNCH_D(2); break;
    case 4: F4_LAUNCH_D(4); break;
    case 8: F4_LAUNCH_D(8); break;
#endif
    default: return vsp::fail(VSP_ENOTSUP, "conv2d_winograd4f: dilation %d", q.dil[0]);
  }
#undef F4_LAUNCH_D
#undef F4_LAUNCH
#ifdef VSP_F4_TRACE
  {
    static int shots = 0;
    if (++shots == 3) {
      unsigned long long h[4 * 64];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(h, HIP_SYMBOL(f4_trace_buf), sizeof(h));
      for (int w = 0; w < 4; ++w) {
        auto d = [&](int a, int b0) { return (unsigned)(h[w * 64 + a] - h[w * 64 + b0]); };
        printf("f4trace wave %d (cycles): setup %u prologue %u | k-steps", w, d(1, 0), d(2, 1));
        for (int k = 0; k < pl.nks; ++k) printf(" %u", d(3 + k, 2 + k));
        printf(" | to epilogue loads %u | (cb, r) passes", d(40, 2 + pl.nks));
        for (int k = 0; k < 8; ++k) printf(" %u", d(41 + k, 40 + k));
        printf(" | item %u\n", d(48, 0));
      }
      fflush(stdout);
    }
  }
#endif
  return VSP_OK;
}

}  // namespace vspconv

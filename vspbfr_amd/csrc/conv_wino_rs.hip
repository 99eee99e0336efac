// Winograd F(2x2, 3x3) on fp32 MFMA, "register-resident U" form (round 5) for the LOW-CHANNEL layers (Cin <= 64): the 64 -> 4 x 16
// dilation groups of the SMART layers at 512^2 (reference models/RestoreNet.py:179-244, 270-418) and the plain 64 / 32-channel layers.
//
// What the row-owner kernels (conv_wino_ro.hip / conv_wino_rod.hip) wait for on these layers is not the matrix pipe: a 64-channel layer
// is 8 sub-stages long, so the prologue (first patch fetch), the U stream (one fresh fragment set per k-step, ~20 % of the time) and the
// epilogue are a third of every workgroup's life, and a 16-channel dilation group re-uses a B fragment for ONE MFMA.  Here
//   * a workgroup is PERSISTENT and owns ONE block of 16 output channels of one group of one image for a whole chunk of tiles: its U
//     (16 positions x Cin x 16 channels = 64 KB at Cin = 64) lives in REGISTERS -- wave xi holds the four positions (xi, nu = 0..3),
//     Cin / 4 k-steps x 4 = 64 VGPRs, already multiplied by the image's style scale -- so the main loop has no weight loads at all and
//     the patch is committed without arithmetic;
//   * a wave owns a whole ROW of the transformed tile: two window rows x four columns (four 8-byte LDS reads) give all four B fragments
//     of its positions with eight additions -- one read and two VALU per MFMA where the (xi, nu pair) ownership pays two and three;
//   * dilation is removed by POLYPHASE staging in both directions: an item is 8 rows of one row-residue class (rows ry, ry + d, ...)
//     by 32 image columns, committed to LDS de-interleaved into d column-residue strips; inside a strip the windows of a dilated tile
//     are adjacent words, so the main loop is the SAME code for d = 1, 2, 4, 8 (only two lane offsets differ) and never sees d;
//   * the patch ring runs on across items (the loads of the next item's first stage leave two stages ahead), so only a chunk has a
//     prologue; the epilogue (row transform in registers, column transform through a 32 KB LDS exchange, stores) is covered by the
//     second resident workgroup of the CU (two 256-thread workgroups per CU, <= 256 VGPRs each).
// The plain layers are served as "groups" of 16 channels with d = 1: every block stages the patch and computes its V again -- the
// price of keeping U in registers -- which the LDS (25 % busy) and the VALU (two operations per MFMA) have room for.
// U is the fragment order of vsp_winograd_weight_f32 (weight_pack.hip); epilogue chain and operands are those of conv_wino.hip.
#include "conv_kernel.h"

namespace vspconv {

namespace {

constexpr int RS_NTHR = 256;
constexpr int RS_ROWP = 48;                   // LDS row pitch in floats: d strips of SW(d) floats (42, 44, 40, 48 used)
constexpr int RS_PR = 10;                     // staged rows of the residue class: 8 + halo
constexpr int RS_PLANE = RS_PR * RS_ROWP;     // 480 == 32 (mod 64): the second channel of a 32-lane read group takes the other bank half
constexpr int RS_STAGE = 8 * RS_PLANE;        // floats per ring slot (8 channels)
constexpr int RS_RING = 2 * RS_STAGE;
constexpr int RS_XCH = 4 * 2 * 4 * 64 * 4;    // exchange: [nb 4][xi 4][j 2][lane 64][r 4] floats = 32 KB
constexpr int RS_TAB = 64;                    // epilogue operands of the block's 16 channels
constexpr int RS_LDS_FLOATS = RS_RING + RS_XCH + RS_TAB;
constexpr int RS_NLD = 4;                     // 16-byte segments per thread and stage (8 planes x <= 120 segments)
constexpr unsigned RS_OOB = 0x7fffffffu;      // lane offset past every buffer: the load returns zeros (padding, absent channels)

struct RsPlan {
  int nwg;          // workgroups launched (multiple of 8)
  int J;            // chunks per (image, block): chunk j owns items j, j + J, ...
  int nblk, bpg;    // blocks = G * bpg, 16-channel blocks per group
  int nkeys, kx;    // keys = (image, chunk) pairs; keys per XCD
  int cbk;          // column blocks of 32 per image row
  int cbk_shift;    // log2(cbk) when it is a power of two, else -1
  int n_items[4];   // items of a block of group g: d * row blocks * cbk
  int mbw, nct, nch;  // U layout (weight_pack.hip): 16-channel blocks per unit, units per group, k-steps
};

#ifdef VSP_RS_TRACE   // tuning only: shader-clock stamps of one workgroup's waves inside one item (tools/build_abl.sh conv_wino_rs.hip VSP_RS_TRACE rstrace)
__device__ unsigned long long rs_trace_buf[4 * 64];
__device__ int rs_trace_wg = 77;
#define RS_STAMP(idx)                                                                                      \
  do {                                                                                                     \
    if (trace_on && lane == 0) rs_trace_buf[xi * 64 + (idx)] = __builtin_readcyclecounter();              \
  } while (0)
#else
#define RS_STAMP(idx) do {} while (0)
#endif

typedef float f32x2r __attribute__((ext_vector_type(2)));
typedef float f32x4r __attribute__((ext_vector_type(4)));

template <int NST>
__global__ __launch_bounds__(RS_NTHR, 2) void conv_wino_rs_kernel(const ConvK p, const RsPlan pl) {
  constexpr int NKS = 2 * NST;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xl = smem + RS_RING;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int wgid = blockIdx.x;
  const int xcd = wgid & 7, slot0 = wgid >> 3, S = pl.nwg >> 3;
  // wave xi -> W = d[rA] + sgn d[rB] (row combination xi of B^T d)
  const int rA = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
  const int rB = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
  const float sgn = xi == 1 ? 1.f : -1.f;
  const int chw = p.H * p.W;
  const int Cout = p.G * p.cout_g;
  const int y_plane = p.y_h * p.y_w;
  const float nw = p.nwp[0];
#if defined(VSP_RS_ABL)   // tuning only: 1 no MFMAs, 2 no window reads / fragment arithmetic, 4 no commits, 8 no patch loads, 16 no epilogue, 32 no barriers in the stage loop, 64 no stores
  constexpr int ab = VSP_RS_ABL;          // (compile-time switches: tools/build_abl.sh conv_wino_rs.hip VSP_RS_ABL=<bits> <name>; run-time switches cost a branch per MFMA)
#elif defined(VSP_WINO_ABLATE)
  const int ab = p.dbg;
#else
  constexpr int ab = 0;
#endif

  for (int m = slot0; m < pl.kx * pl.nblk; m += S) {
    const int key = xcd * pl.kx + m / pl.nblk, blk = m % pl.nblk;
    if (key >= pl.nkeys) break;
    const int b = key / pl.J, j0 = key - b * pl.J;
    const int g = blk / pl.bpg, cb = blk - g * pl.bpg;
    const int n_it = pl.n_items[g];
    if (j0 >= n_it) continue;
    const int d = p.dil[g];
    const int dl = d == 1 ? 0 : (d == 2 ? 1 : (d == 4 ? 2 : 3));
    const int SW = d == 1 ? 42 : (d == 2 ? 22 : (d == 4 ? 10 : 6));      // strip width: 32 / d + halo, even
    const int OFF = d == 1 ? 5 : (d == 2 ? 3 : 1);                        // strip index of sub-column -1 is OFF - 1 (even: 8-byte window reads)
    const int HL = d == 8 ? 8 : 4;                                        // staged column halo: whole 16-byte segments
    const int SPR = (32 + 2 * HL) >> 2;                                   // segments per staged row
    const int NSEG = RS_PR * SPR;
    const float* xb = p.x + (int64_t)b * p.x_ch * chw;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.Cin * chw * 4, 0x00020000);

    // ---- staging constants of this thread, packed in one register per load: segment idx = tid + 256 i -> row (4 bits) | seg (4) | plane (3) |
    //      LDS word of the segment's first column (12); threads past the stage repeat segment 0
    unsigned s_pk[RS_NLD];
    int s_off[RS_NLD];
#pragma unroll
    for (int i = 0; i < RS_NLD; ++i) {
      int idx = tid + RS_NTHR * i;
      idx = idx < 8 * NSEG ? idx : 0;
      const int pln = idx / NSEG, rem = idx - pln * NSEG;
      const int row = rem / SPR, seg = rem - row * SPR;
      const int co8 = 4 * seg - HL + 8;                                   // first column of the segment relative to the item's X0, + 8 (>= 0)
      const int rx0 = co8 & (d - 1), sc0 = (co8 >> dl) - (8 >> dl) + OFF;
      const int dst = pln * RS_PLANE + row * RS_ROWP + rx0 * SW + sc0;
      s_pk[i] = (unsigned)(row | (seg << 4) | (pln << 8) | (dst << 11));
      s_off[i] = ((row - 1) * d * p.W + 4 * seg - HL + pln * chw) * 4;     // byte offset of the segment from the item's first output pixel
    }
    // the four words of a segment: consecutive (d = 1), alternating strips (d = 2), one strip each (d >= 4)
    const int dlt1 = d == 1 ? 1 : SW, dlt2 = d == 1 ? 2 : (d == 2 ? 1 : 2 * SW), dlt3 = d == 1 ? 3 : (d == 2 ? SW + 1 : 3 * SW);

    // ---- item geometry: item i -> column block i % cbk, t = i / cbk -> row residue t % d, row block t / d
    auto item_xy = [&](int it, int& ry, int& oy0, int& X0) {
      const int t = pl.cbk_shift >= 0 ? it >> pl.cbk_shift : it / pl.cbk;
      const int cbi = it - t * pl.cbk;
      ry = t & (d - 1);
      oy0 = 8 * (t >> dl);
      X0 = 32 * cbi;
    };
    auto slot_voff = [&](int i, int ry, int oy0, int X0) -> unsigned {
      const int row = s_pk[i] & 15, seg = (s_pk[i] >> 4) & 15, pln = (s_pk[i] >> 8) & 7;
      const int sr = oy0 - 1 + row, iy = sr * d + ry, ix = X0 + 4 * seg - HL;
      const bool in = sr >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      return in ? (unsigned)((iy * p.W + ix + pln * chw) * 4) : RS_OOB;
    };
    // An item whose staged rows and columns all lie inside the map (most of them) needs one addition per load: item base + s_off.
    struct ItemGeo { int mode, base, ry, oy0, X0; };   // mode 0: no item (padding marker), 1: interior, 2: border (per-segment validity)
    auto item_geo = [&](int it, bool exists) -> ItemGeo {
      ItemGeo gq;
      item_xy(it, gq.ry, gq.oy0, gq.X0);
      const bool interior = gq.oy0 >= 1 && (gq.oy0 + 8) * d + gq.ry < p.H && gq.X0 - HL >= 0 && gq.X0 + 32 + HL <= p.W;
      gq.mode = exists ? (interior ? 1 : 2) : 0;
      gq.base = ((gq.oy0 * d + gq.ry) * p.W + gq.X0) * 4;
      return gq;
    };
    auto geo_voff = [&](const ItemGeo& gq, int i) -> unsigned {
      if (gq.mode == 1) return (unsigned)(gq.base + s_off[i]);
      if (gq.mode == 0) return RS_OOB;
      return slot_voff(i, gq.ry, gq.oy0, gq.X0);
    };
    f32x4r preg[RS_NLD];
    // (the stage's channel offset rides in the LANE offset, the scalar offset stays 0: the range check is then simply lane offset >= size;
    //  the padding marker stays above every buffer size: 0x7fffffff + offset < 2^32)
    auto load_slot = [&](const ItemGeo& gq, int s, int i) {
      if (ab & 8) return;
      preg[i] = __builtin_bit_cast(f32x4r, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(geo_voff(gq, i) + (unsigned)(s * 8 * chw * 4)), 0, 0));
    };
    auto load_stage = [&](const ItemGeo& gq, int s) {
#pragma unroll
      for (int i = 0; i < RS_NLD; ++i) load_slot(gq, s, i);
    };
    auto commit_slot = [&](float* dst, int i) {
      if (ab & 4) return;
      float* q = dst + (s_pk[i] >> 11);
      q[0] = preg[i][0];
      q[dlt1] = preg[i][1];
      q[dlt2] = preg[i][2];
      q[dlt3] = preg[i][3];
    };
    auto commit_stage = [&](float* dst) {
#pragma unroll
      for (int i = 0; i < RS_NLD; ++i) commit_slot(dst, i);
    };

    // ---- U of this block: position (xi, nu), k-step k, times the image's style scale of the lane's input channel.  Buffer loads: k-steps
    //      past the layer's channels lie outside the resource (zeros).
    float u[4][NKS];
    {
      const int t = cb / pl.mbw, mb = cb - t * pl.mbw;
      const int unit_b = 1024 * pl.mbw * 4;                                // bytes of one (group, unit, k-step) run
      const float* ug = p.w + (int64_t)(g * pl.nct + t) * pl.nch * (1024 * pl.mbw);
      const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ug), 0, pl.nch * unit_b, 0x00020000);
      const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wtp + b * p.wt_bs), 0,
                                                                             ((p.Cin - 1) * p.wt_cs + 1) * 4, 0x00020000);
      const int lo = ((4 * xi * 64 + lane) * pl.mbw + mb) * 4;
#pragma unroll
      for (int k = 0; k < NKS; ++k) {
        const int ci = 4 * k + kq;
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc, (ci < p.Cin ? ci : p.Cin - 1) * p.wt_cs * 4, 0, 0));
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
          u[nu][k] = sc * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ursrc, lo + nu * 64 * pl.mbw * 4 + k * unit_b, 0, 0));
      }
    }
    // ---- epilogue operands of the block's 16 channels in LDS: [a = out_scale * ch_scale | b = ch_bias + bias1 | bias2 | slope2][16]
    float* Tl = Xl + RS_XCH;
    if (tid < 16) {
      const int col = cb * 16 + tid;
      const int cg = g * p.cout_g + (col < p.cout_g ? col : p.cout_g - 1);
      Tl[tid] = p.osp[((int64_t)b * Cout + cg) * p.oss] * p.csp[cg * p.css];
      Tl[16 + tid] = p.cbp[cg * p.cbs] + p.b1p[cg * p.b1s];
      Tl[32 + tid] = p.b2p[cg * p.b2s];
      Tl[48 + tid] = p.s2p[cg * p.s2s];
    }
    // output / residual / noise planes of this image as buffer resources: 32-bit lane offsets, an offset past the resource drops the store
    // (tile positions outside the map, channels past the group) and loads zero
    const int ybytes = Cout * y_plane * 4;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + ((int64_t)b * p.y_ch + p.y_coff) * y_plane, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.r1p + ((int64_t)b * p.res_ch + p.res_coff) * y_plane * p.r1s), 0, p.r1s ? ybytes : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.r2p + ((int64_t)b * p.res_ch + p.res_coff) * y_plane * p.r2s), 0, p.r2s ? ybytes : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t nzrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.nzp + (int64_t)b * p.OH * p.OW * p.nzs), 0,
                                                                           p.nzs ? p.OH * p.OW * 4 : 4, 0x00020000);

    // ---- lane's window origin: tile column lr -> strip rx = lr / (16 / d), tile tx = lr % (16 / d); N-block nb = tile row.  The window
    //      reads are volatile: left alone the compiler merges two 8-byte reads off one base into a ds_read2_b64, which the LDS serves at
    //      half the rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table)
    const int tpr = 16 >> dl;
    const int wrx = lr / tpr, wtx = lr - wrx * tpr;
    const int wbase = kq * RS_PLANE + wrx * SW + 2 * wtx + (OFF - 1);
    typedef const float __attribute__((address_space(3))) * lds_f_t;
    typedef const volatile f32x2r __attribute__((address_space(3))) * lds_win_t;
    const lds_f_t winA = (lds_f_t)smem + wbase + rA * RS_ROWP;
    const lds_f_t winB = (lds_f_t)smem + wbase + rB * RS_ROWP;

    // ---- chunk prologue
    ItemGeo gcur = item_geo(j0, true);
    load_stage(gcur, 0);
    commit_stage(smem);
    load_stage(gcur, 1);
    __syncthreads();

    for (int it = j0; it < n_it; it += pl.J) {
      const bool has_next = it + pl.J < n_it;
      const ItemGeo gnxt = item_geo(has_next ? it + pl.J : it, has_next);
      float nzv[2][2];
#ifdef VSP_RS_TRACE
      const bool trace_on = wgid == rs_trace_wg && it == j0 + 5 * pl.J;
#endif
      f32x4 acc[4][4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nu][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
      for (int s = 0; s < NST; ++s) {
        // A stage is eight (k-step, N-block) units q = 4 ks + nb, run as a three-deep private pipeline: window reads of unit q + 2 |
        // fragment arithmetic of unit q + 1 | the four MFMAs of unit q, the VALU pairs placed between the MFMAs.  The scheduling fences
        // keep that order (left alone the compiler sinks every read group next to its first use and the wave stalls on LDS latency
        // once per unit).  The pipeline drains at the end of a stage: the next stage's slot is only readable behind the barrier.
        const int sl = s & 1;
        f32x2r wa[2][2], wb[2][2];
        // The fragment arithmetic as PACKED fp32 (round 5, tools/ubench/mfma_valu_gap.hip: fp32 MFMAs and fp32 VALU share the SIMD's FMA
        // lanes -- a vector instruction beside a v_mfma_f32_16x16x4_f32 costs its full 5-6 cycles of matrix time, packed or not): the two
        // row combinations of a window quad are one v_pk_fma_f32 each, (V0, V1) = (W0 - W2, W1 + W2) and (V2, V3) = (W2 - W1, W1 - W3) one
        // v_pk_add_f32 each with half selection / negation -- four vector instructions per unit of four MFMAs where there were eight.
        f32x2r vcA = {1.f, 1.f}, vcB = {1.f, 1.f}, vnA = {1.f, 1.f}, vnB = {1.f, 1.f}, w01 = {1.f, 2.f}, w23 = {3.f, 4.f};
        const f32x2r sgn2 = {sgn, sgn};
        if (ab & 2) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) wa[a][c] = wb[a][c] = f32x2r{1.f, 2.f};
        }
        auto read_win = [&](int q) {       // q = 4 ks + nb
          if (ab & 2) return;
          const int buf = q & 1;
          const int off = sl * RS_STAGE + (q >> 2) * 4 * RS_PLANE + (q & 3) * 2 * RS_ROWP;
          wa[buf][0] = *(lds_win_t)(winA + off);
          wa[buf][1] = *(lds_win_t)(winA + off + 2);
          wb[buf][0] = *(lds_win_t)(winB + off);
          wb[buf][1] = *(lds_win_t)(winB + off + 2);
        };
        auto rows01 = [&](int q) { if (ab & 2) return; const int buf = q & 1; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(w01) : "v"(wb[buf][0]), "v"(sgn2), "v"(wa[buf][0])); };
        auto rows23 = [&](int q) { if (ab & 2) return; const int buf = q & 1; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(w23) : "v"(wb[buf][1]), "v"(sgn2), "v"(wa[buf][1])); };
        auto cols01 = [&](f32x2r& v) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v) : "v"(w01), "v"(w23)); };                      // (W0 - W2, W1 + W2)
        auto cols23 = [&](f32x2r& v) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v) : "v"(w23), "v"(w01)); };       // (W2 - W1, W1 - W3)
        RS_STAMP(4 * s + 0);
        read_win(0);
        read_win(1);
        __builtin_amdgcn_sched_barrier(0);
        rows01(0); rows23(0);
        cols01(vcA); cols23(vcB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int ks = q >> 2, nb = q & 3;
          const int k = 2 * s + ks;
          const bool nxt = q < 7;
          const float vc0 = vcA.x, vc1 = vcA.y, vc2 = vcB.x, vc3 = vcB.y;
          // (the four vector instructions in ONE run behind the second MFMA: a lone vector instruction between two MFMAs costs 13 cycles,
          //  one inside a run 5.5 -- tools/ubench/mfma_valu_gap.hip)
          if (!(ab & 1)) acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0][k], vc0, acc[0][nb], 0, 0, 0);
          if (!(ab & 1)) acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1][k], vc1, acc[1][nb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (nxt) { rows01(q + 1); rows23(q + 1); cols01(vnA); cols23(vnB); }
          __builtin_amdgcn_sched_barrier(0);
          if (q + 2 < 8) read_win(q + 2);   // (the buffer of unit q: its rows were combined one unit ago)
          if (!(ab & 1)) acc[2][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2][k], vc2, acc[2][nb], 0, 0, 0);
          if (!(ab & 1)) acc[3][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[3][k], vc3, acc[3][nb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          vcA = vnA; vcB = vnB;
          // Staging rides on the units: units 4..7 commit one load slot each to the other ring slot and re-issue it for the stage after
          // next (the last two stages load the NEXT item's first stages; none: padding marker, zeros) -- a slot's registers are in flight
          // for a whole stage.
          if (q >= 4) {
            if (q == 4) RS_STAMP(4 * s + 1);
            commit_slot(smem + (sl ^ 1) * RS_STAGE, q - 4);
            if (s + 2 < NST) load_slot(gcur, s + 2, q - 4); else load_slot(gnxt, s + 2 - NST, q - 4);
            if (s == NST - 1 && q == 4) {   // the noise of this wave's output rows leaves now: the epilogue finds it in registers
              const int ry = gcur.ry, oy0 = gcur.oy0, X0 = gcur.X0;
              const int ox0 = X0 + 2 * wtx * d + wrx;
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int oy = (oy0 + 2 * xi + i) * d + ry;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                  const int ox = ox0 + jj * d;
                  const bool pin = oy < p.OH && ox < p.OW;
                  nzv[i][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nzrs, pin ? (oy * p.OW + ox) * p.nzs * 4 : 0, 0, 0));
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q == 7) RS_STAMP(4 * s + 2);
          }
        }
        RS_STAMP(4 * s + 3);
        if (s < NST - 1 && !(ab & 32)) __syncthreads();
      }
#if defined(VSP_WINO_ABLATE) || defined(VSP_RS_ABL)
      if (ab & 16) {   // no epilogue (one word keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) sum += acc[nu][nb][0] + acc[nu][nb][1] + acc[nu][nb][2] + acc[nu][nb][3];
        if (sum == 123.456f) p.y[0] = 1.f;
        __syncthreads();
        continue;
      }
#endif

      // ---- epilogue: row transform over nu in registers, column transform over xi through LDS (exchange [nb][xi][j][lane] x 4 channels);
      //      wave w finishes tile row w: its two output rows x 32 columns x 16 channels go back through ITS OWN part of the exchange
      //      (the slots it has just read) as a dense [channel][row][column] image and leave as whole 16-byte quads -- every store
      //      instruction writes complete 128-byte row segments whatever the dilation (the tiles of a dilated group are scattered
      //      d apart: stored from the fragments they are 4-byte words, 16 per line and instruction)
      {
        f32x4r* Xw = reinterpret_cast<f32x4r*>(Xl) + (xi * 2) * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          f32x4r t0, t1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            t0[r] = acc[0][nb][r] + acc[1][nb][r] + acc[2][nb][r];
            t1[r] = acc[1][nb][r] - acc[2][nb][r] - acc[3][nb][r];
          }
          Xw[(nb * 8 + 0) * 64] = t0;
          Xw[(nb * 8 + 1) * 64] = t1;
        }
      }
      RS_STAMP(4 * NST + 0);
      __syncthreads();
      RS_STAMP(4 * NST + 1);
      {
        const int ry = gcur.ry, oy0 = gcur.oy0, X0 = gcur.X0;
        float* Zw = Xl + xi * (8 * 64 * 4);                                        // this wave's 8 slots: 2048 floats
        const f32x4r* Xr = reinterpret_cast<const f32x4r*>(Zw) + lane;            // [x 4][j 2]
        f32x4r tt[4][2];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) tt[x][jj] = Xr[(x * 2 + jj) * 64];
        const f32x4r e_a = *reinterpret_cast<const f32x4r*>(Tl + 4 * kq), e_b = *reinterpret_cast<const f32x4r*>(Tl + 16 + 4 * kq);
        const f32x4r e_b2 = *reinterpret_cast<const f32x4r*>(Tl + 32 + 4 * kq), e_sl = *reinterpret_cast<const f32x4r*>(Tl + 48 + 4 * kq);
        constexpr int ZP = 68;                                                     // floats per channel: 2 rows x 32 columns + 4 (16-byte rows, bank shift)
        const int zc = (2 * wtx) * d + wrx;                                        // column of j = 0 inside the item; j = 1 lies d to the right
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float yv = i == 0 ? tt[0][jj][r] + tt[1][jj][r] + tt[2][jj][r] : tt[1][jj][r] - tt[2][jj][r] - tt[3][jj][r];
              float v = fmaf(yv, e_a[r], e_b[r]);
              v = (v > 0.f ? v : v * p.s1) * p.g1;
              v += nzv[i][jj] * nw + e_b2[r];
              v = (v > 0.f ? v : v * e_sl[r]) * p.g2;
              Zw[(4 * kq + r) * ZP + i * 32 + zc + jj * d] = v;                    // (after the reads above: same wave, LDS operations complete in order)
            }
        const int col0 = cb * 16;
        const int cbase = (g * p.cout_g + col0) * y_plane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int id = lane + 64 * t;
          const int co = id >> 4, i = (id >> 3) & 1, c4 = id & 7;
          const int oy = (oy0 + 2 * xi + i) * d + ry, ox = X0 + 4 * c4;
          const bool pin = oy < p.OH && ox < p.OW && col0 + co < p.cout_g;
          const unsigned ro = pin ? (unsigned)((cbase + co * y_plane + oy * p.y_w + ox) * 4) : RS_OOB;
          f32x4r o4 = *reinterpret_cast<const f32x4r*>(Zw + co * ZP + i * 32 + 4 * c4);
          if (p.r1s) o4 += __builtin_bit_cast(f32x4r, __builtin_amdgcn_raw_buffer_load_b128(r1rs, (int)ro, 0, 0));
          if (p.r2s) o4 += __builtin_bit_cast(f32x4r, __builtin_amdgcn_raw_buffer_load_b128(r2rs, (int)ro, 0, 0));
          if (!(ab & 64)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o4), yrs, (int)ro, 0, 0);
        }
      }
      RS_STAMP(4 * NST + 2);
      gcur = gnxt;
    }
    __syncthreads();   // (the next chunk's prologue rewrites the ring and the operand table)
  }
}

template <int NST>
int launch_rs(const ConvK& q, const RsPlan& pl, hipStream_t stream) {
  static vsp::LdsAttrOnce attr;
  const size_t lds = (size_t)RS_LDS_FLOATS * sizeof(float);
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_wino_rs_kernel<NST>), (int)lds, "conv2d_winograd (register-resident U)")) return rc;
#ifdef VSP_RS_TRACE
  {
    static bool once = false;
    if (!once && vsp::tune_env("VSP_RS_TRACE_WG")) {
      const int wg = atoi(vsp::tune_env("VSP_RS_TRACE_WG"));
      hipMemcpyToSymbol(HIP_SYMBOL(rs_trace_wg), &wg, sizeof(int));
    }
    once = true;
  }
#endif
  conv_wino_rs_kernel<NST><<<dim3((unsigned)pl.nwg), RS_NTHR, lds, stream>>>(q, pl);
#ifdef VSP_RS_TRACE
  {
    static int shots = 0;

    if (++shots == 3) {
      unsigned long long h[4 * 64];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(h, HIP_SYMBOL(rs_trace_buf), sizeof(h));
      for (int w = 0; w < 4; ++w) {
        printf("rs trace wave %d: ", w);
        for (int s = 0; s < NST; ++s)
          printf("| s%d start+%llu q3 %llu commit %llu rest %llu bar ", s, s ? h[w * 64 + 4 * s] - h[w * 64 + 4 * s - 1] : 0ull, h[w * 64 + 4 * s + 1] - h[w * 64 + 4 * s],
                 h[w * 64 + 4 * s + 2] - h[w * 64 + 4 * s + 1], h[w * 64 + 4 * s + 3] - h[w * 64 + 4 * s + 2]);
        printf("| epi: rowxf+write %llu barrier %llu finish %llu | item %llu\n", h[w * 64 + 4 * NST] - h[w * 64 + 4 * NST - 1],
               h[w * 64 + 4 * NST + 1] - h[w * 64 + 4 * NST], h[w * 64 + 4 * NST + 2] - h[w * 64 + 4 * NST + 1], h[w * 64 + 4 * NST + 2] - h[w * 64]);
      }
    }
  }
#endif
  return VSP_OK;
}

}  // namespace

// The register-resident form serves launches of up to 64 input channels whose rows are whole 16-byte segments, without an affine input
// shift (the patch is committed as it is loaded; the style scale rides on U).
bool wino_rs_eligible(const ConvK& q) {
  if (q.Cin > 64 || q.W % 4 != 0 || (reinterpret_cast<uintptr_t>(q.x) & 15) != 0 || ((int64_t)q.H * q.W) % 4 != 0) return false;
  if (q.wc_cs != 0 || q.wsh_cs != 0 || q.wc_bs != 0) return false;
  if ((int64_t)q.Cin * q.H * q.W * 4 >= ((int64_t)1 << 30)) return false;   // (lane offsets carry the channel: the padding marker lies above every offset)
  if ((int64_t)q.G * q.cout_g * q.y_h * q.y_w * 4 >= ((int64_t)1 << 31) || (int64_t)q.OH * q.OW * 4 >= ((int64_t)1 << 31)) return false;
  // the output (and residual) rows leave as 16-byte quads
  if (q.y_w % 4 != 0 || ((int64_t)q.y_h * q.y_w) % 4 != 0 || (reinterpret_cast<uintptr_t>(q.y) & 15) != 0 || q.r1s > 1 || q.r2s > 1 ||
      (q.r1s && (reinterpret_cast<uintptr_t>(q.r1p) & 15)) || (q.r2s && (reinterpret_cast<uintptr_t>(q.r2p) & 15)))
    return false;
  for (int g = 0; g < q.G; ++g)
    if (q.dil[g] != 1 && q.dil[g] != 2 && q.dil[g] != 4 && q.dil[g] != 8) return false;
  return true;
}

// Where it measured faster than the row-owner / direct kernels (tools/bench_wino_rs.py)
bool wino_rs_profitable(const ConvK& q) {
  int dmax = 1;
  for (int g = 0; g < q.G; ++g) dmax = q.dil[g] > dmax ? q.dil[g] : dmax;
  return dmax > 1 && q.cout_g == 16 && q.Cin == 64 && q.H * q.W >= 128 * 128;
}

int wino_rs_launch(ConvK q, hipStream_t stream) {
  RsPlan pl{};
  pl.bpg = (q.cout_g + 15) / 16;
  pl.nblk = q.G * pl.bpg;
  pl.cbk = (q.W + 31) / 32;
  pl.cbk_shift = (pl.cbk & (pl.cbk - 1)) == 0 ? __builtin_ctz(pl.cbk) : -1;
  int nmax = 0;
  for (int g = 0; g < q.G; ++g) {
    const int d = q.dil[g];
    const int sh = (q.H + d - 1) / d;
    pl.n_items[g] = d * ((sh + 7) / 8) * pl.cbk;
    nmax = pl.n_items[g] > nmax ? pl.n_items[g] : nmax;
  }
  static const int wgs_env = vsp::tune_env("VSP_WINO_RS_WGS") ? atoi(vsp::tune_env("VSP_WINO_RS_WGS")) : 0;
  pl.nwg = wgs_env > 0 ? (wgs_env + 7) / 8 * 8 : 2 * vsp::kNumCU;
  // chunks per (image, block): about one chunk per workgroup slot, at least ~2 items per chunk, keys a multiple of 8 when they fill the XCDs
  int J = pl.nwg / (q.B * pl.nblk);
  J = J < 1 ? 1 : J;
  if (J > (nmax + 1) / 2) J = (nmax + 1) / 2;
  J = J < 1 ? 1 : J;
  pl.J = J;
  pl.nkeys = q.B * J;
  pl.kx = (pl.nkeys + 7) / 8;
  const int chunks = pl.nkeys * pl.nblk;
  if (chunks < pl.nwg) pl.nwg = (chunks + 7) / 8 * 8;
  pl.mbw = wino_mbw(q.cout_g);
  pl.nct = (q.cout_g + 16 * pl.mbw - 1) / (16 * pl.mbw);
  pl.nch = (q.Cin + 3) / 4;
  return q.Cin <= 32 ? launch_rs<4>(q, pl, stream) : launch_rs<8>(q, pl, stream);
}

}  // namespace vspconv

// The dilation groups of a SMART branch launch on the bf16 matrix pipe (round 5): 64 -> 4 x 16 at 512^2 with bf16 activations (BASELINE
// configs[2]; reference models/RestoreNet.py:179-244, 270-418) -- the layer the general bf16 kernels were slowest on (conv_bf16.hip:
// 1.4 ms per launch at batch 16, 177 TFLOP/s: every group stages the whole 64-channel patch with an element-wise NCHW -> channel-last
// transpose for its 16 output channels and pads them to a 32-row MFMA block; conv_bf16_rv.hip's row-vector fragments need the taps of a
// pixel adjacent in LDS, i.e. one polyphase image per dilation, and feed one MFMA per fragment at 16 channels).
//
// What bounds a 16-channel group is the work per B fragment, not the matrix pipe (its floor for this layer is 0.15 ms): so
//   * the patch lives in LDS CHANNEL-LAST, [row][pixel][32 channels]: the B fragment of v_mfma_f32_16x16x32_bf16 for one tap is ONE aligned
//     16-byte read (8 channels of one pixel; lane group k reads channels 8 k ...), no shifts, no unpacking, any tap = an address offset;
//   * it is staged POLYPHASE (an item is 8 rows of one row-residue class x 32 image columns of ONE group, as in conv_wino_rs.hip): a
//     loaded 16-byte segment (8 pixels of one channel) is transposed in registers with three more channels (16 v_perm_b32 per 4 x 8
//     block) and every pixel's 4-channel word goes to its residue strip -- de-interleaving costs nothing in this layout, and the taps
//     x - d, x, x + d of a dilated group are neighbouring pixels of a strip: the main loop never sees d;
//   * the group's weights (9 taps x Cin x 16 channels, 18 KB at Cin = 64) sit in LDS as ready A fragments, converted from the fp32 weights
//     and multiplied by the image's style scale once per chunk of tiles (persistent workgroups: no weight traffic and no prologue
//     arithmetic in the loop); a fragment is read once per tap and serves the wave's four N-blocks;
//   * the 16 x 16-pixel accumulator blocks leave through a lane-pair exchange (DPP) as packed bf16 dwords, 32-byte row segments.
// Numerics: bf16(w * s) x bf16(x) products are exact in the fp32 accumulators; the style scale is rounded INTO THE WEIGHT here (the other
// bf16 kernels round x * s): the same number of roundings per product, another place -- tests/test_hip_ops.py::test_conv2d_bf16dg states the
// model and checks against float64 on exactly those operands.
#include "conv_kernel.h"
#include "vsp_bf16.h"

namespace vspconv {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4d __attribute__((ext_vector_type(4)));
typedef unsigned u32x2d __attribute__((ext_vector_type(2)));

constexpr int DG_NTHR = 256;
constexpr int DG_PXB = 80;                      // bytes per staged pixel: 32 channels (64 B) + 16 (16-byte aligned, 20-dword lane stride)
constexpr int DG_ROWPX = 48;                    // pixels per staged row: d strips of 32 / d + 2
constexpr int DG_ROWB = DG_ROWPX * DG_PXB;      // 3840
constexpr int DG_PR = 10;                       // staged rows of the residue class: 8 + halo
constexpr int DG_SLOT = DG_PR * DG_ROWB;        // 38400 bytes: ONE slot (loads are prefetched in registers, two barriers per stage)
constexpr int DG_JUNK = DG_SLOT;                // 80 bytes behind the slot: words of pixels outside the strips land here
constexpr int DG_TAB = DG_SLOT + 128;           // epilogue operands of the block's 16 channels: 4 x 16 floats
constexpr int DG_AFR = DG_TAB + 256;            // A fragments: [stage 2][tap 9][lane 64] x 16 bytes = 18432
constexpr int DG_Z = DG_AFR + 2 * 9 * 64 * 16;  // epilogue exchange: per wave [16 channels][4 rows x 16 pixels + 4] floats
constexpr int DG_ZC = 4 * 16 + 4;               // floats per channel (16-byte rows, bank shift)
constexpr int DG_ZW = 16 * DG_ZC;               // floats per wave
constexpr int DG_LDS = DG_Z + 4 * DG_ZW * 4;
constexpr unsigned DG_OOB = 0x7fffffffu;

#ifdef VSP_DG_TRACE   // tuning only: shader-clock stamps of one workgroup's waves inside one item
__device__ unsigned long long dg_trace_buf[4 * 32];
#define DG_STAMP(idx) do { if (trace_on && lane == 0) dg_trace_buf[wave * 32 + (idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define DG_STAMP(idx) do {} while (0)
#endif

struct DgPlan {
  int nwg, J, nblk, nkeys, kx, cbk, cbk_shift;
  int n_items[4];
};

// NST = stages of 32 input channels; FULL = the whole epilogue chain (first activation, noise, second activation) instead of scale + bias;
// RES = residual tensors.  Compile-time: the tile loop of the common form (the SMART branch launch: demodulation only) is straight-line code --
// at a control-flow join hipcc's wait-count pass gives up counting and waits for vmcnt(0), i.e. for the prefetch just issued.
template <int NST, bool FULL, bool RES>
__global__ __launch_bounds__(DG_NTHR, 2) void conv_bf16_dg_kernel(const ConvK p, const DgPlan pl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Tl = reinterpret_cast<float*>(smem + DG_TAB);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kb = lane >> 4;
  const int wgid = blockIdx.x;
  const int xcd = wgid & 7, slot0 = wgid >> 3, S = pl.nwg >> 3;
  const int chw = p.H * p.W;
  const int Cout = p.G * p.cout_g;
  const int y_plane = p.y_h * p.y_w;
  const float nw = p.nwp[0];
  const unsigned short* xg = reinterpret_cast<const unsigned short*>(p.x);
  unsigned short* yg = reinterpret_cast<unsigned short*>(p.y);

  for (int m = slot0; m < pl.kx * pl.nblk; m += S) {
    const int key = xcd * pl.kx + m / pl.nblk, g = m % pl.nblk;
    if (key >= pl.nkeys) break;
    const int b = key / pl.J, j0 = key - b * pl.J;
    const int n_it = pl.n_items[g];
    if (j0 >= n_it) continue;
    const int d = p.dil[g];
    const int dl = d == 1 ? 0 : (d == 2 ? 1 : (d == 4 ? 2 : 3));
    const int SW = (32 >> dl) + 2;                                       // pixels per residue strip: 32 / d + halo
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xg + (int64_t)b * p.x_ch * chw), 0, p.Cin * chw * 2, 0x00020000);

    // ---- staging units of this thread: u = tid, tid + 256 -> (channel quad q, segment, row): four channels x eight pixels each
    //      (10 rows x 6 segments of 8 pixels (columns X0 - 8 ... X0 + 39) x 8 channel quads = 480 units per stage)
    //      packed per unit: q (3 bits) | segment (3) | row (4; 15 = no unit) | LDS byte of the segment's first pixel + 1024 (18)
    unsigned u_pk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + DG_NTHR * i;
      const bool live = u < 480;
      const int uu = live ? u : 0;
      const int q = uu & 7, seg = (uu >> 3) % 6, row = (uu >> 3) / 6;
      // pixel j of the segment: column c = 8 seg - 8 + j relative to X0 -> strip c mod d = j mod d, entry (c + d) / d = e0 + (j >> dl)
      const int e0 = (8 * seg - 8 + d) >> dl;
      const int dst = (row * DG_ROWPX + e0) * DG_PXB + q * 8 + 1024;
      u_pk[i] = (unsigned)(q | (seg << 3) | ((live ? row : 15) << 6) | (dst << 10));
    }
    auto item_xy = [&](int it, int& ry, int& oy0, int& X0) {
      const int t = pl.cbk_shift >= 0 ? it >> pl.cbk_shift : it / pl.cbk;
      const int cbi = it - t * pl.cbk;
      ry = t & (d - 1);
      oy0 = 8 * (t >> dl);
      X0 = 32 * cbi;
    };
    // one register set per stage: the loads of (next item, stage s) leave right behind the commit of (this item, stage s) -- a whole item
    // (NST commits and compute phases) between a load and its use
    u32x4d pregs[NST][2][4];
    auto load_stage = [&](int ry, int oy0, int X0, int s, bool exists) {
      u32x4d (&preg)[2][4] = pregs[s];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int urow = (u_pk[i] >> 6) & 15, useg = (u_pk[i] >> 3) & 7, uq = u_pk[i] & 7;
        const int sr = oy0 - 1 + urow, iy = sr * d + ry, ix = X0 - 8 + 8 * useg;
        // (bitwise: one mask, no short-circuit control flow around the loads -- a branchy form made hipcc load twice into the same registers
        //  with vmcnt(0) between)
        const int in = (int)exists & (int)(urow < 15) & (int)(sr >= 0) & (int)(iy < p.H) & (int)(ix >= 0) & (int)(ix < p.W);
        const int c0 = 32 * s + 4 * uq;
        const int pix = (iy * p.W + ix) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ok = in & (int)(c0 + k < p.Cin);
          const unsigned off = ok ? (unsigned)((c0 + k) * chw * 2 + pix) : DG_OOB;
          preg[i][k] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off, 0, 0);
        }
      }
    };
    // one unit: 4 channels x 8 pixels (16 dwords, pixel pairs per dword) -> 8 pixels x 4 channels (two dwords each), scattered to the strips
    auto commit_stage = [&](int s) {
      u32x4d (&preg)[2][4] = pregs[s];
      constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;             // v_perm_b32: low / high halves of (S0, S1) = (second, first) argument
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int dead = (int)(((u_pk[i] >> 6) & 15) == 15);
        const int e0 = (8 * (int)((u_pk[i] >> 3) & 7) - 8 + d) >> dl;
        const int udst = (int)(u_pk[i] >> 10) - 1024;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {                                  // pixel pair (2 jp, 2 jp + 1) = dword jp of every channel's segment
          const unsigned a0 = preg[i][0][jp], a1 = preg[i][1][jp], a2 = preg[i][2][jp], a3 = preg[i][3][jp];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * jp + h;
            const u32x2d w = {__builtin_amdgcn_perm(a1, a0, h ? HI : LO), __builtin_amdgcn_perm(a3, a2, h ? HI : LO)};
            const int ent = e0 + (j >> dl);                               // entry of the pixel in its strip; outside [0, SW): not staged
            const int off = udst + ((j & (d - 1)) * SW + (j >> dl)) * DG_PXB;
            const int ok = (int)(ent >= 0) & (int)(ent < SW) & (dead ^ 1);
            const int adr = ok ? off : DG_JUNK + (lane & 7) * 8;            // (a select, not a branch around the store)
            *reinterpret_cast<u32x2d*>(smem + adr) = w;
          }
        }
      }
    };

    // ---- A fragments in LDS: [stage][tap][lane 64] x 16 bytes, lane (kb, n16) = row co = n16, k = channels 32 s + 8 kb + 0..7 of tap t.  `w`
    //      arrives in FRAGMENT ORDER (hip_ops.bf16dg_weight: fp32 [group][stage][tap][lane][8], zero-padded); the workgroup multiplies it by
    //      the image's style scale of the eight channels and rounds to bf16 once per chunk.  (Round 5, first version: the 18 fragments in
    //      REGISTERS of every wave, 72 VGPRs -- with the two prefetch register sets of the staging the kernel then spilled inside the tile
    //      loop, and a scratch reload waits on vmcnt(0): every patch load was serialised behind the previous one, 4-6k cycles per stage.)
    __syncthreads();   // (the previous chunk's readers of the table, the fragments and the slot are done)
    {
      const __amdgpu_buffer_rsrc_t wrs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w + (int64_t)g * NST * 9 * 512), 0, NST * 9 * 512 * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_scale + (int64_t)b * p.in_scale_bstride), 0,
                                                                           p.bf_isc_s ? p.Cin * 4 : 4, 0x00020000);
      typedef float f32x4w __attribute__((ext_vector_type(4)));
      for (int f = tid; f < NST * 9 * 64; f += DG_NTHR) {                  // fragment slot (stage, tap, lane)
        const int fl = f & 63, st = f >> 6, fs = st / 9;
        u32x4d pk;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4w wv = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(wrs, (f * 8 + 4 * h) * 4, 0, 0));
          f32x4w sv = {1.f, 1.f, 1.f, 1.f};
          if (p.bf_isc_s) sv = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(srs, (32 * fs + 8 * (fl >> 4) + 4 * h) * 4, 0, 0));   // (past Cin: zeros)
          pk[2 * h] = vsp::bf16_pack(wv[0] * sv[0], wv[1] * sv[1]);
          pk[2 * h + 1] = vsp::bf16_pack(wv[2] * sv[2], wv[3] * sv[3]);
        }
        *reinterpret_cast<u32x4d*>(smem + DG_AFR + f * 16) = pk;
      }
    }
    // ---- epilogue operands of the group's 16 channels: [a = out_scale * ch_scale | b = ch_bias + bias1 | bias2 | slope2][16]
    if (tid < 16) {
      const int cg = g * p.cout_g + (tid < p.cout_g ? tid : p.cout_g - 1);
      Tl[tid] = p.osp[((int64_t)b * Cout + cg) * p.oss] * p.csp[cg * p.css];
      Tl[16 + tid] = p.cbp[cg * p.cbs] + p.b1p[cg * p.b1s];
      Tl[32 + tid] = p.b2p[cg * p.b2s];
      Tl[48 + tid] = p.s2p[cg * p.s2s];
    }
    const int64_t y_img = ((int64_t)b * p.y_ch + p.y_coff + g * p.cout_g) * y_plane;           // elements
    const int64_t r_img = ((int64_t)b * p.res_ch + p.res_coff + g * p.cout_g) * y_plane;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yg + y_img, 0, p.cout_g * y_plane * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(p.r1p) + r_img * p.r1s), 0, p.r1s ? p.cout_g * y_plane * 2 : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(p.r2p) + r_img * p.r2s), 0, p.r2s ? p.cout_g * y_plane * 2 : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t nzrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.nzp + (int64_t)b * p.OH * p.OW * p.nzs), 0,
                                                                           p.nzs ? p.OH * p.OW * 4 : 4, 0x00020000);

    // ---- lane's windows: N-block (row r, half h) = pixels 16 h + n16 of sub-row r; window origin (tap (0, 0)) = row r, entry sc of strip rx
    //      A wave owns FOUR rows of one half (wave w: half w & 1, rows 4 (w >> 1) ...): its 4 x 9 MFMAs per stage read 6 input rows x 3
    //      horizontal taps = 18 B fragments (two rows x two halves would read 24, one read per MFMA 36).
    const int whalf = wave & 1, wrow0 = 4 * (wave >> 1);
    const int wc = 16 * whalf + n16;
    const int wbase = ((wc & (d - 1)) * SW + (wc >> dl)) * DG_PXB + kb * 16 + wrow0 * DG_ROWB;

    // the two output stores of a tile are issued behind the first commit of the NEXT tile: that commit waits for its patch loads, and a wait
    // on the in-order memory counter would otherwise also wait for the stores just issued to drain
    u32x4d st_val[2] = {u32x4d{0, 0, 0, 0}, u32x4d{0, 0, 0, 0}};
    unsigned st_off[2] = {DG_OOB, DG_OOB};
    int ry, oy0, X0;
    item_xy(j0, ry, oy0, X0);
#pragma unroll
    for (int s = 0; s < NST; ++s) load_stage(ry, oy0, X0, s, true);

    for (int it = j0; it < n_it; it += pl.J) {
      const bool has_next = it + pl.J < n_it;
#ifdef VSP_DG_TRACE
      const bool trace_on = wgid == 77 && it == j0 + 9 * pl.J;
#endif
      int n_ry = 0, n_oy0 = 0, n_X0 = 0;
      item_xy(has_next ? it + pl.J : it, n_ry, n_oy0, n_X0);
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float nzv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        DG_STAMP(5 * s + 0);
        __syncthreads();                                                   // every wave is done reading the slot
        DG_STAMP(5 * s + 1);
        commit_stage(s);
        if (s == 0) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) __builtin_amdgcn_raw_buffer_store_b128(st_val[hh], yrs, (int)st_off[hh], 0, 0);
        }
        DG_STAMP(5 * s + 2);
        load_stage(n_ry, n_oy0, n_X0, s, has_next);
        if (FULL && s == NST - 1) {   // the noise of this lane's four output pixels leaves with the last stage's loads
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = wrow0 + i, c = X0 + wc;
            const int oy = (oy0 + r) * d + ry;
            const bool pin = oy < p.OH && c < p.OW;
            nzv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nzrs, pin ? (oy * p.OW + c) * p.nzs * 4 : 0, 0, 0));
          }
        }
        DG_STAMP(5 * s + 3);
        __syncthreads();                                                   // the stage is in LDS
        DG_STAMP(5 * s + 4);
        // the stage's nine A fragments into registers, then input-row-major: the three fragments of input row ir + 1 are read under the
        // MFMAs of row ir (tap row ty = ir - output row)
        {
          bf16x8 af[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) af[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4d*>(smem + DG_AFR + ((s * 9 + t) * 64 + lane) * 16));
          u32x4d rb[2][3];
          auto read_row = [&](int ir, int buf) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) rb[buf][tx] = *reinterpret_cast<const u32x4d*>(smem + wbase + ir * DG_ROWB + tx * DG_PXB);
          };
          read_row(0, 0);
#pragma unroll
          for (int ir = 0; ir < 6; ++ir) {
            if (ir + 1 < 6) read_row(ir + 1, (ir + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
              const int i = ir - ty;
              if (i < 0 || i > 3) continue;
#pragma unroll
              for (int tx = 0; tx < 3; ++tx)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ty * 3 + tx], __builtin_bit_cast(bf16x8, rb[ir & 1][tx]), acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      DG_STAMP(5 * NST + 0);
      // ---- epilogue: lane (kb, n16) holds channels 4 kb + q of pixel n16 of its four rows.  Stored from there the 64 values of a lane are
      //      sixteen 4-byte store instructions (pixel pairs): the store ISSUE, not the bytes, was a quarter of the tile (traced: 3.3k of 11.5k
      //      cycles).  They go through a 4 KB fp32 image of the wave in LDS, [channel][row][16 pixels]; lane L then owns (channel L / 4,
      //      row L % 4): sixteen pixels = ONE 32-byte row segment, two 16-byte stores per lane and tile.
      {
        float* Zw = reinterpret_cast<float*>(smem + DG_Z) + wave * DG_ZW;
        const f32x4 e_a = *reinterpret_cast<const f32x4*>(Tl + 4 * kb), e_b = *reinterpret_cast<const f32x4*>(Tl + 16 + 4 * kb);
        const f32x4 e_b2 = *reinterpret_cast<const f32x4*>(Tl + 32 + 4 * kb), e_sl = *reinterpret_cast<const f32x4*>(Tl + 48 + 4 * kb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = fmaf(acc[i][q], e_a[q], e_b[q]);
            if (FULL) {
              v = (v > 0.f ? v : v * p.s1) * p.g1;
              v += nzv[i] * nw + e_b2[q];
              v = (v > 0.f ? v : v * e_sl[q]) * p.g2;
            } else {
              v += e_b2[q];
            }
            Zw[(4 * kb + q) * DG_ZC + i * 16 + n16] = v;
          }
        const int zco = lane >> 2, zi = lane & 3;
        const int oy = (oy0 + wrow0 + zi) * d + ry, c = X0 + 16 * whalf;
        const bool pin = oy < p.OH && c < p.OW && zco < p.cout_g;
        const unsigned ro = pin ? (unsigned)((zco * y_plane + oy * p.y_w + c) * 2) : DG_OOB;
        typedef float f32x4z __attribute__((ext_vector_type(4)));
        const f32x4z* zr = reinterpret_cast<const f32x4z*>(Zw + zco * DG_ZC + zi * 16);   // (same wave wrote it: LDS operations complete in order)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4z v0 = zr[2 * hh], v1 = zr[2 * hh + 1];
          if (RES) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const __amdgpu_buffer_rsrc_t& rrs = rr ? r2rs : r1rs;
              if (rr ? p.r2s : p.r1s) {
                const u32x4d w = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)((pin && c + 8 * hh < p.OW) ? ro + 16 * hh : DG_OOB), 0, 0);
                v0[0] += vsp::bf16_lo(w[0]); v0[1] += vsp::bf16_hi(w[0]); v0[2] += vsp::bf16_lo(w[1]); v0[3] += vsp::bf16_hi(w[1]);
                v1[0] += vsp::bf16_lo(w[2]); v1[1] += vsp::bf16_hi(w[2]); v1[2] += vsp::bf16_lo(w[3]); v1[3] += vsp::bf16_hi(w[3]);
              }
            }
          }
          st_val[hh] = u32x4d{vsp::bf16_pack(v0[0], v0[1]), vsp::bf16_pack(v0[2], v0[3]), vsp::bf16_pack(v1[0], v1[1]), vsp::bf16_pack(v1[2], v1[3])};
          st_off[hh] = (pin && c + 8 * hh < p.OW) ? ro + 16 * hh : DG_OOB;          // (W % 8 == 0: an 8-pixel group lies inside the row or outside)
        }
      }
      DG_STAMP(5 * NST + 1);
      ry = n_ry; oy0 = n_oy0; X0 = n_X0;
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) __builtin_amdgcn_raw_buffer_store_b128(st_val[hh], yrs, (int)st_off[hh], 0, 0);   // the chunk's last tile
  }
}

template <int NST, bool FULL, bool RES>
int launch_dg(const ConvK& q, const DgPlan& pl, hipStream_t stream) {
  static vsp::LdsAttrOnce attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(conv_bf16_dg_kernel<NST, FULL, RES>), DG_LDS, "conv2d_bf16dg")) return rc;
  conv_bf16_dg_kernel<NST, FULL, RES><<<dim3((unsigned)pl.nwg), DG_NTHR, DG_LDS, stream>>>(q, pl);
#ifdef VSP_DG_TRACE
  {
    static int shots = 0;
    if (++shots == 3) {
      unsigned long long h[4 * 32];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(h, HIP_SYMBOL(dg_trace_buf), sizeof(h));
      for (int w = 0; w < 4; ++w) {
        printf("dg trace wave %d:", w);
        for (int s = 0; s < NST; ++s)
          printf(" | s%d gap+%llu bar1 %llu commit %llu loads %llu bar2 %llu compute", s, s ? h[w * 32 + 5 * s] - h[w * 32 + 5 * s - 1] : 0ull, h[w * 32 + 5 * s + 1] - h[w * 32 + 5 * s],
                 h[w * 32 + 5 * s + 2] - h[w * 32 + 5 * s + 1], h[w * 32 + 5 * s + 3] - h[w * 32 + 5 * s + 2], h[w * 32 + 5 * s + 4] - h[w * 32 + 5 * s + 3]);
        printf(" %llu | epilogue %llu | item %llu\n", h[w * 32 + 5 * NST] - h[w * 32 + 5 * NST - 1], h[w * 32 + 5 * NST + 1] - h[w * 32 + 5 * NST], h[w * 32 + 5 * NST + 1] - h[w * 32]);
      }
    }
  }
#endif
  return VSP_OK;
}

}  // namespace

// What the kernel serves: bf16 activations, up to four dilation groups (1, 2, 4, 8; padding = dilation) of at most 16 channels over one
// shared input of at most 64 channels (a multiple of 8), rows of whole 16-byte segments, dense same-size output.
bool bf16dg_eligible(const ConvK& q) {
  if (!q.io_bf16 || q.KH != 3 || q.KW != 3 || q.G < 1 || q.G > 4 || q.x_gs != 0 || q.cout_g > 16 || q.Cin > 64 || q.Cin % 8) return false;
  for (int g = 0; g < q.G; ++g)
    if (q.dil[g] != 1 && q.dil[g] != 2 && q.dil[g] != 4 && q.dil[g] != 8) return false;
  if (q.bf_ish_s != 0 || q.W % 8 || q.OH != q.H || q.OW != q.W || (q.y_w & 1) || ((int64_t)q.y_h * q.y_w) % 2) return false;
  if ((reinterpret_cast<uintptr_t>(q.x) & 15) || (reinterpret_cast<uintptr_t>(q.y) & 3)) return false;
  if ((q.r1s && (reinterpret_cast<uintptr_t>(q.r1p) & 3)) || (q.r2s && (reinterpret_cast<uintptr_t>(q.r2p) & 3)) || q.r1s > 1 || q.r2s > 1 || q.nzs > 1) return false;
  const int64_t lim = ((int64_t)1 << 30);
  if ((int64_t)q.Cin * q.H * q.W * 2 >= lim || (int64_t)q.cout_g * q.y_h * q.y_w * 2 >= lim || (int64_t)q.OH * q.OW * 4 >= lim) return false;
  return true;
}

int bf16dg_launch(const ConvK& q, bool full, hipStream_t stream) {   // full: a first / second activation or a noise term (else scale + bias only)
  DgPlan pl{};
  pl.nblk = q.G;
  pl.cbk = (q.W + 31) / 32;
  pl.cbk_shift = (pl.cbk & (pl.cbk - 1)) == 0 ? __builtin_ctz(pl.cbk) : -1;
  int nmax = 0;
  for (int g = 0; g < q.G; ++g) {
    const int d = q.dil[g];
    const int sh = (q.H + d - 1) / d;
    pl.n_items[g] = d * ((sh + 7) / 8) * pl.cbk;
    nmax = pl.n_items[g] > nmax ? pl.n_items[g] : nmax;
  }
  static const int wgs_env = vsp::tune_env("VSP_BF16DG_WGS") ? atoi(vsp::tune_env("VSP_BF16DG_WGS")) : 0;
  pl.nwg = wgs_env > 0 ? (wgs_env + 7) / 8 * 8 : 2 * vsp::kNumCU;
  int J = pl.nwg / (q.B * pl.nblk);
  J = J < 1 ? 1 : J;
  if (J > (nmax + 1) / 2) J = (nmax + 1) / 2;
  J = J < 1 ? 1 : J;
  pl.J = J;
  pl.nkeys = q.B * J;
  pl.kx = (pl.nkeys + 7) / 8;
  const int chunks = pl.nkeys * pl.nblk;
  if (chunks < pl.nwg) pl.nwg = (chunks + 7) / 8 * 8;
  // the SMART branch launch itself (demodulation + bias only, no residuals) runs the lean straight-line form
  const bool res = q.r1s != 0 || q.r2s != 0;
  if (q.Cin <= 32) return (full || res) ? launch_dg<1, true, true>(q, pl, stream) : launch_dg<1, false, false>(q, pl, stream);
  return (full || res) ? launch_dg<2, true, true>(q, pl, stream) : launch_dg<2, false, false>(q, pl, stream);
}

}  // namespace vspconv

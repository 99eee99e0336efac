// The Code_diffuser sampler chain as ONE persistent launch (round 3; VERDICT r2 "missing" item 1).
//
// Reference: ldm/ddpm.py:400-429 (p_sample_loop) around models/CodeDiffuser.py:86-140 (four TACC blocks per denoiser call).
// tacc_chain.hip enqueues three launches per block (600 per batch at T = 50); here a CLUSTER of G workgroups (template argument:
// 16 = the latency form described below; 8 / 4 / 2 / 1: each workgroup takes 16 / G slices, see the note at the kernel) owns one image for
// the whole chain -- all T x n_blocks block evaluations and the sampler updates -- and the only things that cross workgroups are
// the three all-to-all tensors of a block, exchanged through L2-bypassing (sc0 sc1) stores and loads behind a cluster barrier:
//
//   proj   workgroup g computes P[:, 128 g .. 128 g + 128) = pixelnorm(y) @ Wcat[cols]^T from ITS OWN copy of y in LDS
//          (8 waves split K, MFMA, rows 16 / 17 as plain FMAs: the arithmetic of tacc_proj_kernel, two 64-column passes)
//   ------ cluster barrier (P complete)
//   attn   workgroup g: channel attention for columns [32 g, 32 g + 32) on MFMA (the arithmetic of chan_attn_mfma_body), then
//          token attention for rows g (and 16 + g for g < 2), one wave per row
//   ------ cluster barrier (t, h complete)
//   post   EVERY workgroup evaluates LN(t), LN(h + LN(t)), FiLM for all 18 rows (36 KB in, redundantly: it removes the third
//          barrier of a block -- the next proj needs all of y) into its LDS copy of y; after the last block of a step the sampler
//          update x' = c1 f + c2 x, x held in two step-parity buffers (the leader workgroup writes X(s+1) while the others may
//          still read X(s))
//
// Two barriers per block instead of three launch boundaries.  Workgroup -> (image, slice) keeps the XCD-aware weight slices of
// tacc_proj_kernel (XCD x only ever touches rows [256 x, 256 x + 256) of a block's 2048 x 512 weight): speed only.  Every
// cross-workgroup word is written with sc0 sc1 stores and read with sc0 sc1 loads (MI355X_MICROARCH.md, valid forms), the
// barrier is one relaxed agent-scope counter per image behind a per-wave vmcnt(0) drain; spins are bounded (a timeout word stops
// every later wait, the caller's parity tests see the garbage).  The grid must be co-resident: G B workgroups of 512 threads
// with 145 KB of LDS, one per CU -> G B <= 256 (the entry refuses larger batches; callers fall back to vsp_tacc_chain_f32).
#include "tacc_kernels.h"
#include <cstdlib>

namespace {

using vsptacc::D;
using vsptacc::NTOK;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// workgroups per image: template argument G of the kernel.  16 = latency form (one 128-column slice of the projection, one 32-column
// block of the channel attention and one or two token rows per workgroup); 4 / 2 / 1 = THROUGHPUT forms for the two-stream batch loop,
// where the chain hides under the previous batch's convolutions and what it costs is the CU-time it holds (round 3: the launched
// chain takes 4.2 ms of a 46.6 ms step although its latency is hidden -- 600 launches x 256 workgroups x ~9 us of mostly waiting):
// fewer, fatter workgroups do the same arithmetic in the same order (bit-identical results) on B * G CUs.
constexpr int NTH = 512;     // threads per workgroup (8 waves)
constexpr int COH = 17;      // cache policy of the exchanged tensors: sc0 | sc1 (write-through stores, L1 / L2-bypassing loads)
constexpr int MAXS = 64;     // steps per launch
constexpr unsigned SPIN_LIMIT = 4000000u;

struct PK {
  int B, n_blocks, n_steps;
  vsp_tacc_block blk[4];
  float* xio;        // (B, 18, 512): chain input and result
  float* xb;         // 2 x (B, 18, 512): step-parity copies of x
  float* P;          // (B 18, 2048)
  float* tb;         // (B 18, 512)
  float* hb;         // (B 18, 512)
  unsigned* sync;    // [B] arrival counters, [32] timeout word
  const float* c1;
  const float* c2;
  float t_div;
  int64_t head_stride;   // floats per step of gamma / beta (= B 18 512)
  short step[MAXS], cidx[MAXS];
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  const int iv = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}

// coherent accesses of the exchanged tensors (byte offsets inside the buffer)
__device__ __forceinline__ float4 cld4(__amdgpu_buffer_rsrc_t r, int off) {
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, COH));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float cld1(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, COH));
}
__device__ __forceinline__ void cst4(__amdgpu_buffer_rsrc_t r, int off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), r, off, 0, COH);
}
__device__ __forceinline__ void cst1(__amdgpu_buffer_rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, COH);
}

// cluster barrier: arrival k of this image (k = 1, 2, ...).  Every wave drains its write-through stores, one lane arrives and polls.
template <int G>
__device__ __forceinline__ void cluster_barrier(unsigned* cnt, unsigned* tmo, unsigned k) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = k * G;
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0) {
        if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > SPIN_LIMIT) {
          __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __syncthreads();
}

constexpr int PJ_NS = D / 16 / 8;   // 16-wide k-steps per wave
constexpr int PJ_NJ = 4;            // 16-column blocks per pass
constexpr int PJ_RED = 8 * PJ_NJ * 4 * 64;
constexpr int PJ_SCR = 2 * (PJ_RED + 8 * 4 * 2 * PJ_NJ * 16);   // two buffer pairs
#ifndef VSP_TP_YPAD
#define VSP_TP_YPAD 4
#endif
constexpr int YP = D + VSP_TP_YPAD;   // row pitch of the workgroup's copy of y: the 16 rows of a fragment read land in 16 different 16-byte slots
constexpr int SCR_FLOATS = (int)vsptacc::CA_LDS_FLOATS > PJ_SCR ? (int)vsptacc::CA_LDS_FLOATS : PJ_SCR;
constexpr size_t LDS_BYTES = (size_t)(NTOK * YP + SCR_FLOATS) * sizeof(float);

template <int G>
__global__ __launch_bounds__(NTH, 2) void tacc_persist_kernel(const PK p) {
  static_assert(G == 16 || G == 8 || G == 4 || G == 2 || G == 1, "cluster size");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ylds = smem;             // [18][YP]: this workgroup's copy of the block input (normalised IN PLACE by the projection)
  float* scr = smem + NTOK * YP;  // phase scratch
  const int tid0 = threadIdx.x, tid = tid0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wid = blockIdx.x;
  // G = 16: XCD x (= wid % 8, observed) owns slices 2x, 2x + 1 of every image; smaller clusters: slice g of an image on XCD wid % 8
  const int b = wid / G, g = G == 16 ? 2 * (wid & 7) + ((wid >> 3) & 1) : wid % G;
  const int M = p.B * NTOK;
  const __amdgpu_buffer_rsrc_t Prs = __builtin_amdgcn_make_buffer_rsrc(p.P, 0, M * 4 * D * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(p.tb, 0, M * D * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(p.hb, 0, M * D * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(p.xb, 0, 2 * M * D * 4, 0x00020000);
  unsigned* cnt = p.sync + b;
  unsigned* tmo = p.sync + 32;
#ifndef VSP_TP_ABL   // tuning builds: 1 no projection MFMAs / FMAs, 2 no projection weight loads, 4 no projection reduction + stores
#define VSP_TP_ABL 0
#endif
#ifdef VSP_TP_TIMING   // tuning builds: cycles of workgroup 0 per phase (proj, barrier 1, attention, barrier 2, post) in sync[40..44]
  unsigned long long tph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define TP_MARK(i) { const unsigned long long tn = __builtin_readcyclecounter(); tph[i] += tn - tlast; tlast = tn; }
#else
#define TP_MARK(i)
#endif
  unsigned bar = 0;
  const int rowb = b * NTOK;   // first token row of this image

  for (int i = tid; i < NTOK * D / 4; i += NTH)
    *reinterpret_cast<float4*>(ylds + (i >> 7) * YP + (i & 127) * 4) = reinterpret_cast<const float4*>(p.xio + (int64_t)rowb * D)[i];
  __syncthreads();

  for (int s = 0; s < p.n_steps; ++s) {
    const int step = p.step[s], cidx = p.cidx[s];
    const float tf = (float)step / p.t_div;
    for (int bi = 0; bi < p.n_blocks; ++bi) {
      const vsp_tacc_block& k = p.blk[bi];
      const bool last = bi == p.n_blocks - 1;
      // The thread index is re-made OPAQUE per block evaluation: otherwise the compiler hoists every lane-dependent byte offset of the
      // three phases (about 60 of them) out of the step / block loops, spills them, and reloads them at their uses with a scratch load
      // + vmcnt(0) -- which also waits for the weight prefetch in flight.  Recomputing an offset is one VALU instruction.
      int tidz;
#ifndef VSP_TP_NO_OPAQUE
      asm volatile("v_mov_b32 %0, %1" : "=v"(tidz) : "v"(tid0));
#else
      tidz = tid0;
#endif
      const int tid = tidz, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
      // ------------------------------------------------------------------------------------------------ proj
      {
        float* red = scr;   // two [partial sums | rows 16 / 17] buffer pairs
        float4 rn[PJ_NS];   // PixelNorm factors of this lane's k columns (the passes re-read y from LDS and scale it: 16 registers, not 48)
        float* y0 = ylds + lr * YP + 4 * kq;
        float* y16 = ylds + 16 * YP + 4 * kq;
        // PixelNorm once per block, IN PLACE: a wave normalises exactly the k columns it multiplies (all 18 rows), so no other wave reads
        // what it rewrites, and the passes re-read their A fragments from LDS instead of keeping 48 registers per lane alive next to two
        // weight register sets.  (The un-normalised y is not needed again: the post phase writes the next block's input.)
#pragma unroll
        for (int q = 0; q < PJ_NS; ++q) {
          const int k0 = (wave + 8 * q) * 16;
          float4 a = *reinterpret_cast<const float4*>(y0 + k0);
          float4 pp = *reinterpret_cast<const float4*>(y16 + k0);
          float4 qq = *reinterpret_cast<const float4*>(y16 + YP + k0);
          const float rx = rsqrtf((row16_sum(a.x * a.x) + fmaf(pp.x, pp.x, qq.x * qq.x)) * (1.f / NTOK) + 1e-8f);
          const float ry = rsqrtf((row16_sum(a.y * a.y) + fmaf(pp.y, pp.y, qq.y * qq.y)) * (1.f / NTOK) + 1e-8f);
          const float rz = rsqrtf((row16_sum(a.z * a.z) + fmaf(pp.z, pp.z, qq.z * qq.z)) * (1.f / NTOK) + 1e-8f);
          const float rw = rsqrtf((row16_sum(a.w * a.w) + fmaf(pp.w, pp.w, qq.w * qq.w)) * (1.f / NTOK) + 1e-8f);
          a.x *= rx; a.y *= ry; a.z *= rz; a.w *= rw;
          pp.x *= rx; pp.y *= ry; pp.z *= rz; pp.w *= rw;
          qq.x *= rx; qq.y *= ry; qq.z *= rz; qq.w *= rw;
#ifdef VSP_TP_INPLACE
          *reinterpret_cast<float4*>(y0 + k0) = a;
          if (lr == 0) {
            *reinterpret_cast<float4*>(y16 + k0) = pp;
            *reinterpret_cast<float4*>(y16 + YP + k0) = qq;
          }
#else
          rn[q] = make_float4(rx, ry, rz, rw);
#endif
        }
        // (every wave reads back only what it wrote itself: no workgroup barrier)
        constexpr int NPASS = 4 * D / 64 / G;   // 64-column passes over this workgroup's slice of the 2048 projection columns
        static_assert(NPASS % 2 == 0, "passes come in pairs (two weight register sets)");
        // The weights of pass i + 1 are loaded before the MFMAs of pass i (two register sets): with one set every pass began with an
        // exposed L2 round trip (cluster 4: 8 passes, 45 us per block against 15 us of MFMA issue).  The partial-sum buffers alternate
        // too, so a pass has ONE workgroup barrier and its reduction + stores run under the next pass's loads.
        auto load_w = [&](float4 (&bw)[PJ_NS][PJ_NJ], int pass) {
          const int n0 = g * (4 * D / G) + pass * 64;
          const float* w0 = k.wcat + (int64_t)(n0 + lr) * D + 4 * kq;
          // fragment-order copy: [n / 16][k / 16][lane][4] -- one wave instruction = 1 KiB of consecutive memory
          const float4* wf = reinterpret_cast<const float4*>(k.wcat_frag) + ((int64_t)(n0 >> 4) * (D / 16)) * 64 + lane;
#pragma unroll
          for (int q = 0; q < PJ_NS; ++q) {
            const int k0 = (wave + 8 * q) * 16;
#pragma unroll
            for (int j = 0; j < PJ_NJ; ++j)
              bw[q][j] = (VSP_TP_ABL & 2) ? make_float4(1.f, 1.f, 1.f, (float)k0)
                         : k.wcat_frag  ? wf[(j * (D / 16) + (k0 >> 4)) * 64]
                                        : *reinterpret_cast<const float4*>(w0 + (int64_t)j * 16 * D + k0);
          }
        };
        auto run_pass = [&](const float4 (&bw)[PJ_NS][PJ_NJ], float4 (&bwn)[PJ_NS][PJ_NJ], int pass) {
          const int n0 = g * (4 * D / G) + pass * 64;
          float* redp = red + (pass & 1) * (PJ_RED + 8 * 4 * 2 * PJ_NJ * 16);
          float* red2p = redp + PJ_RED;
          f32x4 acc[PJ_NJ];
          float e[2][PJ_NJ];
#pragma unroll
          for (int j = 0; j < PJ_NJ; ++j) {
            acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            e[0][j] = e[1][j] = 0.f;
          }
#pragma unroll
          for (int q = 0; q < PJ_NS; ++q) {
            // the next pass's weights leave AFTER this pass's first k-slice has been consumed: every older load has landed by then,
            // so the in-order vmcnt wait in front of the first MFMA never covers a load that was just issued
            if (q == 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (pass + 1 < NPASS) load_w(bwn, pass + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            const int k0 = (wave + 8 * q) * 16;
            float4 a = *reinterpret_cast<const float4*>(y0 + k0), pp = *reinterpret_cast<const float4*>(y16 + k0),
                   qq = *reinterpret_cast<const float4*>(y16 + YP + k0);
#ifndef VSP_TP_INPLACE
            a.x *= rn[q].x; a.y *= rn[q].y; a.z *= rn[q].z; a.w *= rn[q].w;
            pp.x *= rn[q].x; pp.y *= rn[q].y; pp.z *= rn[q].z; pp.w *= rn[q].w;
            qq.x *= rn[q].x; qq.y *= rn[q].y; qq.z *= rn[q].z; qq.w *= rn[q].w;
#endif
#pragma unroll
            for (int j = 0; j < PJ_NJ; ++j) {
              const float4 w = bw[q][j];
              if (VSP_TP_ABL & 1) { acc[j][0] += w.x + w.y + w.z + w.w + a.x; continue; }
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc[j], 0, 0, 0);
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc[j], 0, 0, 0);
              e[0][j] = fmaf(pp.x, w.x, fmaf(pp.y, w.y, fmaf(pp.z, w.z, fmaf(pp.w, w.w, e[0][j]))));
              e[1][j] = fmaf(qq.x, w.x, fmaf(qq.y, w.y, fmaf(qq.z, w.z, fmaf(qq.w, w.w, e[1][j]))));
            }
          }
#pragma unroll
          for (int j = 0; j < PJ_NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) redp[((wave * PJ_NJ + j) * 4 + r) * 64 + lane] = acc[j][r];
            red2p[(((wave * 4 + kq) * 2 + 0) * PJ_NJ + j) * 16 + lr] = e[0][j];
            red2p[(((wave * 4 + kq) * 2 + 1) * PJ_NJ + j) * 16 + lr] = e[1][j];
          }
          __syncthreads();   // (the other buffer pair was last read before the previous pass's barrier)
          if (VSP_TP_ABL & 4) return;
          for (int jr = wave; jr < PJ_NJ * 4; jr += 8) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += redp[(w * PJ_NJ * 4 + jr) * 64 + lane];
            cst1(Prs, ((rowb + kq * 4 + (jr & 3)) * 4 * D + n0 + (jr >> 2) * 16 + lr) * 4, v);
          }
          if (wave < 2) {
            for (int j = kq; j < PJ_NJ; j += 4) {
              float v = 0.f;
#pragma unroll
              for (int i = 0; i < 8 * 4; ++i) v += red2p[((i * 2 + wave) * PJ_NJ + j) * 16 + lr];
              cst1(Prs, ((rowb + 16 + wave) * 4 * D + n0 + j * 16 + lr) * 4, v);
            }
          }
        };
        float4 bwA[PJ_NS][PJ_NJ], bwB[PJ_NS][PJ_NJ];
        load_w(bwA, 0);
        for (int pass = 0; pass < NPASS; pass += 2) {
          run_pass(bwA, bwB, pass);
          run_pass(bwB, bwA, pass + 1);
        }
        __syncthreads();   // the scratch area changes hands
      }
      TP_MARK(0)
      cluster_barrier<G>(cnt, tmo, ++bar);
      TP_MARK(1)
      // ------------------------------------------------------------------------------------------------ attn: channel attention
      {
        constexpr int NW = 8, MT = D / 16 / NW;
        constexpr int KP = vsptacc::CA_KP, VP = vsptacc::CA_VP;
        float* k2s = scr;                    // [20][KP], token rows 18, 19 zero
        float* v2s = k2s + 20 * KP;          // [18][VP]
        float* redm = v2s + NTOK * VP;       // [NW][32]
        float* reds = redm + NW * 32;        // [NW][32]
        float* tpart = reds + NW * 32;       // [NW-1][4][4][64]
        const float scale = 0.044194173824159216f;   // 1 / sqrt(512)
        {
          constexpr int TR = NTH / 128, NIT = (20 + TR - 1) / TR;
          const int c4 = tid & 127, tr = tid >> 7;
          const float4 w = *reinterpret_cast<const float4*>(k.wk + c4 * 4);
          float4 kreg[NIT], vreg[NIT];
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int tok = tr + TR * it;
            const int tc = tok < NTOK ? tok : NTOK - 1;
            kreg[it] = *reinterpret_cast<const float4*>(k.ek + ((int64_t)rowb + tc) * D + c4 * 4);
            vreg[it] = cld4(Prs, ((rowb + tc) * 4 * D + 3 * D + c4 * 4) * 4);
          }
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int tok = tr + TR * it;
            if (tok >= 20) continue;
            float4 v = kreg[it];
            v.x = fmaf(tf, w.x, v.x); v.y = fmaf(tf, w.y, v.y); v.z = fmaf(tf, w.z, v.z); v.w = fmaf(tf, w.w, v.w);
            if (tok >= NTOK) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(k2s + tok * KP + c4 * 4) = v;
            if (tok < NTOK) {
              float* d = v2s + tok * VP + c4 * 4;
              d[0] = vreg[it].x; d[1] = vreg[it].y; d[2] = vreg[it].z; d[3] = vreg[it].w;
            }
          }
        }
        __syncthreads();
        constexpr int NCB = 16 / G;   // 32-column blocks of the channel attention per workgroup (k2 / v2 are staged once)
        for (int cbi = 0; cbi < NCB; ++cbi) {
        const int cb = g * NCB + cbi;
        float qb[5][2];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int tok = 4 * q + kq;
            const float v = cld1(Prs, ((rowb + (tok < NTOK ? tok : 0)) * 4 * D + 2 * D + cb * 32 + nt * 16 + lr) * 4);
            qb[q][nt] = tok < NTOK ? v * scale : 0.f;
          }
        f32x4 L[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          L[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
          L[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const float a = k2s[(4 * q + kq) * KP + (wave * MT + mt) * 16 + lr];
            L[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qb[q][0], L[mt][0], 0, 0, 0);
            L[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qb[q][1], L[mt][1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float m = L[0][nt][0];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, L[mt][nt][j]);
          m = fmaxf(m, __shfl_xor(m, 16, 64));
          m = fmaxf(m, __shfl_xor(m, 32, 64));
          if (kq == 0) redm[wave * 32 + nt * 16 + lr] = m;
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float m = redm[nt * 16 + lr];
#pragma unroll
          for (int w = 1; w < NW; ++w) m = fmaxf(m, redm[w * 32 + nt * 16 + lr]);
          float sm = 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float ev = expf(L[mt][nt][j] - m);
              L[mt][nt][j] = ev;
              sm += ev;
            }
          sm += __shfl_xor(sm, 16, 64);
          sm += __shfl_xor(sm, 32, 64);
          if (kq == 0) reds[wave * 32 + nt * 16 + lr] = sm;
        }
        f32x4 T[2][2];
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) T[mt2][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tok1 = 16 + lr;
        const bool ok1 = tok1 < NTOK;
        const float* va0 = v2s + lr * VP + wave * (MT * 16) + kq * 4;
        const float* va1 = v2s + (ok1 ? tok1 : NTOK - 1) * VP + wave * (MT * 16) + kq * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a0v = va0[mt * 16 + j];
            const float a1v = ok1 ? va1[mt * 16 + j] : 0.f;
            T[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, L[mt][0][j], T[0][0], 0, 0, 0);
            T[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v, L[mt][1][j], T[0][1], 0, 0, 0);
            T[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, L[mt][0][j], T[1][0], 0, 0, 0);
            T[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, L[mt][1][j], T[1][1], 0, 0, 0);
          }
        if (wave > 0) {
#pragma unroll
          for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int j = 0; j < 4; ++j) tpart[(((wave - 1) * 4 + mt2 * 2 + nt) * 4 + j) * 64 + lane] = T[mt2][nt][j];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            float den = reds[nt * 16 + lr];
#pragma unroll
            for (int w = 1; w < NW; ++w) den += reds[w * 32 + nt * 16 + lr];
            const float inv = 1.f / den;
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float v = T[mt2][nt][j];
#pragma unroll
                for (int w = 1; w < NW; ++w) v += tpart[(((w - 1) * 4 + mt2 * 2 + nt) * 4 + j) * 64 + lane];
                const int tok = mt2 * 16 + kq * 4 + j;
                if (tok < NTOK) cst1(trs, ((rowb + tok) * D + cb * 32 + nt * 16 + lr) * 4, v * inv);
              }
          }
        }
        if (cbi + 1 < NCB) __syncthreads();   // redm / reds / tpart are rewritten by the next column block
        }
        // ---------------------------------------------------------------------------------------------- attn: token attention
        // rows g, g + G, ... of this image, one wave each, starting at wave 1 (wave 0 is finishing the channel attention)
        for (int trow = g + G * ((wave + 7) & 7); trow < NTOK; trow += 8 * G) {
          const float sscale = 0.23570226039551584f;  // 1 / sqrt(18)
          float4 kv[2], wv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            kv[u] = cld4(Prs, ((rowb + trow) * 4 * D + 4 * lane + 256 * u) * 4);
            wv[u] = *reinterpret_cast<const float4*>(k.wq + 4 * lane + 256 * u);
            wv[u].x *= tf; wv[u].y *= tf; wv[u].z *= tf; wv[u].w *= tf;
          }
          float sc[NTOK];
#pragma unroll
          for (int j = 0; j < NTOK; ++j) {
            float a = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float4 q = *reinterpret_cast<const float4*>(k.eQ + ((int64_t)rowb + j) * D + 4 * lane + 256 * u);
              a = fmaf(kv[u].x, q.x + wv[u].x, a);
              a = fmaf(kv[u].y, q.y + wv[u].y, a);
              a = fmaf(kv[u].z, q.z + wv[u].z, a);
              a = fmaf(kv[u].w, q.w + wv[u].w, a);
            }
            sc[j] = a;
          }
#pragma unroll
          for (int j = 0; j < NTOK; ++j) sc[j] = wave_sum(sc[j]) * sscale;
          float m = sc[0];
#pragma unroll
          for (int j = 1; j < NTOK; ++j) m = fmaxf(m, sc[j]);
          float den = 0.f;
#pragma unroll
          for (int j = 0; j < NTOK; ++j) {
            sc[j] = expf(sc[j] - m);
            den += sc[j];
          }
          const float rden = 1.f / den;
          float4 hacc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
          for (int j = 0; j < NTOK; ++j) {
            const float pj = sc[j] * rden;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float4 v = cld4(Prs, ((rowb + j) * 4 * D + D + 4 * lane + 256 * u) * 4);
              hacc[u].x = fmaf(pj, v.x, hacc[u].x);
              hacc[u].y = fmaf(pj, v.y, hacc[u].y);
              hacc[u].z = fmaf(pj, v.z, hacc[u].z);
              hacc[u].w = fmaf(pj, v.w, hacc[u].w);
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) cst4(hrs, ((rowb + trow) * D + 4 * lane + 256 * u) * 4, hacc[u]);
        }
      }
      TP_MARK(2)
      cluster_barrier<G>(cnt, tmo, ++bar);
      TP_MARK(3)
      // ------------------------------------------------------------------------------------------------ post (all 18 rows, every workgroup)
      {
        const int64_t hoff = (int64_t)step * p.head_stride;
        const bool mix = last && p.c1 != nullptr;
        const float a1 = mix ? p.c1[cidx] : 1.f, a2 = mix ? p.c2[cidx] : 0.f;
        const int xsrc = (s & 1) * M * D;            // X(s) for s >= 1; X(0) is xio (read with plain loads: written before the launch)
        for (int r = wave; r < NTOK; r += 8) {
          const int o = (rowb + r) * D + 4 * lane;
          float4 tv[2], hv[2], gv[2], bv[2], xv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            tv[u] = cld4(trs, (o + 256 * u) * 4);
            hv[u] = cld4(hrs, (o + 256 * u) * 4);
            gv[u] = *reinterpret_cast<const float4*>(k.gamma + hoff + o + 256 * u);
            bv[u] = *reinterpret_cast<const float4*>(k.beta + hoff + o + 256 * u);
            xv[u] = !mix ? make_float4(0.f, 0.f, 0.f, 0.f)
                         : (s == 0 ? *reinterpret_cast<const float4*>(p.xio + o + 256 * u) : cld4(xrs, (xsrc + o + 256 * u) * 4));
          }
          auto stats = [&](const float4 (&v)[2], float& mean, float& inv) {
            const float sm = (v[0].x + v[0].y) + (v[0].z + v[0].w) + (v[1].x + v[1].y) + (v[1].z + v[1].w);
            mean = wave_sum(sm) * (1.f / D);
            float var = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float d0 = v[u].x - mean, d1 = v[u].y - mean, d2 = v[u].z - mean, d3 = v[u].w - mean;
              var = fmaf(d0, d0, var); var = fmaf(d1, d1, var); var = fmaf(d2, d2, var); var = fmaf(d3, d3, var);
            }
            inv = rsqrtf(wave_sum(var) * (1.f / D) + 1e-5f);
          };
          float mean, inv;
          stats(tv, mean, inv);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            hv[u].x += (tv[u].x - mean) * inv; hv[u].y += (tv[u].y - mean) * inv;
            hv[u].z += (tv[u].z - mean) * inv; hv[u].w += (tv[u].w - mean) * inv;
          }
          stats(hv, mean, inv);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float4 yv;
            yv.x = (hv[u].x - mean) * inv * (1.f + gv[u].x) + bv[u].x;
            yv.y = (hv[u].y - mean) * inv * (1.f + gv[u].y) + bv[u].y;
            yv.z = (hv[u].z - mean) * inv * (1.f + gv[u].z) + bv[u].z;
            yv.w = (hv[u].w - mean) * inv * (1.f + gv[u].w) + bv[u].w;
            if (mix) {
              yv.x = a1 * yv.x + a2 * xv[u].x; yv.y = a1 * yv.y + a2 * xv[u].y;
              yv.z = a1 * yv.z + a2 * xv[u].z; yv.w = a1 * yv.w + a2 * xv[u].w;
            }
            *reinterpret_cast<float4*>(ylds + r * YP + 4 * lane + 256 * u) = yv;
            if (last && g == 0 && s + 1 < p.n_steps && p.c1 != nullptr)   // X(s+1), read by every workgroup of the image at the end of step s+1
              cst4(xrs, (((s + 1) & 1) * M * D + o + 256 * u) * 4, yv);
          }
        }
        __syncthreads();   // y complete in LDS before the next projection reads it
      }
      TP_MARK(4)
    }
  }
  // result: the leader's copy of y (= x after the last step).  Every workgroup has finished READING xio long ago unless the chain
  // had a single step (x(0) is read in the last post): one more barrier keeps that case safe.
#ifdef VSP_TP_TIMING
  if (wid == 0 && tid == 0)
    for (int i = 0; i < 5; ++i) p.sync[40 + i] = (unsigned)(tph[i] >> 6);   // units of 64 cycles of the 100 MHz counter... (s_memtime: constant clock)
#endif
  cluster_barrier<G>(cnt, tmo, ++bar);
  if (g == 0)
    for (int i = tid; i < NTOK * D / 4; i += NTH)
      reinterpret_cast<float4*>(p.xio + (int64_t)rowb * D)[i] = *reinterpret_cast<const float4*>(ylds + (i >> 7) * YP + (i & 127) * 4);
}

}  // namespace

extern "C" {

size_t vsp_tacc_chain_persistent_work_floats(int B) {
  if (B <= 0) return 0;
  const size_t M = (size_t)B * NTOK;
  return M * 4 * D /* P */ + M * D /* t */ + M * D /* h */ + 2 * M * D /* X parity buffers */ + 64 /* counters + timeout word */;
}

int vsp_tacc_chain_persistent_f32(const vsp_tacc_chain_params* pp, vsp_stream_t stream) {
  return vsp_tacc_chain_cluster_f32(pp, 16, stream);
}

int vsp_tacc_chain_cluster_f32(const vsp_tacc_chain_params* pp, int cluster, vsp_stream_t stream) {
  VSP_REQUIRE(pp != nullptr, "tacc_chain_persistent: null params");
  VSP_REQUIRE(cluster == 16 || cluster == 8 || cluster == 4 || cluster == 2 || cluster == 1,
              "tacc_chain_persistent: cluster size must be 1, 2, 4, 8 or 16 workgroups per image (got %d)", cluster);
  const vsp_tacc_chain_params& p = *pp;
  VSP_REQUIRE(p.n_tok == NTOK && p.dim == D, "tacc_chain_persistent: built for 18 tokens x 512 channels (got %d x %d)", p.n_tok, p.dim);
  VSP_REQUIRE(p.B >= 0 && p.n_blocks >= 0 && p.n_steps >= 0, "tacc_chain_persistent: negative size");
  if (p.B == 0 || p.n_blocks == 0 || p.n_steps == 0) return VSP_OK;
  if (p.B * cluster > vsp::kNumCU || p.B > 32 || p.n_blocks > 4)   // (32: the arrival counters of the sync area)
    return vsp::fail(VSP_ENOTSUP, "tacc_chain_persistent: at most %d images (one workgroup per CU, %d per image; 32 in any case) and 4 blocks (got %d, %d)",
                     vsp::kNumCU / cluster, cluster, p.B, p.n_blocks);
  VSP_REQUIRE(p.blocks && p.x && p.work && p.step, "tacc_chain_persistent: null pointer");
  VSP_REQUIRE(p.work_floats >= vsp_tacc_chain_persistent_work_floats(p.B), "tacc_chain_persistent: work buffer too small");
  VSP_REQUIRE(p.t_div > 0.f, "tacc_chain_persistent: t_div must be positive");
  VSP_REQUIRE(!p.c1 == !p.c2, "tacc_chain_persistent: c1 and c2 come together");
  VSP_REQUIRE(vsp::aligned16(p.x) && vsp::aligned16(p.work), "tacc_chain_persistent: x and work must be 16-byte aligned");
  for (int i = 0; i < p.n_blocks; ++i) {
    const vsp_tacc_block& k = p.blocks[i];
    VSP_REQUIRE(k.wcat && k.eQ && k.ek && k.wq && k.wk && k.gamma && k.beta, "tacc_chain_persistent: block %d has a null pointer", i);
    VSP_REQUIRE(vsp::aligned16(k.wcat) && vsp::aligned16(k.eQ) && vsp::aligned16(k.ek) && vsp::aligned16(k.wq) &&
                    vsp::aligned16(k.wk) && vsp::aligned16(k.gamma) && vsp::aligned16(k.beta) && vsp::aligned16(k.wcat_frag),
                "tacc_chain_persistent: block %d operands must be 16-byte aligned", i);
  }
  static vsp::LdsAttrOnce attr[5];
  const void* kfn = cluster == 16 ? reinterpret_cast<const void*>(tacc_persist_kernel<16>)
                    : cluster == 8 ? reinterpret_cast<const void*>(tacc_persist_kernel<8>)
                    : cluster == 4 ? reinterpret_cast<const void*>(tacc_persist_kernel<4>)
                    : cluster == 2 ? reinterpret_cast<const void*>(tacc_persist_kernel<2>)
                                   : reinterpret_cast<const void*>(tacc_persist_kernel<1>);
  if (int rc = attr[cluster == 16 ? 0 : cluster == 8 ? 1 : cluster == 4 ? 2 : cluster == 2 ? 3 : 4].ensure(kfn, (int)LDS_BYTES, "tacc_chain_persistent")) return rc;
  hipStream_t st = vsp::as_stream(stream);
  const size_t M = (size_t)p.B * NTOK;
  PK q{};
  q.B = p.B; q.n_blocks = p.n_blocks;
  for (int i = 0; i < p.n_blocks; ++i) q.blk[i] = p.blocks[i];
  q.xio = p.x;
  q.P = p.work;
  q.tb = q.P + M * 4 * D;
  q.hb = q.tb + M * D;
  q.xb = q.hb + M * D;
  q.sync = reinterpret_cast<unsigned*>(q.xb + 2 * M * D);
  q.c1 = p.c1; q.c2 = p.c2; q.t_div = p.t_div;
  q.head_stride = (int64_t)M * D;
  // at most MAXS steps per launch (the step list travels in the kernel arguments); longer chains continue in further launches
  for (int s0 = 0; s0 < p.n_steps; s0 += MAXS) {
    const int ns = p.n_steps - s0 < MAXS ? p.n_steps - s0 : MAXS;
    q.n_steps = ns;
    for (int s = 0; s < ns; ++s) {
      const int step = p.step[s0 + s];
      VSP_REQUIRE(step >= 0 && step < p.head_steps && step < 32768, "tacc_chain_persistent: step %d outside the prepared heads [0, %d)", step, p.head_steps);
      q.step[s] = (short)step;
      const int ci = p.coef_idx ? p.coef_idx[s0 + s] : step;
      VSP_REQUIRE(ci >= 0 && ci < 32768, "tacc_chain_persistent: coefficient index %d out of range", ci);
      q.cidx[s] = (short)ci;
    }
    if (hipMemsetAsync(q.sync, 0, 64 * sizeof(unsigned), st) != hipSuccess) return vsp::fail(VSP_ELAUNCH, "tacc_chain_persistent: memset failed");
    switch (cluster) {
      case 16: tacc_persist_kernel<16><<<16 * p.B, NTH, LDS_BYTES, st>>>(q); break;
      case 8: tacc_persist_kernel<8><<<8 * p.B, NTH, LDS_BYTES, st>>>(q); break;
      case 4: tacc_persist_kernel<4><<<4 * p.B, NTH, LDS_BYTES, st>>>(q); break;
      case 2: tacc_persist_kernel<2><<<2 * p.B, NTH, LDS_BYTES, st>>>(q); break;
      default: tacc_persist_kernel<1><<<p.B, NTH, LDS_BYTES, st>>>(q); break;
    }
  }
  return vsp::check_launch("tacc_chain_persistent");
}

}  // extern "C"

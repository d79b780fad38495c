// Flash attention forward for gfx950: varlen windows, GQA, optional bottom-right causal mask.
// Replaces flash_attn_varlen_func (reference modeling/g2vlm/qwen2vl.py:643-652,
// modeling/g2vlm/dinov2_model.py:49-58, modeling/qwen2vl/modeling_qwen2_vl.py:400) and the SDPA
// flash calls of the Pi3 decoders (modeling/pi3/models/layers/attention.py:255-264, 370-375).
//
// Structure (MI355X guide, "Fused attention prefill"): block = 4 waves x 32 query rows; K/V tiles of
// 64 keys double-buffered in LDS (register-staged, issue-early / write-late); S^T = K.Q^T with
// mfma_f32_32x32x16_bf16 so that a lane owns ONE query column and 32 of the tile's 64 keys in
// registers -> the softmax row reduction is in-register plus one lane^32 exchange; the S^T
// accumulator is re-used directly as the B operand of O^T += V^T.P^T (guide §3 "accumulator tile as
// the next MFMA's operand", permuted-k order), with V^T fragments fetched by ds_read_b64_tr_b16 from
// a row-major V tile.  LDS image = 256-byte rows with the dual-use XOR (guide T10 layout (b)),
// conflict-free for both the b128 row reads of K and the transposed reads of V.
//
// Numerics: S and softmax statistics in fp32, P rounded to bf16 for the PV product (as flash-attn
// does), O accumulated in fp32 and normalised once at the end.
#include <type_traits>
#include "common.h"
#include "g2vlm_hip.h"

namespace {

// timing experiments only (tools/attn_variants.sh builds the library once per macro; results are wrong with EXP_*)
#ifdef EXP_NO_EXP
#define G2V_EXP2(x) (x)
#else
#define G2V_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

constexpr int KV_TILE = 64;
constexpr int ROWB = 256;                      // LDS bytes per key row (head dim padded to 128)
constexpr int TILE_B = KV_TILE * ROWB;         // 16 KiB

struct FlashArgs {
  const __bf16* q; const __bf16* k; const __bf16* v; __bf16* o;
  const g2v_attn_tile* tiles;
  const g2v_attn_seg* segs;   // the schedule: workgroup lb runs segments [seg_ptr[lb], seg_ptr[lb + 1])
  const int* seg_ptr;         // [n_blocks + 1]
  const int* comb;            // [4 * n_comb] (descriptor, head, first slot, slots) of every output tile that is merged from partials
  float* ws;                  // partial results: slot s at ws + s * SLOT_FLOATS
  int ldq, ldk, ldv, ldo, n_tiles, Hq, Hkv, n_blocks;
  float scale_log2;
};

constexpr int SLOT_ROWS = 256;                  // query rows per item at most (8 waves x 32)
constexpr int SLOT_FLOATS = SLOT_ROWS * 130;    // m[256], l[256], O[256][128]

// Dual-use LDS image (guide T10, layout (a)): 8-row x 32-column subtiles of 512 B.  Conflict-free for the ds_read_b128
// row reads of K and the ds_read_b64_tr_b16 transposed reads of V, and every fragment address of a wave is one of TWO
// per-lane bases plus an immediate (layout (b), plain 256-byte rows, needs 8 bases per read kind: ~30 VGPRs more).
__device__ __forceinline__ int lds_off(int row, int ch) {
  return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// Persistent schedule: the host (g2vlm_amd/hip.py::make_attn_plan) hands every workgroup a list of SEGMENTS = (tile
// descriptor, head, KV-tile range, output slot).  A segment that covers its item's whole KV range writes the normalised
// output; any other leaves unnormalised partials (m, l, O) in its workspace slot and flash_combine_kernel merges an
// item's slots.  Whole items go to workgroups as long as they divide evenly; only the remainder is cut (stream-K), so
// every CU does the same number of KV tiles whatever the item count while few items pay the partial round trip
// (cutting EVERY workgroup's range, the first form, wrote and re-read 151 MB of partials per launch at C3).
template <int D, int NW>
__global__ __launch_bounds__(64 * NW, 2) void flash_fwd_kernel(FlashArgs a) {
  constexpr int KSTEPS = D / 16;               // k-steps of the QK^T product
  constexpr int DBLK = (D + 31) / 32;          // 32-wide d blocks of O^T
  constexpr int CH = D / 8;                    // 16-byte chunks per row
  constexpr int NT = 64 * NW;                  // threads per block
  // K/V tile ring: 3 slots for the 8-wave form (one workgroup per CU: the DMA of tile t+2 is issued at the start of
  // tile t, so a tile has two tile-times to land and the per-tile barrier only absorbs wave skew), 2 slots for the 4-wave
  // form (two workgroups per CU share the 160 KiB)
  constexpr int SLOTS = (NW == 8 && D == 128) ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char smem[SLOTS * 2 * TILE_B];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // XCD-aware bijective remap: blocks sharing an XCD (blockIdx % 8) take a contiguous range of logical blocks, i.e.
  // ~1.5 query heads of ONE kv head, whose 5.6 MB of K/V then live in that XCD's L2
  int lb = blockIdx.x;
  {
    int xcd = lb & 7, qn = a.n_blocks >> 3, rn = a.n_blocks & 7;
    lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (lb >> 3);
  }
  const int seg_end = a.seg_ptr[lb + 1];

  // ---- K/V staging by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  One wave-instruction
  // writes 1 KiB lane-linear = two 512-byte subtiles = 8 rows x 64 columns of the layout-(a) image, so the image's XOR
  // goes on the per-lane SOURCE address (guide rule 21) and each row contributes one full 128-byte line.
  constexpr int NCG2 = (((CH + 3) / 4) + 1) / 2;           // 1-KiB pieces per 8-row group (2 for D > 64)
  constexpr int NPW = (8 * NCG2 + NW - 1) / NW;            // pieces per wave per operand tile
  const int wu = __builtin_amdgcn_readfirstlane(w);
  int p_dst[NPW], p_row[NPW], p_col[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = min(wu + NW * i, 8 * NCG2 - 1);         // surplus waves repeat the last piece (same bytes, harmless)
    const int g = pi / NCG2, h2 = pi - g * NCG2;
    const int cg = 2 * h2 + (lane >> 5), row8 = (lane & 31) >> 2, slot = lane & 3;
    p_dst[i] = 2048 * g + 1024 * h2;
    p_row[i] = 8 * g + row8;
    p_col[i] = 8 * min(4 * cg + (slot ^ ((p_row[i] >> 2) & 3)), CH - 1);
  }
  // K row reads (A operand of S^T = K.Q^T): row 32b + r, chunk 2ks + hh -> k_lb[ks & 1] + 8192 b + 512 (ks >> 1)
  int k_lb[2];
  k_lb[0] = 2048 * (r >> 3) + 64 * (r & 7) + 16 * (hh ^ ((r >> 2) & 3));
  k_lb[1] = k_lb[0] ^ 32;
  // V transposed reads (A operand of O^T += V^T.P^T): row 32b + 16s + 8jj + 4hh + tq, chunk 4d + t_ch
  //   -> v_lb[jj] + 2048 (4b + 2s + jj) + 512 d
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int t_ch = 2 * ((lane >> 4) & 1) + (tp >> 1);
  int v_lb[2];
  v_lb[0] = 64 * (4 * hh + tq) + 16 * (t_ch ^ hh) + 8 * (tp & 1);
  v_lb[1] = v_lb[0] ^ 32;
  const float c = a.scale_log2;

  for (int si = a.seg_ptr[lb]; si < seg_end; ++si) {
    // ---- the segment: (descriptor, head, kt range, output slot)
    const g2v_attn_seg sg = a.segs[si];
    const int head = sg.head, kt0 = sg.kt0, kt1 = sg.kt1;
    const g2v_attn_tile T = a.tiles[sg.desc];
    const int kvh = head / (a.Hq / a.Hkv);

    // ---- Q fragments (B operand: Q^T[k=d][col=query]) straight from global
    const int qi = min(32 * w + r, T.q_rows - 1);          // clamp: padded rows duplicate the last one
    const __bf16* qp = a.q + (size_t)(T.q0 + qi) * a.ldq + head * D + 8 * hh;
    bf16x8 qf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    const int q_rel = (T.q0 - T.q_win0) + 32 * w + r;     // query index inside its window
    const int kmax_row = q_rel + T.causal_shift;          // last allowed key (window-local), may be huge

    const __bf16* kbase = a.k + (size_t)T.k0 * a.ldk + kvh * D;
    const __bf16* vbase = a.v + (size_t)T.k0 * a.ldv + kvh * D;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    auto stage = [&](int kt, int buf) {
      char* sk = smem + buf * 2 * TILE_B;
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int kr = min(kt * KV_TILE + p_row[i], T.k_len - 1);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kbase + (size_t)kr * a.ldk + p_col[i]), (lds_ptr_t)(sk + p_dst[i]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vbase + (size_t)kr * a.ldv + p_col[i]), (lds_ptr_t)(sk + TILE_B + p_dst[i]), 16, 0, 0);
      }
    };

    f32x16 O[DBLK];
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;                     // m_run: running max of the RAW scores

    // s_waitcnt vmcnt(n): leave the n youngest DMA instructions of this wave in flight (n = 0 or one tile's 2 NPW)
    auto wait_tile = [&](bool one_tile_in_flight) {
      if (!one_tile_in_flight) __builtin_amdgcn_s_waitcnt(0x0F70);
      else if constexpr (2 * NPW == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
      else if constexpr (2 * NPW == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
      else __builtin_amdgcn_s_waitcnt(0x0F78);
    };
    stage(kt0, 0);
    if (SLOTS == 3 && kt0 + 1 < kt1) stage(kt0 + 1, 1);
    wait_tile(false);
    __builtin_amdgcn_s_barrier();

    // S^T = K . Q^T of one tile (first k-step starts from the constant-0 accumulator).  K fragments run two k-steps ahead
    // of their MFMAs through a 3-deep register window, so LDS latency is covered by MFMA time instead of a wait per step.
    auto qk_tile = [&](const char* sK, f32x16 (&S)[2]) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      bf16x8 kf[3][2];
      auto kread = [&](int ks, int b) {
#ifdef EXP_NO_KREAD
        return qf[(ks + b) % KSTEPS];
#else
        return *reinterpret_cast<const bf16x8*>(sK + k_lb[ks & 1] + 8192 * b + 512 * (ks >> 1));
#endif
      };
#pragma unroll
      for (int ks = 0; ks < 2 && ks < KSTEPS; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b) kf[ks][b] = kread(ks, b);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + 2 < KSTEPS) {
#pragma unroll
          for (int b = 0; b < 2; ++b) kf[(ks + 2) % 3][b] = kread(ks + 2, b);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
          S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks % 3][b], qf[ks], ks == 0 ? zero : S[b], 0, 0, 0);
      }
    };

    // PIPE (8-wave form, 3-slot ring): the scores of tile kt+1 are multiplied INSIDE iteration kt, in the same basic
    // block as the exponentials of tile kt (guide T15): the two are independent, so the MFMA pipe works on QK^T while the
    // VALU does the softmax instead of the two taking turns.  S_cur holds tile kt's scores on entry.
    constexpr bool PIPE = SLOTS == 3;                      // 8 waves, head dim 128 (KSTEPS = 8 slices of 4 scores)
#ifdef EXP_SETPRIO
    if (wu >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    f32x16 S_a[2], S_b[2];                                 // ping-pong: scores of the current / the next tile
    if constexpr (PIPE) qk_tile(smem, S_a);

    // one KV tile; MASKED (tile crosses the window end or the causal diagonal) is a separate instantiation run by a
    // separate loop below, so full tiles pay no compare/select per score and the two forms never meet in a phi
    auto tile_step = [&](int kt, auto masked_t, f32x16 (&S)[2], f32x16 (&S_next)[2]) {
      constexpr bool MASKED = decltype(masked_t)::value;
      const int buf = (kt - kt0) % SLOTS;
      const char* sK = smem + buf * 2 * TILE_B;
      const char* sV = sK + TILE_B;
      // the slot being refilled was last read one barrier ago (tile kt-1)
      if (kt + SLOTS - 1 < kt1) stage(kt + SLOTS - 1, (kt - kt0 + SLOTS - 1) % SLOTS);

      if constexpr (!PIPE) qk_tile(sK, S);
      float l_part = 0.f;                                  // PIPE: this tile's row sum, finished in the PV phase

      // ---- mask, online softmax in the log2 domain: p = exp2(s*c - m*c).  The masked form (tile crosses the window end
      // or the causal diagonal; wave-uniform) is a separate instantiation: full tiles pay no compare/select per element.
      const int kb = kt * KV_TILE;
      // PIPE: the next tile's first K fragments are requested before this tile's row maxima are reduced, so the first
      // QK^T slices below do not start behind an LDS round trip; the window runs KW - 1 k-steps ahead of the MFMAs
#ifdef EXP_KW
      constexpr int KW = EXP_KW;
#else
      constexpr int KW = 4;
#endif
      bf16x8 kfp[KW][2];
      const char* sKn = smem + ((kt + 1 - kt0) % SLOTS) * 2 * TILE_B;
      auto kread_n = [&](int ks, int b) {
#ifdef EXP_NO_KREAD
        return qf[(ks + b) % KSTEPS];
#else
        return *reinterpret_cast<const bf16x8*>(sKn + k_lb[ks & 1] + 8192 * b + 512 * (ks >> 1));
#endif
      };
      if constexpr (PIPE) {
#pragma unroll
        for (int ks = 0; ks < KW - 1; ++ks)
#pragma unroll
          for (int b = 0; b < 2; ++b) kfp[ks][b] = kread_n(ks, b);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        float rmax = -1e30f;
        if constexpr (MASKED) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              int key = kb + 32 * b + (e & 3) + 8 * (e >> 2) + 4 * hh;
              if (key >= T.k_len || key > kmax_row) S[b][e] = -1e30f;
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) rmax = fmaxf(rmax, S[b][e]);
        {  // lane <-> lane^32 exchange on the VALU (v_permlane32_swap) instead of an LDS bpermute
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
          rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_run, rmax);
        if (__any(m_new != m_run)) {                         // rescale only when some row's max moved (wave-uniform)
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
          l_run *= alpha;
#pragma unroll
          for (int d = 0; d < DBLK; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[d][e] *= alpha;
          m_run = m_new;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        if constexpr (PIPE) {
          // The next tile's 16 QK^T MFMAs and this tile's 32 exponentials, hand-interleaved in eight slices fenced by
          // sched_barrier(0): slice ks = {K fragments of k-step ks+2, two MFMAs of k-step ks, exp2/sum of four scores}.
          // Left to itself hipcc issues the MFMAs back to back behind one LDS wait each and packs the VALU elsewhere;
          // in this order the matrix pipe and the VALU of one wave work at the same time (guide T15 / T19).
          // (Unconditional: after the last tile the slot is stale and S_next is dropped - a branch would split the block.)
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            if (ks + KW - 1 < KSTEPS) {
#pragma unroll
              for (int b = 0; b < 2; ++b) kfp[(ks + KW - 1) % KW][b] = kread_n(ks + KW - 1, b);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
              S_next[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfp[ks % KW][b], qf[ks], ks == 0 ? zero : S_next[b], 0, 0, 0);
            // the b = 0 half of this tile's scores (needed by the first eight PV MFMAs): two per slice; the b = 1 half is
            // exponentiated between those eight MFMAs below, so both MFMA phases carry the same VALU load
#pragma unroll
            for (int e = 2 * ks; e < 2 * ks + 2; ++e) {
              float p = G2V_EXP2(fmaf(S[0][e], c, -mc));
              if constexpr (MASKED) { if (S[0][e] <= -1e30f) p = 0.f; }
              S[0][e] = p;
              psum += p;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float p = __builtin_amdgcn_exp2f(fmaf(S[b][e], c, -mc));
              if constexpr (MASKED) { if (S[b][e] <= -1e30f) p = 0.f; }
              S[b][e] = p;
              psum += p;
            }
        }
        if constexpr (!PIPE) l_run += psum;
        else l_part = psum;
      }

      // ---- O^T += V^T . P^T   (P^T taken from the S accumulator, permuted-k order); V^T fragments two MFMAs ahead
      {
        bf16x8 pf[2][2];
#pragma unroll
        for (int b = 0; b < (PIPE ? 1 : 2); ++b)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[b][s2][j] = f2bf(S[b][8 * s2 + j]);
        constexpr int NPV = 4 * DBLK;                        // MFMAs of this product: (b, s, d)
        auto vread = [&](int i) {
          const int bs = i / DBLK, d = i - bs * DBLK;        // bs = 2b + s
          const char* p0 = sV + v_lb[0] + 2048 * (2 * bs) + 512 * d;
          const char* p1 = sV + v_lb[1] + 2048 * (2 * bs + 1) + 512 * d;
#ifdef EXP_NO_VREAD
          (void)p0; (void)p1;
          return qf[i % KSTEPS];
#else
          union { struct { s16x4 a, b; } s; bf16x8 v; } uu;
          uu.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
          uu.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
          return uu.v;
#endif
        };
#ifdef EXP_VW
        constexpr int VW = EXP_VW;
#else
        constexpr int VW = 3;                                // V^T fragment window: VW - 1 MFMAs ahead
#endif
        bf16x8 vf[VW];
#pragma unroll
        for (int i = 0; i < VW - 1; ++i) vf[i] = vread(i);
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
          if (i + VW - 1 < NPV) vf[(i + VW - 1) % VW] = vread(i + VW - 1);
          const int bs = i / DBLK, d = i - bs * DBLK;
          if constexpr (PIPE) {
            if (i == NPV / 2) {                              // the b = 1 half is complete: pack it for MFMAs 8..15
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[1][s2][j] = f2bf(S[1][8 * s2 + j]);
            }
          }
          O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % VW], pf[bs >> 1][bs & 1], O[d], 0, 0, 0);
          if constexpr (PIPE) {
            if (i < NPV / 2) {                               // two scores of the b = 1 half behind each of the first 8 MFMAs
              const float mc2 = m_run * c;
#pragma unroll
              for (int e = 2 * i; e < 2 * i + 2; ++e) {
                float p = G2V_EXP2(fmaf(S[1][e], c, -mc2));
                if constexpr (MASKED) { if (S[1][e] <= -1e30f) p = 0.f; }
                S[1][e] = p;
                l_part += p;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr (PIPE) l_run += l_part;
      }

      wait_tile(false);                                    // this wave's DMA pieces (tile kt+1; PIPE: kt+2) have landed ...
#ifndef EXP_NO_BARRIER
      __builtin_amdgcn_s_barrier();                        // ... and so have everyone's; all reads of tile kt are done
#endif
    };

    // tiles [kt0, full_end) need no mask: kb + 64 <= k_len and kb + 63 <= first query's last allowed key
    int full_end;
    {
      const long lim = (long)(T.q0 - T.q_win0) + T.causal_shift;
      const long f2 = lim >= KV_TILE - 1 ? (lim - (KV_TILE - 1)) / KV_TILE + 1 : 0;
      full_end = (int)min((long)(T.k_len / KV_TILE), f2);
    }
    // two steps per trip so that the ping-pong roles of S_a / S_b are static; an odd tail copies S_b back once
    int kt = kt0;
    {
      const int end = min(kt1, full_end);
      for (; kt + 1 < end; kt += 2) {
        tile_step(kt, std::false_type{}, S_a, S_b);
        tile_step(kt + 1, std::false_type{}, S_b, S_a);
      }
      if (kt < end) {
        tile_step(kt, std::false_type{}, S_a, S_b);
        if constexpr (PIPE) { S_a[0] = S_b[0]; S_a[1] = S_b[1]; }
        ++kt;
      }
    }
    for (; kt + 1 < kt1; kt += 2) {
      tile_step(kt, std::true_type{}, S_a, S_b);
      tile_step(kt + 1, std::true_type{}, S_b, S_a);
    }
    if (kt < kt1) {
      tile_step(kt, std::true_type{}, S_a, S_b);
      ++kt;
    }

    // ---- finish the segment: lane = query row, registers = d
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const bool valid = 32 * w + r < T.q_rows;
    if (sg.slot < 0) {
      const float inv = 1.0f / l_tot;
      if (valid) {
        __bf16* op = a.o + (size_t)(T.q0 + 32 * w + r) * a.ldo + head * D;
#pragma unroll
        for (int d = 0; d < DBLK; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            int dc = 32 * d + 8 * g + 4 * hh;
            if (dc < D) {
              u32x2 wv = {pack_bf16x2(O[d][4 * g] * inv, O[d][4 * g + 1] * inv),
                          pack_bf16x2(O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv)};
              *reinterpret_cast<u32x2*>(op + dc) = wv;
            }
          }
      }
    } else {
      float* slot = a.ws + (size_t)sg.slot * SLOT_FLOATS;
      const int qq = 32 * w + r;
      if (hh == 0) { slot[qq] = m_run * c; slot[SLOT_ROWS + qq] = l_tot; }
      float* orow = slot + 2 * SLOT_ROWS + qq * 128;
#pragma unroll
      for (int d = 0; d < DBLK; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int dc = 32 * d + 8 * g + 4 * hh;
          if (dc < D) *reinterpret_cast<f32x4*>(orow + dc) = f32x4{O[d][4 * g], O[d][4 * g + 1], O[d][4 * g + 2], O[d][4 * g + 3]};
        }
    }
  }
}

// merge the partials of one split item: out = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M)
// grid (split items, 8-row slices): a thread owns 4 consecutive d of one row, D/4 consecutive threads cover a row, so
// every partial read is a 16-byte lane access on contiguous 4*D bytes per row and the bf16 result an 8-byte store.
// Few items are ever split (C3: 4 items in 32 pieces each), so the grid must not be sized by the item count alone: one thread
// per (row, 16-byte chunk), 8 rows per block, i.e. 32 blocks per 256-row item, and the loads of the slot loop are issued four
// slots at a time (the one-slot-per-iteration form was a chain of 2 x 32 dependent round trips on 32 blocks: 48.8 us per launch
// at C3, 7 % of the attention it finished).  The slots are still accumulated in ascending order: same bits.
constexpr int COMB_ROWS = 8;
template <int D>
__global__ __launch_bounds__(256) void flash_combine_kernel(FlashArgs a) {
  constexpr int CPR = D / 4;                               // float4 chunks per row
  const int desc = a.comb[4 * blockIdx.x], head = a.comb[4 * blockIdx.x + 1], s_lo = a.comb[4 * blockIdx.x + 2];
  const int s_hi = s_lo + a.comb[4 * blockIdx.x + 3];
  const g2v_attn_tile T = a.tiles[desc];
  const int row0 = blockIdx.y * COMB_ROWS;
  if (row0 >= T.q_rows) return;
  for (int idx = threadIdx.x; idx < COMB_ROWS * CPR; idx += 256) {
    const int q = row0 + idx / CPR, c = idx % CPR;
    if (q >= T.q_rows) break;
    const float* base = a.ws + (size_t)s_lo * SLOT_FLOATS;
    const int ns = s_hi - s_lo;
    float M = -INFINITY;
    for (int s0 = 0; s0 < ns; s0 += 8) {
      float mv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) mv[u] = base[(size_t)min(s0 + u, ns - 1) * SLOT_FLOATS + q];
#pragma unroll
      for (int u = 0; u < 8; ++u) M = fmaxf(M, mv[u]);
    }
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ns; s0 += 4) {
      float mv[4], lv[4];
      f32x4 ov[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* slot = base + (size_t)min(s0 + u, ns - 1) * SLOT_FLOATS;
        mv[u] = slot[q];
        lv[u] = slot[SLOT_ROWS + q];
        ov[u] = *reinterpret_cast<const f32x4*>(slot + 2 * SLOT_ROWS + q * 128 + 4 * c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (s0 + u < ns) {
          const float wgt = __builtin_amdgcn_exp2f(mv[u] - M);
          L = fmaf(lv[u], wgt, L);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(ov[u][e], wgt, acc[e]);
        }
      }
    }
    const float inv = 1.0f / L;
    __bf16* op = a.o + (size_t)(T.q0 + q) * a.ldo + head * D + 4 * c;
    *reinterpret_cast<u32x2*>(op) = u32x2{pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv)};
  }
}

template <int D>
int launch_flash(const FlashArgs& a, int n_comb, int waves, hipStream_t s) {
  if (a.n_blocks > 0) {
    if (waves == 8) hipLaunchKernelGGL((flash_fwd_kernel<D, 8>), dim3(a.n_blocks), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((flash_fwd_kernel<D, 4>), dim3(a.n_blocks), dim3(256), 0, s, a);
    G2V_CHECK_LAUNCH();
  }
  if (n_comb > 0) {
    hipLaunchKernelGGL(flash_combine_kernel<D>, dim3(n_comb, (32 * waves + COMB_ROWS - 1) / COMB_ROWS), dim3(256), 0, s, a);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

}  // namespace

extern "C" int64_t g2v_flash_attn_workspace(int n_slots) { return (int64_t)(n_slots > 0 ? n_slots : 0) * SLOT_FLOATS * 4; }

extern "C" int g2v_flash_attn(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                              const g2v_attn_tile* tiles, int n_tiles, int Hq, int Hkv, int D, float scale,
                              const g2v_attn_seg* segs, const int32_t* seg_ptr, int n_blocks, const int32_t* comb, int n_comb,
                              int tile_rows, void* workspace, void* stream) {
  if (!q || !k || !v || !o || !tiles || n_tiles < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || n_blocks < 0 || n_comb < 0) return G2V_ERR_ARG;
  if ((n_blocks > 0 && (!segs || !seg_ptr)) || (n_comb > 0 && (!comb || !workspace))) return G2V_ERR_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return G2V_ERR_ARG;
  if (tile_rows != 128 && tile_rows != 256) return G2V_ERR_ARG;
  if (n_tiles == 0 || (n_blocks == 0 && n_comb == 0)) return G2V_OK;
  FlashArgs a{(const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (__bf16*)o, tiles, segs, seg_ptr, comb,
              (float*)workspace, ldq, ldk, ldv, ldo, n_tiles, Hq, Hkv, n_blocks, scale * 1.4426950408889634f};
  const int waves = tile_rows / 32;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 16: return launch_flash<16>(a, n_comb, waves, s);
    case 64: return launch_flash<64>(a, n_comb, waves, s);
    case 80: return launch_flash<80>(a, n_comb, waves, s);
    case 96: return launch_flash<96>(a, n_comb, waves, s);
    case 128: return launch_flash<128>(a, n_comb, waves, s);
    default: return G2V_ERR_ARG;
  }
}

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
#include <utility>
#include "common.h"
#include "lds_dma.h"
#include "g2vlm_hip.h"

namespace {

// timing experiments only (tools/attn_variants.sh, tools/attn64_ablate.sh build the library once per macro; results are wrong
// with EXP_NO_* / EXP64_NO*; EXP64_ADDSUM = row sums by v_add_f32 instead of the matrix pipe, a correct A/B arm)
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
#ifdef EXP_STAMPS
  unsigned long long* dbg;   // diagnostic build only (tools/attn_stamps.py): s_memtime stamps of a few workgroups' waves 0 and 4
  unsigned long long* clk;
  int dbg_pos;               // flash_fwd64_kernel: which of the slice-end positions takes the movable stamp (one per launch: least perturbation)
#endif
};

#ifdef EXP_STAMPS
constexpr int STAMP_TILES = 24, STAMP_PTS = 8;
unsigned long long* g_attn_dbg = nullptr;
int g_attn_dbg_pos = -1;
unsigned long long* g_attn_clk = nullptr;     // per workgroup: {shader cycles, 100 MHz ticks} of wave 0 from kernel entry to exit
#define G2V_CLK_BEGIN() const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime()
#define G2V_CLK_END(lbv, wv) do { if (a.clk && (wv) == 0 && (threadIdx.x & 63) == 0) { a.clk[2 * (lbv)] = __builtin_amdgcn_s_memtime() - clk_c0; \
    a.clk[2 * (lbv) + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0; } } while (0)
#ifdef EXP_PSTAMPS
#define G2V_PSTAMP(idx) do { if (stamping && a.dbg_pos == (idx)) st_[1] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G2V_PSTAMP(idx) do { } while (0)
#endif
#define G2V_STAMP(i) do { if (stamping) st_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G2V_STAMP(i) do { } while (0)
#define G2V_PSTAMP(idx) do { } while (0)
#define G2V_CLK_BEGIN() do { } while (0)
#define G2V_CLK_END(lbv, wv) do { } while (0)
#endif

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
  G2V_CLK_BEGIN();

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
#ifdef EXP_OLD_DMA_BUILTIN
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kbase + (size_t)kr * a.ldk + p_col[i]), (lds_ptr_t)(sk + p_dst[i]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vbase + (size_t)kr * a.ldv + p_col[i]), (lds_ptr_t)(sk + TILE_B + p_dst[i]), 16, 0, 0);
#else
        // asm statements: hipcc would otherwise park an s_waitcnt vmcnt(0) in front of the first LDS read after these (see dma16_*)
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr(sk) + p_dst[i]);
        dma16_vaddr(reinterpret_cast<const char*>(kbase + (size_t)kr * a.ldk + p_col[i]), la);
        dma16_vaddr(reinterpret_cast<const char*>(vbase + (size_t)kr * a.ldv + p_col[i]), la + TILE_B);
#endif
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
#ifndef EXP_NO_DEAD_WAVE_SKIP
    // A wave without a query row in this item (the last, partly filled item of a window: 94 of 256 rows in a DINO window,
    // 89 in a decoder window) keeps its staging duty - its DMA pieces, waits and barriers, exactly those of the tile loop
    // below - and nothing else: its SIMD then belongs to the wave that shares it, and a SIMD issues its waves' instructions
    // one after the other (the item's tiles take about half the time).  The combine pass never reads rows past q_rows.
    if (32 * wu >= T.q_rows) {
      stage(kt0, 0);
      if (SLOTS == 3 && kt0 + 1 < kt1) stage(kt0 + 1, 1);
      wait_tile(false);
      __builtin_amdgcn_s_barrier();
      for (int kt = kt0; kt < kt1; ++kt) {
        if (kt + SLOTS - 1 < kt1) stage(kt + SLOTS - 1, (kt - kt0 + SLOTS - 1) % SLOTS);
        wait_tile(false);
        __builtin_amdgcn_s_barrier();
      }
      continue;
    }
#endif
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
#ifdef EXP_STAMPS
      const int st_blk = lb == 0 ? 0 : lb == 97 ? 1 : lb == 200 ? 2 : -1;
      const bool stamping = a.dbg && st_blk >= 0 && (wu == 0 || wu == 4) && si == a.seg_ptr[lb] && kt - kt0 >= 8 && kt - kt0 < 8 + STAMP_TILES;
      unsigned long long st_[STAMP_PTS] = {0, 0, 0, 0, 0, 0, 0, 0};
      G2V_STAMP(0);
#endif
      const char* sK = smem + buf * 2 * TILE_B;
      const char* sV = sK + TILE_B;
      // the slot being refilled was last read one barrier ago (tile kt-1)
      if (kt + SLOTS - 1 < kt1) stage(kt + SLOTS - 1, (kt - kt0 + SLOTS - 1) % SLOTS);

      G2V_STAMP(1);
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
        G2V_STAMP(2);
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
        G2V_STAMP(3);
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
      G2V_STAMP(4);

      wait_tile(false);                                    // this wave's DMA pieces (tile kt+1; PIPE: kt+2) have landed ...
      G2V_STAMP(5);
#ifndef EXP_NO_BARRIER
      __builtin_amdgcn_s_barrier();                        // ... and so have everyone's; all reads of tile kt are done
#endif
#ifdef EXP_STAMPS
      G2V_STAMP(6);
      if (stamping && lane == 0) {
        unsigned long long* d = a.dbg + ((size_t)(st_blk * 2 + (wu >> 2)) * STAMP_TILES + (kt - kt0 - 8)) * STAMP_PTS;
#pragma unroll
        for (int i = 0; i < STAMP_PTS; ++i) d[i] = st_[i];
      }
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
  G2V_CLK_END(lb, w);
}

// ======================================================================================================================
// flash_fwd64_kernel: head dim 128, 256-row items - the MoT prefill form.  4 waves x 64 query rows, ONE wave per SIMD.
//
// Why (round 3, tools/attn_stamps.py + the ISA of the 8 x 32 form above): a SIMD issues the instructions of its waves one
// after the other - the stamped tile time of the 8-wave form is the SUM of its two waves' issue costs (335 instructions per
// wave and tile for 32 MFMAs = 10.4 per MFMA, ~1 900 cycles per wave, two waves back to back = 3 940 against 2 048 of MFMA
// time: the measured 52 % MFMA utilisation).  What helps is fewer instructions per MFMA, and that is what 64 rows per wave
// give: every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16 pair) feeds TWO MFMAs, and the per-tile
// overhead (DMA address arithmetic, waits, barrier, loop control) is paid once per 64 MFMAs instead of once per 32.
//
// Registers (guide, 'Fused attention prefill', 4-wave form): a wave owns the whole 512-register file of its SIMD.  The O^T
// accumulators (2 q-blocks x 4 d-blocks x 16 = 128) and the Q^T fragments (64) live in ACCUMULATION registers for the whole
// item - MFMA takes A / B / C / D operands from them directly - and the scores, the exponentials and the K / V fragment
// windows in the architectural VGPRs.  hipcc cannot be made to keep a value in an AGPR (given "a"-constrained operands it
// duplicated the 128 O registers around the rescale and spilled Q to scratch), so the accumulation registers are OWNED here:
// a[ACC_O .. ACC_O + 127] = O, a[ACC_Q .. ACC_Q + 63] = Q, named literally in every asm statement that touches them and
// declared once as clobbers so that the kernel descriptor allocates them; a[0 .. 63] stay the compiler's.  The kernel must
// therefore build with NO spill and NO compiler-generated v_accvgpr_* (guide 5.7 item 4; checked by tests/test_build_cpu.py).
// hipcc does not pad hazards around an asm statement either, so producer / consumer pairs that cross one are spaced by hand
// (H1..H4 below).
//
// Softmax reference = a power of two, chosen once.  Rescaling O in AGPRs costs 3 x 128 instructions, so it must not happen in
// steady state.  The reference m_ref of a row is an INTEGER in the log2 domain, set from the row maximum of the segment's
// first tile (ceil of max s * scale * log2 e) and raised only if a later row maximum exceeds it by more than RESCALE_THR = 64:
// p = 2^(s c - m_ref) <= 2^64, far inside the range of bf16 / fp32, and floating point is scale invariant, so nothing is lost
// by p > 1.  Because the reference moves by integers, alpha = 2^(m_old - m_new) is exact and so are O alpha, l alpha and
// 2^(x - n) = 2^x 2^-n: results are bit-identical for ANY threshold.  Against the exact-maximum form the only difference is
// that a row's largest p is not exactly 1 (its bf16 rounding error is that of every other p).  The rescale path exists
// (adversarial score growth, rows whose first tile is fully masked) and is cold.
//
// Pipeline per KV tile t (S = scores of t, complete on entry; Sn = scores of t + 1):
//   phase 1: 8 slices x {2 K-fragment reads, 4 MFMAs Sn += K(t+1) . Q^T, exp / sum / pack of 4 scores of S's first key half}
//   phase 2: 16 slices x {2 V^T reads, 2 MFMAs O^T += V^T(t) . P^T}; slices 0-7 also exponentiate the second key half,
//            slices 8-15 carry the row maxima of Sn and one LDS-DMA piece each (K of tile t + 3, V of tile t + 2: the K ring
//            runs two tiles ahead of its reads, the V ring one, so every piece has a whole tile time to land and the wait at
//            the tile's end is a COUNTED vmcnt(8) that leaves this tile's own pieces in flight).
//   One barrier per tile.
//
// Hand-spaced hazards (ISA 4.5; nothing here is interlocked):
//   H1  MFMA writes S / Sn (VGPR) -> VALU reads it: the first reader is >= 16 MFMAs later in program order (32 wait states
//       after the prologue's un-pipelined scores).
//   H2  VALU writes a packed P register -> MFMA reads it as B: the first P.V MFMA of every 16-key group (the only ones whose
//       P operand can be fresh) opens with s_nop 1 (the guide's figure) INSIDE its asm string - an untied nop statement does not hold: hipcc sinks the
//       pure pack instructions below it, right in front of the MFMA (measured: wrong values in exactly that MFMA's d-block).
//   Pure VALU code is also free to move across asm statements and sched_barriers as long as its operands allow; what must stay
//   inside a slice (the fillers: exp / sum / pack) or behind a point (the row maxima of Sn) is pinned by an EMPTY asm statement
//   that names the values "+v" (guide 5.7 item 3).
//   H3  MFMA writes O (AGPR) -> v_accvgpr_read (rescale, epilogue): 32 wait states first.
//   H4  v_accvgpr_write (zero-init, Q, rescale write-back) -> MFMA reads it: 32 wait states after.
constexpr int RESCALE_THR = 64;
constexpr int ACC_Q = 64, ACC_O = 128;          // first accumulation register of Q^T ([q-block][k-step] x 4) and of O^T ([q-block][d-block] x 16)
// Row sums on the matrix pipe.  The kernel is bound by vector ISSUE slots, not by the MFMA pipe (which idles ~25 % of a tile), so
// the 64 v_add_f32 per tile of l = sum_k p are traded for 8 MFMAs: l^T[any row][query] += ONES[32 x 16] . P^T[16 keys x 32 queries]
// with a constant all-ones A operand (a[ACC_ONE..+3], bf16 1.0) accumulates in a[ACC_L + 16 qb ..] the sum of the SAME bf16-rounded
// p that P.V multiplies - every row of the 32 x 32 result, i.e. every register of both lane halves, holds the query's full sum.
constexpr int ACC_ONE = 28, ACC_L = 32;

#define G2V_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
__device__ __forceinline__ void acc_declare() {     // a28 .. a255 are used by this kernel (descriptor allocation)
  asm volatile("" ::: "a28", "a29", G2V_A8(3), G2V_A8(4), G2V_A8(5), "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", G2V_A8(7), G2V_A8(8), G2V_A8(9), G2V_A8(10), G2V_A8(11), G2V_A8(12),
               G2V_A8(13), G2V_A8(14), G2V_A8(15), G2V_A8(16), G2V_A8(17), G2V_A8(18), G2V_A8(19), G2V_A8(20), G2V_A8(21), G2V_A8(22),
               G2V_A8(23), G2V_A8(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
template <int R> __device__ __forceinline__ void acc_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" :: "v"(v), "i"(R)); }
template <int R> __device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "i"(R)); }
template <int R> __device__ __forceinline__ float acc_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R)); return v; }
template <int R> __device__ __forceinline__ void acc_scale(float alpha) {   // a[R] *= alpha (cold path: every step padded)
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a%c2\n\ts_nop 1\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a%c2, %0" : "=&v"(t) : "v"(alpha), "i"(R));
}
template <int R0, int... I> __device__ __forceinline__ void acc_zero_seq(std::integer_sequence<int, I...>) { (acc_zero<R0 + I>(), ...); }
template <int R0, int... I> __device__ __forceinline__ void acc_scale_seq(float alpha, std::integer_sequence<int, I...>) { (acc_scale<R0 + I>(alpha), ...); }
template <int R0, int... I> __device__ __forceinline__ void acc_read_seq(float (&out)[sizeof...(I)], std::integer_sequence<int, I...>) { ((out[I] = acc_read<R0 + I>()), ...); }
template <class F, int... G> __device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, G...>) { (f(std::integral_constant<int, G>{}), ...); }
__device__ __forceinline__ void wait32() { asm volatile("s_nop 15\n\ts_nop 15"); }

// S (VGPR) = / += K fragment (VGPR) . Q^T fragment (AGPR a[QR .. QR+3])
template <int QR> __device__ __forceinline__ void mfma_s_init(f32x16& S, const bf16x8& k) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(S) : "v"(k), "i"(QR), "i"(QR + 3));
}
template <int QR> __device__ __forceinline__ void mfma_s_acc(f32x16& S, const bf16x8& k) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(S) : "v"(k), "i"(QR), "i"(QR + 3));
}
// O (AGPR a[OR .. OR+15]) += V^T fragment (VGPR) . P^T fragment (VGPR)
template <int OR> __device__ __forceinline__ void mfma_o_acc(const bf16x8& v, const bf16x8& p) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(p), "i"(OR), "i"(OR + 15));
}
template <int LR> __device__ __forceinline__ void mfma_l_acc(const bf16x8& p) {     // l (AGPR a[LR .. LR+15]) += ones . P^T
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" :: "v"(p), "i"(LR), "i"(LR + 15), "i"(ACC_ONE), "i"(ACC_ONE + 3));
}
template <int OR> __device__ __forceinline__ void mfma_o_acc_fresh_p(const bf16x8& v, const bf16x8& p) {   // H2
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(p), "i"(OR), "i"(OR + 15));
}
__device__ __forceinline__ void pin(uint32_t& a) { asm volatile("" : "+v"(a)); }
// max(a, b, c) as ONE v_max3_f32: fmaxf on values that come out of asm statements makes hipcc canonicalise every input first
// (a v_max_f32 x, x per score: 54 extra instructions per tile)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// H5: keep an MFMA's A / B operand registers untouched while the MFMA is in flight.  hipcc takes an asm statement's inputs for
// read the moment it is issued and hands the registers to the very next instruction (a softmax temporary, an LDS read); the
// matrix pipe reads its operands over several cycles and nothing interlocks a vector WRITE behind that read: `v_mfma .., v[130:133],
// ..` followed by `v_exp_f32 v130, ..` gave NaNs and run-to-run flicker.  A use at the END of the gap (after the fillers, i.e.
// >= 32 cycles later, when the next MFMA is about to issue) keeps them allocated.
__device__ __forceinline__ void keep(const bf16x8& x) { asm volatile("" :: "v"(x)); }
__device__ __forceinline__ void keep(const bf16x8& x, const bf16x8& y) { asm volatile("" :: "v"(x), "v"(y)); }
__device__ __forceinline__ void pin(f32x16 (&S)[2][2]) { asm volatile("" : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[1][0]), "+v"(S[1][1])); }

__global__ __launch_bounds__(256, 1) void flash_fwd64_kernel(FlashArgs a) {
  constexpr int D = 128, KSTEPS = 8, DBLK = 4, NW = 4, SLOTS = 3, NPW = 4;
  __shared__ __attribute__((aligned(16))) char smem[SLOTS * 2 * TILE_B];       // K ring [3][16 KiB] | V ring [3][16 KiB]
  char* const smK = smem;
  char* const smV = smem + SLOTS * TILE_B;
  acc_declare();
  acc_write<ACC_ONE>(0x3F803F80u); acc_write<ACC_ONE + 1>(0x3F803F80u); acc_write<ACC_ONE + 2>(0x3F803F80u); acc_write<ACC_ONE + 3>(0x3F803F80u);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int lb = blockIdx.x;
  {
    int xcd = lb & 7, qn = a.n_blocks >> 3, rn = a.n_blocks & 7;
    lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (lb >> 3);
  }
  const int seg_end = a.seg_ptr[lb + 1];
  G2V_CLK_BEGIN();

  // ---- LDS-DMA pieces: 16 pieces of 1 KiB per operand tile, piece pi = wu + 4 i of wave wu (8 rows x 64 columns of the
  // layout-(a) image): row / column of this lane inside piece pi; byte offset from the tile's first row (ldk == ldv here)
  auto piece_row = [&](int i) { return 8 * ((wu + NW * i) >> 1) + ((lane & 31) >> 2); };
  auto piece_col = [&](int i) {
    const int row = piece_row(i);
    return 8 * (4 * (2 * ((wu + NW * i) & 1) + (lane >> 5)) + ((lane & 3) ^ ((row >> 2) & 3)));
  };
  auto piece_dst = [&](int i) { return 2048 * ((wu + NW * i) >> 1) + 1024 * ((wu + NW * i) & 1); };   // wave-uniform
  uint32_t koff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) koff[i] = ((uint32_t)piece_row(i) * (uint32_t)a.ldk + (uint32_t)piece_col(i)) * 2u;
  int k_lb[2];
  k_lb[0] = 2048 * (r >> 3) + 64 * (r & 7) + 16 * (hh ^ ((r >> 2) & 3));
  k_lb[1] = k_lb[0] ^ 32;
  int v_lb[2];
  {
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int t_ch = 2 * ((lane >> 4) & 1) + (tp >> 1);
    v_lb[0] = 64 * (4 * hh + tq) + 16 * (t_ch ^ hh) + 8 * (tp & 1);
    v_lb[1] = v_lb[0] ^ 32;
  }
  const float c = a.scale_log2;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

  for (int si = a.seg_ptr[lb]; si < seg_end; ++si) {
    // the segment and its tile descriptor are the same for every lane: scalar registers (asm "s" operands need provable uniformity)
    auto U = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    g2v_attn_seg sg = a.segs[si];
    sg.slot = U(sg.slot);
    const int head = U(sg.head), kt0 = U(sg.kt0), kt1 = U(sg.kt1);
    g2v_attn_tile T = a.tiles[U(sg.desc)];
    T.q0 = U(T.q0); T.q_rows = U(T.q_rows); T.k0 = U(T.k0); T.k_len = U(T.k_len); T.causal_shift = U(T.causal_shift); T.q_win0 = U(T.q_win0);
    const int kvh = head / (a.Hq / a.Hkv);
    const char* kbase = uniform_ptr(reinterpret_cast<const char*>(a.k + (size_t)T.k0 * a.ldk + kvh * D));
    const char* vbase = uniform_ptr(reinterpret_cast<const char*>(a.v + (size_t)T.k0 * a.ldv + kvh * D));
    const int n_full = T.k_len / KV_TILE;                                      // tiles whose 64 rows all exist
    const size_t tile_bytes = (size_t)KV_TILE * a.ldk * 2;

    // one operand tile into its ring slot, any tile: the last, partly filled one clamps its rows per lane (rows past the
    // window repeat the last row and are masked).  Prologue and segment tail only; the steady state issues its pieces inside
    // phase 2 (uniform 64-bit base + 32-bit lane offset: the saddr form, no 64-bit vector arithmetic)
    auto stage_any = [&](const char* base, char* ring, int kt) {
      char* dst = ring + (kt % SLOTS) * TILE_B;
      if (kt < n_full) {
        const char* tb = uniform_ptr(base + (size_t)kt * tile_bytes);
        const uint32_t da = __builtin_amdgcn_readfirstlane(lds_addr(dst));
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma16_saddr(tb, koff[i], da + piece_dst(i));
      } else {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const int kr = min(kt * KV_TILE + piece_row(i), T.k_len - 1);
          dma16_vaddr(base + ((size_t)kr * a.ldk + piece_col(i)) * 2, __builtin_amdgcn_readfirstlane(lds_addr(dst)) + piece_dst(i));
        }
      }
    };

    // ---- prologue: K(kt0), V(kt0), K(kt0+1), V(kt0+1), K(kt0+2) in flight while Q is fetched
    stage_any(kbase, smK, kt0);
    stage_any(vbase, smV, kt0);
    if (kt0 + 1 < kt1) { stage_any(kbase, smK, kt0 + 1); stage_any(vbase, smV, kt0 + 1); }
    if (kt0 + 2 < kt1) stage_any(kbase, smK, kt0 + 2);

    // Q^T fragments (B operand) of the wave's two 32-row q-blocks -> a[ACC_Q + 4 (8 qb + ks) ..]; O = 0
    {
      auto q_to_acc = [&](auto qb_t) {
        constexpr int qb = decltype(qb_t)::value;
        const int qi = min(64 * wu + 32 * qb + r, T.q_rows - 1);                // padded rows duplicate the last one
        const __bf16* qp = a.q + (size_t)(T.q0 + qi) * a.ldq + head * D + 8 * hh;
        u32x4 qv[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qv[ks] = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
        auto put = [&](auto ks_t) {
          constexpr int ks = decltype(ks_t)::value, R = ACC_Q + 4 * (8 * qb + ks);
          acc_write<R>(qv[ks][0]); acc_write<R + 1>(qv[ks][1]); acc_write<R + 2>(qv[ks][2]); acc_write<R + 3>(qv[ks][3]);
        };
        put(std::integral_constant<int, 0>{}); put(std::integral_constant<int, 1>{}); put(std::integral_constant<int, 2>{});
        put(std::integral_constant<int, 3>{}); put(std::integral_constant<int, 4>{}); put(std::integral_constant<int, 5>{});
        put(std::integral_constant<int, 6>{}); put(std::integral_constant<int, 7>{});
      };
      q_to_acc(std::integral_constant<int, 0>{});
      q_to_acc(std::integral_constant<int, 1>{});
      acc_zero_seq<ACC_O>(std::make_integer_sequence<int, 128>{});
      acc_zero_seq<ACC_L>(std::make_integer_sequence<int, 32>{});
      wait32();                                                                  // H4
    }
    const int kmax0 = (T.q0 - T.q_win0) + 64 * wu + r + T.causal_shift;         // q-block 0's last allowed key (window-local), may be huge
    float m_ref[2];                                                             // integer-valued, log2 domain
#ifdef EXP64_ADDSUM
    float l_run[2] = {0.f, 0.f};
#endif

    __builtin_amdgcn_s_waitcnt(0x0F70);                                          // vmcnt(0): prologue tiles
    __builtin_amdgcn_s_barrier();

    auto kread = [&](const char* sK, int ks, int b) {
      return *reinterpret_cast<const bf16x8*>(sK + k_lb[ks & 1] + 8192 * b + 512 * (ks >> 1));
    };
    auto vread = [&](const char* sV, int i) {
      const int bs = i / DBLK, d = i - bs * DBLK;                              // bs = 2b + s: 16-key group, d: 32-wide d block

      union { struct { s16x4 a, b; } s; bf16x8 v; } uu;
      uu.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + v_lb[0] + 2048 * (2 * bs) + 512 * d));
      uu.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + v_lb[1] + 2048 * (2 * bs + 1) + 512 * d));
      return uu.v;
    };
    // raw row maximum of one q-block's scores (this lane's 32 keys, then the other half of the row from lane ^ 32)
    auto row_max = [&](const f32x16 (&Sq)[2]) {
      float m0 = fmaxf(Sq[0][0], Sq[1][0]);
#pragma unroll
      for (int e = 1; e < 16; ++e) m0 = fmaxf(m0, fmaxf(Sq[0][e], Sq[1][e]));
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
      return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    auto apply_mask = [&](f32x16 (&Sx)[2][2], int kt) {
      const int kb = kt * KV_TILE;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = kb + 32 * b + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (key >= T.k_len || key > kmax0 + 32 * qb) Sx[qb][b][e] = -1e30f;
          }
    };
    // the 32 MFMAs Sn = K . Q^T of one tile.  Slice ks = {4 MFMAs of k-step ks | K fragment reads of k-step ks + 1, filler(ks)}:
    // hipcc waits for ALL outstanding LDS reads (lgkmcnt(0)) before an asm statement that takes a loaded register, so the reads
    // of the next k-step are issued right AFTER this k-step's MFMAs and have the whole filler (and the MFMAs' own 128 cycles)
    // to return before the wait in front of the next slice
    auto qk_tile = [&](const char* sK, f32x16 (&Sx)[2][2], auto&& filler) {
#ifdef EXP64_KW
      constexpr int KW = EXP64_KW;
#else
      constexpr int KW = 3;                                 // two k-steps ahead: the wait in front of a slice is a counted lgkmcnt(2)
#endif
      bf16x8 kfw[KW][2];
#pragma unroll
      for (int ks = 0; ks < KW - 1; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b) kfw[ks][b] = kread(sK, ks, b);
      __builtin_amdgcn_sched_barrier(0);
      auto kstep = [&](auto ks_t) {
        constexpr int ks = decltype(ks_t)::value;
        if constexpr (ks == 0) {
          mfma_s_init<ACC_Q + 4 * ks>(Sx[0][0], kfw[ks % KW][0]); mfma_s_init<ACC_Q + 4 * (8 + ks)>(Sx[1][0], kfw[ks % KW][0]);
          mfma_s_init<ACC_Q + 4 * ks>(Sx[0][1], kfw[ks % KW][1]); mfma_s_init<ACC_Q + 4 * (8 + ks)>(Sx[1][1], kfw[ks % KW][1]);
        } else {
          mfma_s_acc<ACC_Q + 4 * ks>(Sx[0][0], kfw[ks % KW][0]); mfma_s_acc<ACC_Q + 4 * (8 + ks)>(Sx[1][0], kfw[ks % KW][0]);
          mfma_s_acc<ACC_Q + 4 * ks>(Sx[0][1], kfw[ks % KW][1]); mfma_s_acc<ACC_Q + 4 * (8 + ks)>(Sx[1][1], kfw[ks % KW][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks + KW - 1 < KSTEPS) {
#pragma unroll
          for (int b = 0; b < 2; ++b) kfw[(ks + KW - 1) % KW][b] = kread(sK, ks + KW - 1, b);
        }
        filler(ks);
        keep(kfw[ks % KW][0], kfw[ks % KW][1]);
        __builtin_amdgcn_sched_barrier(0);
      };
      kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
      kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
      kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{});
    };

    // tiles [kt0, full_end) need no mask: all 64 keys exist and lie at or below the first query row's last allowed key
    int full_end;
    {
      const long lim = (long)(T.q0 - T.q_win0) + T.causal_shift;
      const long f2 = lim >= KV_TILE - 1 ? (lim - (KV_TILE - 1)) / KV_TILE + 1 : 0;
      full_end = (int)min((long)n_full, f2);
    }
    // iterations [kt0, dma_end) issue their LDS-DMA inside phase 2: K(kt+3) and V(kt+2) exist, are whole tiles of this segment
    const int dma_end = min(min(kt1, n_full) - 3, full_end);

    // ---- scores of the first tile (not pipelined), its mask, row maxima and the softmax reference
    f32x16 S_a[2][2], S_b[2][2];
    float rmx[2];
    qk_tile(smK + (kt0 % SLOTS) * TILE_B, S_a, [](int) {});
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S_a[0][0]), "+v"(S_a[0][1]), "+v"(S_a[1][0]), "+v"(S_a[1][1]));   // H1
    if (kt0 >= full_end) apply_mask(S_a, kt0);
    rmx[0] = row_max(S_a[0]);
    rmx[1] = row_max(S_a[1]);
    m_ref[0] = ceilf(rmx[0] * c);
    m_ref[1] = ceilf(rmx[1] * c);

    // One KV tile.  S: its scores (masked already, row maxima in rmx); Sn: receives tile kt + 1's scores; on exit rmx holds
    // tile kt + 1's row maxima (raw; the caller masks and recomputes them when that tile needs a mask).
    // MASKED: some score of S is -1e30 (its exponential is forced to 0).  DMA: steady state, see dma_end.
    //
    // The tile is ONE sequence of 72 MFMA gaps, every gap = {one MFMA, at most ~5 vector / LDS instructions} (measured with
    // tools/mfma_gap_bench.py, one wave per SIMD: a gap hides 4 v_fma or 2 v_exp; ONE dependent fma -> exp pair costs 35 cycles,
    // two 42, four 66 - bunching four pairs behind four MFMAs, the first form of this kernel, ran 60 cycles per MFMA):
    //   gaps  0..31  Sn += K(kt+1) . Q^T            k-step g / 4, key half (g % 4) / 2, q-block g % 2
    //   gaps 32..71  per 16-key group bs: 8 x O^T += V^T . P^T (d-block, q-block), then the 2 row-sum MFMAs
    //   gap g < 62   exp of score e = g + 2 (its s c - m_ref was formed one gap earlier: no dependent pair inside a gap), fma for
    //                score e + 1, a bf16 pack on odd e; score e = (key half e / 32, pair (e % 32) / 4, q-block (e % 4) / 2,
    //                element e % 2), i.e. the packed operands complete in the order the P.V MFMAs consume them
    //   gaps 0, 4, .. 20   K fragments of k-step g / 4 + 2;  gaps 24, 28 and each V^T fragment's first gap: V^T fragments VW - 1 ahead
    //   gaps 48..63  row maxima of Sn (complete since gap 31: H1);  gaps 64..71  one LDS-DMA piece each (K(kt+3) x 4, V(kt+2) x 4)
    auto tile_step = [&](int kt, auto masked_t, auto dma_t, f32x16 (&S)[2][2], f32x16 (&Sn)[2][2]) {
      constexpr bool MASKED = decltype(masked_t)::value, DMA = decltype(dma_t)::value;
      const char* sKn = smK + ((kt + 1) % SLOTS) * TILE_B;
      const char* sV = smV + (kt % SLOTS) * TILE_B;
      if constexpr (!DMA) {                                 // segment tail / masked tiles: the ring is fed ahead of the gaps
        if (kt + 3 < kt1) stage_any(kbase, smK, kt + 3);
        if (kt + 2 < kt1) stage_any(vbase, smV, kt + 2);
      }

      // ---- softmax reference: raised only when a row maximum has outgrown it by 2^RESCALE_THR (cold)
      {
        const float x0 = rmx[0] * c, x1 = rmx[1] * c;
        if (__any(x0 > m_ref[0] + (float)RESCALE_THR || x1 > m_ref[1] + (float)RESCALE_THR)) {
          const float mn0 = fmaxf(m_ref[0], ceilf(x0)), mn1 = fmaxf(m_ref[1], ceilf(x1));
          const float al0 = __builtin_amdgcn_exp2f(m_ref[0] - mn0), al1 = __builtin_amdgcn_exp2f(m_ref[1] - mn1);   // exact powers of two (or 0)
          wait32();                                          // H3
          acc_scale_seq<ACC_O>(al0, std::make_integer_sequence<int, 64>{});
          acc_scale_seq<ACC_O + 64>(al1, std::make_integer_sequence<int, 64>{});
          acc_scale_seq<ACC_L>(al0, std::make_integer_sequence<int, 16>{});
          acc_scale_seq<ACC_L + 16>(al1, std::make_integer_sequence<int, 16>{});
          wait32();                                          // H4
          m_ref[0] = mn0; m_ref[1] = mn1;
        }
      }
      const float mr0 = m_ref[0], mr1 = m_ref[1];
      constexpr int KW = 3, VW = 3, NG = 72;
      bf16x8 kfw[KW][2], vfw[VW];
      uint32_t pk[2][2][8];                                 // [q-block][key half b][pair k] = bf16 (p(2k), p(2k+1)); [4 s2, 4 s2 + 4) = one B operand
      float x_cur, p_even = 0.f;                            // s c - m_ref of the score the NEXT gap exponentiates; the pair's first p
#ifdef EXP64_ADDSUM
      float psum[2] = {0.f, 0.f};
#endif
      float nm0 = -3e38f, nm1 = -3e38f, nm0b = -3e38f, nm1b = -3e38f;   // running maxima of Sn (this lane's keys; one chain per key half)
      const char* kdma = uniform_ptr(kbase + (size_t)(kt + 3) * tile_bytes);          // wave-uniform bases of the two tiles fed below
      const char* vdma = uniform_ptr(vbase + (size_t)(kt + 2) * tile_bytes);
      const uint32_t kdst = __builtin_amdgcn_readfirstlane(lds_addr(smK + ((kt + 3) % SLOTS) * TILE_B));
      const uint32_t vdst = __builtin_amdgcn_readfirstlane(lds_addr(smV + ((kt + 2) % SLOTS) * TILE_B));
      // score g of S (see above) and its reference
      auto score = [&](auto g_t) -> float {
        constexpr int g = decltype(g_t)::value;
        return S[(g % 4) / 2][g / 32][2 * ((g % 32) / 4) + (g % 2)];
      };
      // score e: exp (its s c - m_ref was formed one step earlier: no dependent fma -> exp pair back to back), the fma of score
      // e + 1, a bf16 pack on odd e
      auto soft = [&](auto e_t) {
        constexpr int e = decltype(e_t)::value, b = e / 32, k = (e % 32) / 4, qb = (e % 4) / 2;
        float p = G2V_EXP2(x_cur);
        if constexpr (MASKED) { if (score(e_t) <= -1e30f) p = 0.f; }
        if constexpr (e + 1 < 64) x_cur = fmaf(score(std::integral_constant<int, (e + 1 < 64 ? e + 1 : 0)>{}), c, ((e + 1) % 4) / 2 ? -mr1 : -mr0);
        if constexpr (e % 2 == 0) p_even = p;
        else {
          pk[qb][b][k] = pack_bf16x2(p_even, p); pin(pk[qb][b][k]);
#ifdef EXP64_ADDSUM
          psum[qb] += p_even + p;
#endif
        }
      };
#pragma unroll
      for (int b = 0; b < 2; ++b) { kfw[0][b] = kread(sKn, 0, b); kfw[1][b] = kread(sKn, 1, b); }
      x_cur = fmaf(score(std::integral_constant<int, 0>{}), c, -mr0);
#ifndef EXP64_NOVALU
      soft(std::integral_constant<int, 0>{});
      soft(std::integral_constant<int, 1>{});
#endif
      __builtin_amdgcn_sched_barrier(0);

      auto gap = [&](auto g_t) {
        constexpr int g = decltype(g_t)::value;
        // ---- the MFMA
        if constexpr (g < 32) {
          constexpr int ks = g / 4, b = (g % 4) / 2, qb = g % 2;
          if constexpr (ks == 0) mfma_s_init<ACC_Q + 4 * (8 * qb + ks)>(Sn[qb][b], kfw[ks % KW][b]);
          else mfma_s_acc<ACC_Q + 4 * (8 * qb + ks)>(Sn[qb][b], kfw[ks % KW][b]);
        } else {
          constexpr int g2 = g - 32, bs = g2 / 10, rr = g2 % 10, b = bs >> 1, s2 = bs & 1;
          constexpr int qb = rr < 8 ? rr % 2 : rr - 8;
          bf16x8 pb;
          __builtin_memcpy(&pb, &pk[qb][b][4 * s2], 16);
          if constexpr (rr < 8) {
            constexpr int d = rr / 2;
            if constexpr (d == 0) mfma_o_acc_fresh_p<ACC_O + 64 * qb + 16 * d>(vfw[(4 * bs + d) % VW], pb);      // H2
            else mfma_o_acc<ACC_O + 64 * qb + 16 * d>(vfw[(4 * bs + d) % VW], pb);
          } else {
#ifndef EXP64_ADDSUM
            mfma_l_acc<ACC_L + 16 * qb>(pb);
#endif
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LDS fragment reads (issued right after an MFMA: hipcc's wait in front of the consuming asm is counted)
        if constexpr (g < 32 && g % 4 == 0 && g / 4 + KW - 1 < KSTEPS) {
#pragma unroll
          for (int b = 0; b < 2; ++b) kfw[(g / 4 + KW - 1) % KW][b] = kread(sKn, g / 4 + KW - 1, b);
        }
        if constexpr (g == 24) vfw[0] = vread(sV, 0);
        if constexpr (g == 28) vfw[1] = vread(sV, 1);
        if constexpr (g >= 32 && (g - 32) % 10 < 8 && (g - 32) % 2 == 0) {
          constexpr int i = 4 * ((g - 32) / 10) + ((g - 32) % 10) / 2;          // fragment used by this gap and the next
          if constexpr (i + VW - 1 < 4 * DBLK) vfw[(i + VW - 1) % VW] = vread(sV, i + VW - 1);
        }
        // ---- softmax: score g + 2 (two ahead of the gap index, so that the last packed operand exists before gap 62, where the
        // last 16-key group's MFMAs begin)
#ifndef EXP64_NOVALU
        if constexpr (g + 2 < 64) soft(std::integral_constant<int, g + 2 < 64 ? g + 2 : 0>{});
        // ---- row maxima of the next tile: 4 scores of one q-block per gap
        if constexpr (g == 48) pin(Sn);                     // H1: the reads below stay behind 16 P.V MFMAs
        if constexpr (g >= 48 && g < 64) {
          constexpr int j = (g - 48) / 2, qb = (g - 48) % 2;
          // two independent chains per q-block (one per key half): a dependent max3 pair back to back stalls, and hipcc pads it
          if constexpr (qb == 0) { nm0 = max3(nm0, Sn[0][0][2 * j], Sn[0][0][2 * j + 1]); nm0b = max3(nm0b, Sn[0][1][2 * j], Sn[0][1][2 * j + 1]); }
          else { nm1 = max3(nm1, Sn[1][0][2 * j], Sn[1][0][2 * j + 1]); nm1b = max3(nm1b, Sn[1][1][2 * j], Sn[1][1][2 * j + 1]); }
        }
#endif
        // ---- H5: this gap's MFMA operands stay allocated until here
        if constexpr (g < 32) keep(kfw[(g / 4) % KW][(g % 4) / 2]);
        else {
          constexpr int g2 = g - 32, bs = g2 / 10, rr = g2 % 10;
          bf16x8 pb;
          __builtin_memcpy(&pb, &pk[rr < 8 ? rr % 2 : rr - 8][bs >> 1][4 * (bs & 1)], 16);
          if constexpr (rr < 8) keep(vfw[(4 * bs + rr / 2) % VW], pb); else keep(pb);
        }
        // ---- one LDS-DMA piece per gap at the tile's end
#ifndef EXP64_NODMA
        if constexpr (DMA && g >= 64) {                   // (spread over gaps 33, 37, .. 61 instead: 1.5 % slower)
          constexpr int j = g - 64;
          if constexpr (j < 4) dma16_saddr_settled(kdma, koff[j], kdst + piece_dst(j));
          else dma16_saddr_settled(vdma, koff[j - 4], vdst + piece_dst(j - 4));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      };
      static_for(gap, std::make_integer_sequence<int, NG>{});

#ifdef EXP64_ADDSUM
      l_run[0] += psum[0]; l_run[1] += psum[1];
#endif
      {  // finish the next tile's row maxima: the other half of each row lives in lane ^ 32
        nm0 = fmaxf(nm0, nm0b); nm1 = fmaxf(nm1, nm1b);
        auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(nm0), __float_as_uint(nm0), false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(nm1), __float_as_uint(nm1), false, false);
        rmx[0] = fmaxf(__uint_as_float(s0[0]), __uint_as_float(s0[1]));
        rmx[1] = fmaxf(__uint_as_float(s1[0]), __uint_as_float(s1[1]));
      }
      // everything issued before this tile has landed: K(kt+2) and V(kt+1); in steady state this tile's own 8 pieces stay
      // in flight (counted wait), in the tail everything is drained
#ifndef EXP64_NODMA
      if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8)
      else
#endif
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
#ifndef EXP_NO_BARRIER
      __builtin_amdgcn_s_barrier();                         // all reads of K(kt+1) and V(kt) are done; the landed tiles are everyone's
#endif
    };

    // two steps per trip so that the ping-pong roles of S_a / S_b are static
    int kt = kt0;
    auto step = [&](f32x16 (&S)[2][2], f32x16 (&Sn)[2][2]) {
      if (kt < dma_end) tile_step(kt, std::false_type{}, std::true_type{}, S, Sn);
      else if (kt < full_end) tile_step(kt, std::false_type{}, std::false_type{}, S, Sn);
      else tile_step(kt, std::true_type{}, std::false_type{}, S, Sn);
      ++kt;
      if (kt < kt1 && kt >= full_end) {                     // the tile just scored needs a mask: apply it, redo its row maxima
        apply_mask(Sn, kt);
        rmx[0] = row_max(Sn[0]);
        rmx[1] = row_max(Sn[1]);
      }
    };
    while (kt < kt1) {
      step(S_a, S_b);
      if (kt < kt1) step(S_b, S_a);
    }

    // ---- finish the segment: lane = query row (r of q-block qb), registers = d
    wait32();                                                                    // H3
    auto finish = [&](auto qb_t) {
      constexpr int qb = decltype(qb_t)::value;
#ifdef EXP64_ADDSUM
      const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
#else
      const float l_tot = acc_read<ACC_L + 16 * qb>();      // every register of both lane halves holds the row's full sum
#endif
      const int qq = 64 * wu + 32 * qb + r;
      const bool valid = qq < T.q_rows;
      const float inv = 1.0f / l_tot;
      __bf16* op = a.o + (size_t)(T.q0 + qq) * a.ldo + head * D;
      float* slot = a.ws + (size_t)(sg.slot < 0 ? 0 : sg.slot) * SLOT_FLOATS;
      if (sg.slot >= 0 && hh == 0) { slot[qq] = m_ref[qb]; slot[SLOT_ROWS + qq] = l_tot; }
      float* orow = slot + 2 * SLOT_ROWS + qq * 128;
      auto dblock = [&](auto d_t) {
        constexpr int d = decltype(d_t)::value;
        float o[16];
        acc_read_seq<ACC_O + 64 * qb + 16 * d>(o, std::make_integer_sequence<int, 16>{});
        if (sg.slot < 0) {
          if (valid) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int dc = 32 * d + 8 * g + 4 * hh;
              u32x2 wv = {pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv), pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv)};
              *reinterpret_cast<u32x2*>(op + dc) = wv;
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int dc = 32 * d + 8 * g + 4 * hh;
            *reinterpret_cast<f32x4*>(orow + dc) = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
          }
        }
      };
      dblock(std::integral_constant<int, 0>{}); dblock(std::integral_constant<int, 1>{});
      dblock(std::integral_constant<int, 2>{}); dblock(std::integral_constant<int, 3>{});
    };
    finish(std::integral_constant<int, 0>{});
    finish(std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();                           // the next segment's prologue refills ring slots other waves may still read
  }
  G2V_CLK_END(lb, wu);
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

int g_attn_form = 1;          // A/B switch (tools only): 1 = 4 x 64 form for head dim 128 / 256-row items, 0 = the 8 x 32 form

template <int D>
int launch_flash(const FlashArgs& a, int n_comb, int waves, hipStream_t s) {
  if (a.n_blocks > 0) {
    if (D == 128 && waves == 8 && g_attn_form == 1 && a.ldk == a.ldv) hipLaunchKernelGGL(flash_fwd64_kernel, dim3(a.n_blocks), dim3(256), 0, s, a);
    else if (waves == 8) hipLaunchKernelGGL((flash_fwd_kernel<D, 8>), dim3(a.n_blocks), dim3(512), 0, s, a);
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

#ifdef EXP_STAMPS
extern "C" int g2v_debug_attn_stamps(void* buf) { g_attn_dbg = (unsigned long long*)buf; return 0; }   // >= 3*2*24*8 u64
extern "C" int g2v_debug_attn_stamp_pos(int pos) { g_attn_dbg_pos = pos; return 0; }
extern "C" int g2v_debug_attn_clock(void* buf) { g_attn_clk = (unsigned long long*)buf; return 0; }   // >= 2 * n_blocks u64
#endif

extern "C" int g2v_debug_attn_form(int form) { g_attn_form = form; return 0; }   // tools / tests A/B only

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
#ifdef EXP_STAMPS
  a.dbg = g_attn_dbg;
  a.dbg_pos = g_attn_dbg_pos;
  a.clk = g_attn_clk;
#endif
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

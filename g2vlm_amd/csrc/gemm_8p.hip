// 256x256x64 bf16 GEMM, 8 waves (2 M x 4 N), the "8-phase" schedule of the MI355X guide (§5 "The 256^2 8-phase
// template", T3+T4): every K-tile is cut into four phases, each phase = {LDS fragment reads for ONE 64x32 quadrant
// operand, one half-tile DMA (global_load_lds, 2 instructions per lane), barrier, 16 MFMAs, barrier}.  The DMA ring is
// 8 half-tile slots (2 K-tiles x {A-q0, B-q0, B-q1, A-q1}) and runs SEVEN half-tiles ahead of the reads, retired by
// ONE counted `s_waitcnt vmcnt(6)` per K-tile (never 0 in steady state): HBM/L2 latency is covered by ~2 K-tiles of
// MFMA work instead of one barrier-to-barrier interval as in gemm_big.hip.
//
// Half-tile = the rows a phase consumes, not a contiguous half: A-q0 = rows {wr*128 + [0,64)} of both wave rows,
// B-q0 = cols {wc*64 + [0,32)} of all four wave columns, so phase 1 needs A-q0 + B-q0, phase 2 B-q1, phase 3 A-q1,
// phase 4 nothing new (quadrant order (0,0) (0,1) (1,1) (1,0)).
//
// Hazards (guide "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  K-tile t+1's four half-tiles are retired by the vmcnt at phase 4 of K-tile t, placed BEFORE that phase's first
//        barrier; their first ds_read is in phase 1 of K-tile t+1.
//   WAR  phase p's DMA overwrites the slot of half-tile p-1, whose last ds_read was issued in phase <= p-1 and retired
//        by that phase's lgkmcnt(0) before its closing barrier.
// LDS image: 128-byte rows (64 bf16), 16-byte chunk index ^= (row & 7); the DMA writes lane-linear, so the XOR is applied
// to the per-lane SOURCE address and again on the ds_read_b128 (guide rule 21).  Conflict-free for the 16x16x32 operand
// reads (checked against the ds_read_b128 lane groups of MI355X_MICROARCH.md "LDS").
//
// Same math, epilogues, grouping (und / geo experts) and bf16 rounding points as gemm.hip (reference: every nn.Linear
// under autocast, e.g. modeling/qwen2vl/modeling_qwen2_vl.py:508-521, modeling/g2vlm/qwen2vl.py:579-606).
#include <cstdlib>
#include "common.h"
#include "g2vlm_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int BN = 256, BK = 64;
constexpr int BM_MAX = 288;
constexpr int OP_BYTES = BM_MAX * 128;                   // A tile: up to 288 rows x 128 B = 36 KiB; the 256-row B tile follows it
constexpr int KBUF_BYTES = OP_BYTES + 256 * 128;         // A + B = 68 KiB per K-tile slot, two slots
constexpr int OUT_PITCH = 256 * 2 + 16;                  // epilogue staging: bf16 [bm][256] rows padded by 16 B (conflict-free b64 writes)
constexpr int LDS_BYTES = KBUF_BYTES + 160 * OUT_PITCH;  // epilogue image (<= 160 rows per pass) sits above ring slot 0: 150.5 KiB of 160

struct P8Group {
  const __bf16* A; const __bf16* W; const __bf16* bias; void* C; const void* res; const float* gamma;
  int M, tile_start;
};
struct P8Args {
  P8Group g[2];
  int ngroups, N, K, lda, ldc, ldres, tiles_n, flags, sm, sn, total;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// MA0 / MA1 = 16-row m-fragments per wave in the first / second A quadrant: tile height 32 (MA0 + MA1) = 128 ... 288.
// An A half-tile is 4 MA groups of 8 rows, one 1-KiB DMA instruction each, dealt round-robin to the 8 waves; when MA is
// odd the first four waves issue one instruction more, so the counted vmcnt is chosen per wave (wave-uniform branch).
// PIPE = 1: the LDS fragment reads of phase p+1 are issued BEFORE the MFMAs of phase p (separate registers for the two A
// quadrants), one barrier per phase; DMA runs 9 half-tiles ahead.  PIPE = 0: the guide's template as described above.
template <int EPI, int MA0, int MA1, int PIPE>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(P8Args a) {
  constexpr int MT = MA0 + MA1;                            // m-fragments per wave
  constexpr int HB = 16 * MT;                              // rows per wave row
  constexpr int BMv = 2 * HB;                              // tile height
  constexpr int NA0 = (MA0 + 1) / 2, NA1 = (MA1 + 1) / 2;  // DMA instruction slots per wave for A-q0 / A-q1 (the last may be idle)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int N = a.N, K = a.K;
  // ---- DMA geometry.  One wave-instruction fills 8 rows x 128 B (1 KiB, lane-linear): lane -> row srow, 16-B chunk scp;
  // the chunk it FETCHES is scp ^ srow (all staged rows have row & 7 == srow).
  const int srow = lane >> 3, scp = lane & 7;
  const int kcol = (scp ^ srow) << 3;
  // A quadrant q, instruction i2: linear row L = i2*64 + w*8 of the quadrant's 32*MA rows; wave row L / (16*MA)
  // B half-tile q, instruction i2: group gq = w + 8*i2 -> cols (gq>>2)*64 + q*32 + (gq&3)*8 + srow
  int a_row[2][3], b_col[2][2], a_dst[2][3], b_dst[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i2 = 0; i2 < 3; ++i2) {
      const int MA = q ? MA1 : MA0;
      int Lr = min(i2 * 64 + w * 8, 32 * MA - 8);          // an idle slot (odd MA, waves 4-7) is never issued; keep it in range
      a_row[q][i2] = (Lr / (16 * MA)) * HB + (q ? 16 * MA0 : 0) + Lr % (16 * MA);
      a_dst[q][i2] = a_row[q][i2] * 128;
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      int gq = w + 8 * i2;
      b_col[q][i2] = (gq >> 2) * 64 + q * 32 + (gq & 3) * 8;
      b_dst[q][i2] = OP_BYTES + b_col[q][i2] * 128;
    }

  // what a tile needs besides the kernel arguments: its group, origin and per-lane DMA source offsets
  struct TileCtx {
    int gi, m0, n0;
    uint32_t a_src[2][3], b_src[2][2];
  };
  // tile id -> context: XCD-aware bijective remap, then supertile walk (as gemm_big.hip)
  auto setup_tile = [&](int vt, TileCtx& c) {
    const int nwg = a.total;
    int bid = vt;
    {
      int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
      bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    }
    c.gi = (a.ngroups > 1 && bid >= a.g[1].tile_start) ? 1 : 0;
    const int gM = a.g[c.gi].M;
    const int t_id = bid - a.g[c.gi].tile_start;
    int tm, tn;
    {
      const int tiles_m = (gM + BMv - 1) / BMv;
      const int row_sz = a.sm * a.tiles_n;
      int sup_m = t_id / row_sz, r = t_id - sup_m * row_sz;
      int h = min(a.sm, tiles_m - sup_m * a.sm);
      int full_w = a.sn * h;
      int sup_n = r / full_w, p = r - sup_n * full_w;
      tm = sup_m * a.sm + p % h;                           // walk down the column first: consecutive tiles share the W slab
      tn = sup_n * a.sn + p / h;
    }
    c.m0 = tm * BMv; c.n0 = tn * BN;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i2 = 0; i2 < 3; ++i2)
        c.a_src[q][i2] = ((uint32_t)(min(c.m0 + a_row[q][i2] + srow, gM - 1) - c.m0) * (uint32_t)a.lda + kcol) * 2u;   // BYTES from the tile's first row
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
        c.b_src[q][i2] = ((uint32_t)(min(c.n0 + b_col[q][i2] + srow, N - 1) - c.n0) * (uint32_t)K + kcol) * 2u;
  };
  // issue half-tile j (= 4 t + c) of the tile described by (Ab, Wb, src offsets) into ring slot t & 1
  auto stage_of = [&](int j, const __bf16* Ab, const __bf16* Wb, const uint32_t (&as)[2][3], const uint32_t (&bs)[2][2]) {
    const int t = j >> 2, c = j & 3;
    char* base = smem + (t & 1) * KBUF_BYTES;
    // uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: the saddr form of the DMA, no 64-bit VALU adds
    const char* Ak = reinterpret_cast<const char*>(Ab + t * BK);
    const char* Wk = reinterpret_cast<const char*>(Wb + t * BK);
    if (c == 0) {
#pragma unroll
      for (int i2 = 0; i2 < NA0; ++i2)
        if (i2 * 8 + w < 4 * MA0)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ak + as[0][i2]), (lds_ptr_t)(base + a_dst[0][i2]), 16, 0, 0);
    } else if (c == 3) {
#pragma unroll
      for (int i2 = 0; i2 < NA1; ++i2)
        if (i2 * 8 + w < 4 * MA1)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ak + as[1][i2]), (lds_ptr_t)(base + a_dst[1][i2]), 16, 0, 0);
    } else {
      const int q = c == 2;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Wk + bs[q][i2]), (lds_ptr_t)(base + b_dst[q][i2]), 16, 0, 0);
    }
  };

  // Persistent workgroups (one per CU): a workgroup walks tiles vt = blockIdx.x, + gridDim.x, ...: no per-tile dispatch,
  // the store drain of one tile overlaps the next tile's main loop, and the NEXT tile's first K-tile is DMA'd into ring slot
  // 0 before this tile's epilogue starts (the epilogue stages through the LDS above slot 0), so the next main loop does
  // not begin behind an HBM round trip.  gridDim.x is a multiple of 8, so vt % 8 (the XCD group of the remap) is the
  // workgroup's own XCD for every tile.
  TileCtx cur;
  setup_tile(blockIdx.x, cur);
  bool prefetched = false;                                 // K-tile 0 of `cur` is already in flight / landed in slot 0
  for (int vt = blockIdx.x; vt < a.total; vt += gridDim.x) {
  const P8Group g = a.g[cur.gi];
  const int m0 = cur.m0, n0 = cur.n0, M = g.M;
  const __bf16* Abase = g.A + (size_t)m0 * a.lda;
  const __bf16* Wbase = g.W + (size_t)n0 * K;

  const int nk = K / BK;
  const int n_half = 4 * nk;
  // half-tile j = 4*t + c: c = 0 A-q0, 1 B-q0, 2 B-q1, 3 A-q1; ring slot = K-buffer (t & 1)
  auto stage = [&](int j) {
    if (j >= n_half || (prefetched && j < 4)) return;
    stage_of(j, Abase, Wbase, cur.a_src, cur.b_src);
  };
  // s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate)
  auto wait_vm = [&](int n) {
    switch (n) {
      case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
      case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
      case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
      case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
      case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
      default: __builtin_amdgcn_s_waitcnt(0x0F7A); break;  /* 10 */
    }
  };
  const int cnt_a0 = (4 * MA0 - w + 7) >> 3, cnt_a1 = (4 * MA1 - w + 7) >> 3;   // this wave's DMA instructions per A half-tile
  // leave the three youngest half-tiles (A-q0, B-q0, B-q1 of the K-tile after next) in flight
  auto wait_ring = [&]() { wait_vm(cnt_a0 + 4); };

  // acc[i][j] holds C^T fragments (operands swapped in the MFMA): register r of lane (fr, fq) is
  // C[m = i*16 + fr][n = j*16 + fq*4 + r] -> four consecutive columns of one row, packed 8-byte LDS writes in the epilogue
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read offsets (k-step kk toggles byte bit 6)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fq ^ (fr & 7)) << 4;
  const int aoff = (wr * HB + fr) * 128 + sw;                        // + (qa*16*MA0 + i*16) * 128
  const int boff = OP_BYTES + (wc * 64 + fr) * 128 + sw;             // + qb*4096 + jj*2048

  // A quadrants: time-shared registers (PIPE 0) or one set each (PIPE 1); B q0 and q1
  bf16x8 fa[PIPE == 1 ? 2 : 1][MA0][2], fb[2][2][2];

  // EXP_GEMM_* (tools/gemm_variants.sh): timing ablations only, wrong results; the shipped library defines none of them
  auto read_a = [&](const char* base, int qa) {
#pragma unroll
    for (int i = 0; i < (qa ? MA1 : MA0); ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#ifdef EXP_GEMM_HALF_LDS
        if (i & 1) { fa[PIPE == 1 ? qa : 0][i][kk] = fa[PIPE == 1 ? qa : 0][i - 1][kk]; continue; }
#endif
        fa[PIPE == 1 ? qa : 0][i][kk] = *reinterpret_cast<const bf16x8*>(base + ((aoff + (qa * 16 * MA0 + i * 16) * 128) ^ (kk << 6)));
      }
  };
  auto read_b = [&](const char* base, int qb) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#ifdef EXP_GEMM_HALF_LDS
        if (jj & 1) { fb[qb][jj][kk] = fb[qb][0][kk]; continue; }
#endif
        fb[qb][jj][kk] = *reinterpret_cast<const bf16x8*>(base + ((boff + qb * 4096 + jj * 2048) ^ (kk << 6)));
      }
  };
  // A wave row without a valid row in this tile (the und group's 8-16 rows and the geo group's last 8 rows each fill a 288-row
  // tile: 2 of 40 tile rows at C3) keeps its staging and barrier duty and skips its MFMAs: their results are never stored, and
  // in the pipeline MFMA work is what the step's time is made of.  Wave-uniform.
  const bool live = wr * HB < M - m0 || (a.flags & G2V_GEMM_8P_NO_ROW_SKIP);
  auto mma = [&](int qa, int qb) {
    if (!live) return;
#ifdef EXP_GEMM_NO_MFMA
    // keep the fragment reads alive at the price of one VALU op per fragment register pair
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < (qa ? MA1 : MA0); ++i) acc[qa * MA0 + i][qb * 2][0] += (float)fa[PIPE == 1 ? qa : 0][i][kk][0];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[qa * MA0][qb * 2 + jj][1] += (float)fb[qb][jj][kk][0];
    }
    return;
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < (qa ? MA1 : MA0); ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[qa * MA0 + i][qb * 2 + jj] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[qb][jj][kk], fa[PIPE == 1 ? qa : 0][i][kk], acc[qa * MA0 + i][qb * 2 + jj], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // PIPE = 2: the two-barrier form with the two wave rows STAGGERED by one barrier (the guide's `if (wr == 1) s_barrier`):
  // every SIMD holds one wave of each row, so while one of them is in its MFMA segment (between a phase's two barriers) the
  // other is in its load segment (ds_reads + LDS-DMA issue, 100-185 cycles per DMA piece) - in lockstep both waves of a SIMD
  // stall on the DMA issue together and then queue for the MFMA pipe together (ablations: without MFMAs the loop takes 64 % of
  // its time, without the epilogue 90 %, with half the LDS reads 96 %).  Hazards under the stagger: the wave row that is
  // ahead restages slot A-q0 one phase after the row behind has ISSUED its reads of it, so the reads are retired
  // (lgkmcnt(0)) BEFORE a phase's first barrier, not after it; the LDS-DMA retire (counted vmcnt before phase 4's first
  // barrier, first read after its second) already has the extra barrier the guide asks for.
  constexpr bool STAG = PIPE == 2;
  auto sync_lds = [&]() {
    if constexpr (STAG) {
      __builtin_amdgcn_s_waitcnt(0xC07F);               /* lgkmcnt(0) */
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    } else {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);               /* lgkmcnt(0) */
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  if constexpr (PIPE != 1) {
    // one K-tile = 4 phases; p = 4*t is the global phase index, phase p issues half-tile p + 7
    auto ktile = [&](int t) {
      const char* base = smem + (t & 1) * KBUF_BYTES;
      const int p = 4 * t;
      // phase 1: A-q0, B-q0 -> quadrant (0,0)
      read_b(base, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_a(base, 0);
      stage(p + 7);
      sync_lds();
      mma(0, 0);
      __builtin_amdgcn_s_barrier();
      // phase 2: B-q1 -> (0,1)
      read_b(base, 1);
      stage(p + 8);
      sync_lds();
      mma(0, 1);
      __builtin_amdgcn_s_barrier();
      // phase 3: A-q1 -> (1,1)
      read_a(base, 1);
      stage(p + 9);
      sync_lds();
      mma(1, 1);
      __builtin_amdgcn_s_barrier();
      // phase 4: nothing new -> (1,0); retire K-tile t+1's half-tiles, leaving the 3 youngest in flight
      stage(p + 10);
      if (t + 2 < nk) wait_ring();
      else __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */
      __builtin_amdgcn_s_barrier();
      mma(1, 0);
      __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: 7 half-tiles in flight, the first K-tile (4 of them) retired
  #pragma unroll
    for (int j = 0; j < 7; ++j) stage(j);
    if (nk >= 2) wait_ring();
    else __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */
    __builtin_amdgcn_s_barrier();
    if constexpr (STAG) {
      if (wr == 1) __builtin_amdgcn_s_barrier();             // wave row 1 runs one barrier behind row 0 from here on
    }

    for (int t = 0; t < nk; ++t) ktile(t);
    if constexpr (STAG) {
      if (wr == 0) __builtin_amdgcn_s_barrier();             // re-align: row 1's last MFMA segment ends at this barrier
    }
  } else {
    // Software-pipelined form.  Phase p (K-tile t = p / 4, i = p % 4):
    //   lgkmcnt(0) -> [i == 2: retire K-tile t+1's DMA] -> barrier -> DMA of half-tile p + 9 -> LDS reads for phase p + 1
    //   -> 16 MFMAs of phase p.   Operand use: i=0 (A0,B0)  i=1 (A0,B1)  i=2 (A1,B1)  i=3 (A1,B0); reads issued in
    //   i=0: B1   i=1: A1   i=2: A0 of t+1   i=3: B0 of t+1 (after the MFMAs that still use B0).
    // Half-tile j is therefore read in phase j - 2, retired (vmcnt + barrier) at the start of phase 4t + 2 >= ... <= j - 2,
    // and its slot is overwritten by half-tile j + 8, issued in phase j - 1 after that phase's barrier, i.e. after every
    // wave's lgkmcnt(0) for the reads of phase j - 2 (WAR).
    auto begin_phase = [&](bool retire, int t) {
      __builtin_amdgcn_s_waitcnt(0xC07F);               /* lgkmcnt(0) */
      if (retire && t + 1 < nk) {
        if (t + 2 < nk) wait_ring();
        else __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) stage(j);
    if (nk >= 2) {
      wait_vm(cnt_a0 + 4 + cnt_a1);                        // K-tile 1's four half-tiles stay in flight
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */
    }
    __builtin_amdgcn_s_barrier();
    read_a(smem, 0);
    read_b(smem, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);               /* lgkmcnt(0) */
    __builtin_amdgcn_s_barrier();
    stage(8);
    for (int t = 0; t < nk; ++t) {
      const char* base = smem + (t & 1) * KBUF_BYTES;
      const char* nbase = smem + ((t + 1) & 1) * KBUF_BYTES;
      const int p = 4 * t;
      begin_phase(false, t);
      stage(p + 9);
      read_b(base, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      begin_phase(false, t);
      stage(p + 10);
      read_a(base, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      begin_phase(true, t);
      stage(p + 11);
      read_a(nbase, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      begin_phase(false, t);
      stage(p + 12);
      __builtin_amdgcn_sched_barrier(0);
      mma(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_b(nbase, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);               /* lgkmcnt(0) */
    __builtin_amdgcn_s_barrier();                          // the epilogue reuses the LDS the last reads came from
  }

  // ---- the next tile of this workgroup: its first K-tile goes into ring slot 0 now (every wave is past its last LDS
  // read), and lands while the epilogue below runs out of the LDS above slot 0
  const int vn = vt + gridDim.x;
  const bool has_next = vn < a.total;
  if (has_next) {
    setup_tile(vn, cur);                                   // m0 / n0 / g of the tile being finished are in locals
    const P8Group& gn_ = a.g[cur.gi];
    const __bf16* An = gn_.A + (size_t)cur.m0 * a.lda;
    const __bf16* Wn = gn_.W + (size_t)cur.n0 * K;
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_of(j, An, Wn, cur.a_src, cur.b_src);
  }

#ifdef EXP_GEMM_NO_EPI
  {                                                        // keep the accumulators alive; no staging, no stores
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sink == 1.2345e30f) reinterpret_cast<float*>(g.C)[tid] = sink;
    __builtin_amdgcn_s_barrier();
    prefetched = has_next;
    continue;
  }
#endif
  // ------------------------------------------------------------------ epilogue
  // Two passes over the m-fragments (i < I0, then the rest), each:
  // (1) every lane rounds its accumulators to the bf16 Linear output (bias, activation) and writes them, 4 consecutive
  //     columns = 8 bytes at a time, into a row-major bf16 image in LDS: image row wr*16*I0 + (i - h*I0)*16 + fr;
  // (2) the block walks that image in 16-byte chunks, 32 (16 for SwiGLU) consecutive lanes per output row, so every
  //     global access - output store, and for the residual forms the fp32/bf16 residual load - is a full 16-byte
  //     lane access on 512 (256) contiguous bytes per row instead of 2- and 4-byte scattered ones.
  constexpr bool SWI = EPI == G2V_EPI_SWIGLU;
  constexpr int PITCH = OUT_PITCH;
  constexpr int I0 = (MT + 1) / 2;                         // m-fragments per pass
  constexpr int CPR = SWI ? 16 : 32;                       // 16-byte chunks per output row
  constexpr int RPI = 512 / CPR;                           // rows per sweep
  char* const img = smem + KBUF_BYTES;
  // the epilogue's lane constants are recomputed per tile from an opaque copy of the thread id: hoisted out of the
  // persistent loop they would sit in (or spill from) the registers the main loop needs
  int etid = tid;
  asm volatile("" : "+v"(etid));
  const int efr = etid & 15, efq = (etid >> 4) & 3;
  const int ch = etid % CPR, r0 = etid / CPR;
  const int gn = (SWI ? (n0 >> 1) : n0) + ch * 8;          // first of this lane's 8 output columns
  const bool round_gamma = a.flags & G2V_GEMM_GAMMA_ROUND_BF16;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int irow0 = wr * 16 * I0 + efr;
    if constexpr (SWI) {
#pragma unroll
      for (int ii = 0; ii < I0; ++ii) {
        const int i = h * I0 + ii;
        if (i < MT) {
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float gt = bfround(acc[i][2 * jp][r]);
              float up = bfround(acc[i][2 * jp + 1][r]);
              float sl = bfround(siluf_(gt));
              o[r] = sl * up;
            }
            *reinterpret_cast<u32x2*>(img + (irow0 + ii * 16) * PITCH + (wc * 32 + jp * 16 + efq * 4) * 2) =
                u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = wc * 64 + j * 16 + efq * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
          u32x2 bb = *reinterpret_cast<const u32x2*>(g.bias + n0 + cl);
          bv[0] = bits2f_lo(bb[0]); bv[1] = bits2f_hi(bb[0]); bv[2] = bits2f_lo(bb[1]); bv[3] = bits2f_hi(bb[1]);
        }
#pragma unroll
        for (int ii = 0; ii < I0; ++ii) {
          const int i = h * I0 + ii;
          if (i < MT) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = bfround(acc[i][j][r] + bv[r]);
              if constexpr (EPI == G2V_EPI_GELU) v = gelu_fast(v);
              if constexpr (EPI == G2V_EPI_QUICKGELU) {
                float u = bfround(1.702f * v);
                float sg = bfround(sigmoidf_(u));
                v = v * sg;
              }
              o[r] = v;
            }
            *reinterpret_cast<u32x2*>(img + (irow0 + ii * 16) * PITCH + cl * 2) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          }
        }
      }
    }
    float gam[8];
    bool has_gam = false;
    if constexpr (EPI == G2V_EPI_RES_F32) {
      has_gam = g.gamma != nullptr;
      if (has_gam) {
        f32x4 g0 = *reinterpret_cast<const f32x4*>(g.gamma + gn), g1 = *reinterpret_cast<const f32x4*>(g.gamma + gn + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gam[e] = g0[e]; gam[4 + e] = g1[e]; }
      }
    }
    __syncthreads();
    // sweep in batches: residual loads and LDS reads of a batch are all issued before the first store waits on them
    constexpr int NIT = 2 * 16 * I0 / RPI;
    constexpr int SB = NIT <= 5 ? NIT : NIT / 2;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += SB) {
      u32x4 pk[SB];
      int gmv[SB];
      bool ok[SB];
      [[maybe_unused]] f32x4 ra[SB], rb[SB];
      [[maybe_unused]] u32x4 rr[SB];
#pragma unroll
      for (int c = 0; c < SB; ++c) {
        const int irow = (it0 + c) * RPI + r0;
        const int iwr = irow / (16 * I0), rem = irow - iwr * (16 * I0);
        const int trow = h * 16 * I0 + rem;                // row inside the wave row
        gmv[c] = m0 + iwr * HB + trow;
        ok[c] = trow < HB && gmv[c] < M;
        if constexpr (EPI == G2V_EPI_RES_F32) {
          ra[c] = rb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ok[c] && g.res) {
            const float* rp = reinterpret_cast<const float*>(g.res) + (size_t)gmv[c] * a.ldres + gn;
            ra[c] = *reinterpret_cast<const f32x4*>(rp); rb[c] = *reinterpret_cast<const f32x4*>(rp + 4);
          }
        } else if constexpr (EPI == G2V_EPI_RES_BF16) {
          rr[c] = u32x4{0u, 0u, 0u, 0u};
          if (ok[c]) rr[c] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(g.res) + (size_t)gmv[c] * a.ldres + gn);
        }
      }
#pragma unroll
      for (int c = 0; c < SB; ++c) pk[c] = *reinterpret_cast<const u32x4*>(img + ((it0 + c) * RPI + r0) * PITCH + ch * 16);
#pragma unroll
      for (int c = 0; c < SB; ++c) {
        if (!ok[c]) continue;
        const int gm = gmv[c];
        if constexpr (EPI == G2V_EPI_RES_F32) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] = bits2f_lo(pk[c][e]); v[2 * e + 1] = bits2f_hi(pk[c][e]); }
          if (has_gam) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] = __fmul_rn(v[e], gam[e]);
              if (round_gamma) v[e] = bfround(v[e]);
            }
          }
          float* cp = reinterpret_cast<float*>(g.C) + (size_t)gm * a.ldc + gn;
          *reinterpret_cast<f32x4*>(cp) = f32x4{__fadd_rn(ra[c][0], v[0]), __fadd_rn(ra[c][1], v[1]), __fadd_rn(ra[c][2], v[2]), __fadd_rn(ra[c][3], v[3])};
          *reinterpret_cast<f32x4*>(cp + 4) = f32x4{__fadd_rn(rb[c][0], v[4]), __fadd_rn(rb[c][1], v[5]), __fadd_rn(rb[c][2], v[6]), __fadd_rn(rb[c][3], v[7])};
        } else if constexpr (EPI == G2V_EPI_RES_BF16) {
          u32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            ov[e] = pack_bf16x2(bits2f_lo(rr[c][e]) + bits2f_lo(pk[c][e]), bits2f_hi(rr[c][e]) + bits2f_hi(pk[c][e]));
          *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(g.C) + (size_t)gm * a.ldc + gn) = ov;
        } else {
          *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(g.C) + (size_t)gm * a.ldc + gn) = pk[c];
        }
      }
    }
    __syncthreads();                                      // the image is dead: the second pass / the next tile's DMA may overwrite it
  }
  prefetched = has_next;
  }
}

template <int EPI, int MA0, int MA1, int PIPE>
int launch_h(const P8Args& a, int total, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<EPI, MA0, MA1, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess) return G2V_ERR_LAUNCH;
    attr_set = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return G2V_ERR_LAUNCH;
    n_cu = prop.multiProcessorCount & ~7;                  // multiple of 8: a workgroup keeps its XCD group across tiles
    if (n_cu <= 0) n_cu = 256;
  }
  P8Args b = a;
  b.total = total;
  hipLaunchKernelGGL((gemm8p_kernel<EPI, MA0, MA1, PIPE>), dim3(total < n_cu ? total : n_cu), dim3(512), LDS_BYTES, s, b);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

template <int EPI>
int launch(const P8Args& a, int bm, int total, hipStream_t s) {
  if (a.flags & G2V_GEMM_8P_TWO_BARRIER) {                                    // A/B: the two-barrier loop in lockstep
    if (bm == 288) return launch_h<EPI, 5, 4, 0>(a, total, s);
    if (bm == 256) return launch_h<EPI, 4, 4, 0>(a, total, s);
    if (bm == 192) return launch_h<EPI, 4, 2, 0>(a, total, s);
    return launch_h<EPI, 2, 2, 0>(a, total, s);
  }
  if (a.flags & G2V_GEMM_8P_PIPELINED) {                                      // A/B: round 1's default (one barrier per phase)
    if (bm == 288) return launch_h<EPI, 5, 4, 0>(a, total, s);  // 144 accumulators: only the register-lean two-barrier form fits
    if (bm == 256) return launch_h<EPI, 4, 4, 1>(a, total, s);
    if (bm == 224) return launch_h<EPI, 4, 3, 1>(a, total, s);
    if (bm == 192) return launch_h<EPI, 4, 2, 1>(a, total, s);
    if (bm == 160) return launch_h<EPI, 3, 2, 1>(a, total, s);
    return launch_h<EPI, 2, 2, 1>(a, total, s);
  }
  // G2V_GEMM_8P_EIGHT_WAVES (round 2's default): two barriers per phase, the two wave rows staggered by one barrier (1.09 -> 1.36
  // PF at 8192^3, 1.10 -> 1.38 PF on the down projection, tools/gemm_square.py)
  if (bm == 288) return launch_h<EPI, 5, 4, 2>(a, total, s);
  if (bm == 256) return launch_h<EPI, 4, 4, 2>(a, total, s);
  if (bm == 224) return launch_h<EPI, 4, 3, 2>(a, total, s);
  if (bm == 192) return launch_h<EPI, 4, 2, 2>(a, total, s);
  if (bm == 160) return launch_h<EPI, 3, 2, 2>(a, total, s);
  return launch_h<EPI, 2, 2, 2>(a, total, s);
}

}  // namespace

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool g2v_gemm_8p_supported(const g2v_gemm_desc* d) {
  if ((d->K % BK) || (d->N % BN) || (d->lda & 7) || d->K < 2 * BK) return false;
  // 32-bit element offsets inside one tile
  if ((long)BM_MAX * d->lda >= (1L << 31) || (long)BN * d->K >= (1L << 31)) return false;
  // the coalesced epilogue moves 16-byte chunks
  const bool f32out = d->epilogue == G2V_EPI_RES_F32;
  if (d->ldc % (f32out ? 4 : 8)) return false;
  for (int i = 0; i < d->ngroups; ++i) {
    const g2v_gemm_group& g = d->g[i];
    if (g.M <= 0) continue;
    if (!aligned16(g.A) || !aligned16(g.W) || !aligned16(g.C)) return false;
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 7)) return false;
    if (g.gamma && !aligned16(g.gamma)) return false;
    if (g.res && (!aligned16(g.res) || d->ldres % (f32out ? 4 : 8))) return false;
  }
  return true;
}

// Policy (tools/bench_kernels.py on the C3 shapes): filled in from measurements
bool g2v_gemm_8p_preferred(const g2v_gemm_desc* d) {
  long rows = 0;
  for (int i = 0; i < d->ngroups; ++i) rows += d->g[i].M;
  // with the 288-row form the narrow-N / long-K shapes (down-proj, decoder fc2) also fit one round of 256 CUs and beat
  // gemm_big.hip (0.27 vs 0.31 ms, 0.19 vs 0.21 ms).  A few hundred rows (one ViT image's 731 tokens through the und expert):
  // wide N still fills the chip with 128/160-row tiles (gate/up at 731 rows 0.047 vs 0.062 ms); narrow N is one tile-time of
  // latency whatever the kernel and stays with gemm_big.hip's 32-row tiles / the 128x128 kernel
  return rows >= 1024 || (rows >= 256 && d->N >= 4096);
}

int g2v_gemm_8p_launch(const g2v_gemm_desc* d, hipStream_t s) {
  P8Args a;
  a.ngroups = 0; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldres = d->ldres;
  a.tiles_n = d->N / BN; a.flags = d->flags;
  // 32 resident tiles per XCD: sm x sn supertile of 4 x 8 (or all of N when narrower)
  a.sn = a.tiles_n < 8 ? a.tiles_n : 8;
  a.sm = 32 / a.sn > 1 ? 32 / a.sn : 1;
  int order[2] = {0, 1};
  if (d->ngroups == 2 && d->g[1].M > d->g[0].M) { order[0] = 1; order[1] = 0; }
  // tile height from the large group; the small (und) group rides along with the same height
  const long m_big = d->g[order[0]].M;
  const long m_small = d->ngroups == 2 ? d->g[order[1]].M : 0;
  int bm = 256;
  {
    // cost of a launch = rounds of 256 resident tiles x tile height, with a handicap for shorter tiles (less reuse per
    // staged byte, the fixed prologue/epilogue per tile)
    double best = 1e30;
    const int hs[6] = {288, 256, 224, 192, 160, 128};
    // time per row of tile height relative to 256 rows, measured where quantisation plays no part (gate/up at C3, 11-24 rounds,
    // staggered main loop; tools/gemm_heights.py): 1.005 / 1 / 1.077 / 1.123 / 1.165 / 1.23
    const double hc[6] = {1.005, 1.0, 1.08, 1.12, 1.165, 1.23};
    for (int k = 0; k < 6; ++k) {
      const int h = hs[k];
      const int small_rows = m_small > 0 ? (int)((m_small + h - 1) / h) : 0;
      const long tiles = ((m_big + h - 1) / h + small_rows) * a.tiles_n;
      const long rounds = (tiles + 255) / 256;
      const double c = (double)rounds * h * hc[k];
      if (c < best) { best = c; bm = h; }
    }
  }
  if (d->flags & G2V_GEMM_8P_H192) bm = 192;                            // A/B testing of the tile heights
  if (d->flags & G2V_GEMM_8P_H128) bm = 128;
  if (d->flags & G2V_GEMM_8P_H256) bm = 256;
  if (d->flags & G2V_GEMM_8P_H288) bm = 288;
  if (d->flags & G2V_GEMM_8P_H224) bm = 224;
  if (d->flags & G2V_GEMM_8P_H160) bm = 160;
  // four-wave form (gemm_4w.hip) or the eight-wave loops of this file (bit-identical A/B partners): the flags force one, otherwise
  // by shape class (G2V_GEMM_4W_MASK: bit 0 wide N (gate/up), bit 1 long K (down, fc2), bit 2 other fp32-residual Linears (o-proj),
  // bit 3 plain bf16 outputs (qkv), bit 4 GELU (fc1); default 6 = what the C3 step measures fastest IN SITU on a power-limited chip)
  if (!(d->flags & (G2V_GEMM_8P_EIGHT_WAVES | G2V_GEMM_8P_TWO_BARRIER | G2V_GEMM_8P_PIPELINED))) {
    static int mask = -1;
    if (mask < 0) {
      const char* e = getenv("G2V_GEMM_4W_MASK");
      mask = e ? atoi(e) : 6;
    }
    const int cls = d->N >= 8192 ? 1 : (d->K >= 4096 ? 2 : (d->epilogue == G2V_EPI_RES_F32 ? 4 : (d->epilogue == G2V_EPI_GELU ? 16 : 8)));
    if ((d->flags & G2V_GEMM_8P_FOUR_WAVES) || (mask & cls)) return g2v_gemm_4w_launch(d, bm, order, s);
  }
  int total = 0;
  for (int i = 0; i < d->ngroups; ++i) {
    const g2v_gemm_group& sg = d->g[order[i]];
    if (sg.M <= 0) continue;
    P8Group& g = a.g[a.ngroups++];
    g.A = (const __bf16*)sg.A; g.W = (const __bf16*)sg.W; g.bias = (const __bf16*)sg.bias; g.C = sg.C; g.res = sg.res;
    g.gamma = (const float*)sg.gamma; g.M = sg.M; g.tile_start = total;
    total += ((sg.M + bm - 1) / bm) * a.tiles_n;
  }
  if (a.ngroups == 1) a.g[1] = a.g[0];
  if (total == 0) return G2V_OK;
  switch (d->epilogue) {
    case G2V_EPI_BF16: return launch<G2V_EPI_BF16>(a, bm, total, s);
    case G2V_EPI_GELU: return launch<G2V_EPI_GELU>(a, bm, total, s);
    case G2V_EPI_QUICKGELU: return launch<G2V_EPI_QUICKGELU>(a, bm, total, s);
    case G2V_EPI_SWIGLU: return launch<G2V_EPI_SWIGLU>(a, bm, total, s);
    case G2V_EPI_RES_F32: return launch<G2V_EPI_RES_F32>(a, bm, total, s);
    case G2V_EPI_RES_BF16: return launch<G2V_EPI_RES_BF16>(a, bm, total, s);
    default: return G2V_ERR_ARG;
  }
}

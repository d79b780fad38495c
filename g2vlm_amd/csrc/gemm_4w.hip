// 256x256x64 bf16 GEMM, FOUR waves (2 M x 2 N), one wave per SIMD, 128-column x (128 ... 144)-row wave tiles.
//
// Why a second form of gemm_8p.hip's kernel: a SIMD issues the instructions of its waves one after the other, so with two
// waves per SIMD the LDS-DMA issue (60-185 cycles a piece), the fragment reads and the barriers of one wave are not hidden
// behind the other wave's MFMAs - they queue in front of them (gemm_8p.hip, staggered: load segment ~1.7 x the MFMA segment,
// 59 % MFMA busy, 1.35 PFLOP/s at 8192^3).  Here every wave owns a whole SIMD and 512 registers: a 128-column wave tile
// needs 0.25 fragment reads per MFMA instead of 0.375, there is ONE barrier per 64-72 MFMAs, and every load instruction is
// hand-placed between two MFMAs of the SAME wave (the matrix pipe runs on while the next instruction issues).  This is the
// shape the vendor library's own 256x256x64 kernels have on this chip (4 waves, 8 x 8 fragments of 16x16 per wave).
//
// Ring, half-tiles, LDS image, hazards: as gemm_8p.hip's software-pipelined form.  Half-tile j = 4 t + c of K-tile t:
// c = 0 A-q0 (rows wr*HB + [0, 16 MA0)), 1 B-q0 (cols wc*128 + [0, 64)), 2 B-q1 (cols wc*128 + [64, 128)), 3 A-q1.  Phase
// p = 4 t + i:  lgkmcnt(0) -> [i == 2: counted vmcnt retiring K-tile t+1] -> [i even: barrier] -> { MFMAs of quadrant i,
// between them: fragment reads for phase p + 1, LDS-DMA of half-tile p + 8 }.
//   quadrant / operands   i=0 (A0, B0)   i=1 (A0, B1)   i=2 (A1, B1)   i=3 (A1, B0)
//   reads issued          B1 of t        A1 of t        A0 of t+1      B0 of t+1 (second B0 register set: B0 of t is in use)
//   DMA issued            A-q0 of t+2    B-q0 of t+2    B-q1 of t+2    A-q1 of t+2
//   RAW  half-tile j is first read in phase j - 2; K-tile t+1's four half-tiles are retired (vmcnt + barrier) at the start of
//        phase 4 t + 2, whose reads are the first of that K-tile.
//   WAR  the slot of half-tile j is overwritten by half-tile j + 8, issued in phase j; its reads were issued in phase j - 2 and
//        retired by every wave's lgkmcnt(0) at the start of phase j - 1.  The barrier in front of phase j (j even) or j - 1 (j
//        odd) therefore lies between every wave's last read of the slot and every wave's DMA into it: TWO barriers per K-tile.
// The LDS-DMA is an asm statement (lds_dma.h): the landing of the pieces is tracked by the counted vmcnt above, not by hipcc.
//
// Same math, epilogues, grouping (und / geo experts) and bf16 rounding points as gemm.hip / gemm_8p.hip (reference: every
// nn.Linear under autocast, e.g. modeling/qwen2vl/modeling_qwen2_vl.py:508-521, modeling/g2vlm/qwen2vl.py:579-606).
#include <type_traits>
#include <utility>
#include "common.h"
#include "lds_dma.h"
#include "g2vlm_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int BN = 256, BK = 64, NT = 256;
constexpr int BM_MAX = 288;
constexpr int OP_BYTES = BM_MAX * 128;                   // A tile: up to 288 rows x 128 B; the 256-row B tile follows it
constexpr int KBUF_BYTES = OP_BYTES + 256 * 128;         // A + B = 68 KiB per K-tile slot, two slots
constexpr int OUT_PITCH = 256 * 2 + 16;                  // epilogue staging: bf16 [rows][256], rows padded by 16 B
constexpr int LDS_BYTES = KBUF_BYTES + 160 * OUT_PITCH;  // the epilogue image (<= 160 rows per pass) sits above ring slot 0

struct W4Group {
  const __bf16* A; const __bf16* W; const __bf16* bias; void* C; const void* res; const float* gamma;
  int M, tile_start;
};
struct W4Args {
  W4Group g[2];
  int ngroups, N, K, lda, ldc, ldres, tiles_n, flags, sm, sn, total;
};

// the K-tile bases are formed by SALU adds from per-tile bases (no VALU-written SGPR): only the M0 write needs its wait state
#ifdef EXP_4W_NO_READS
#define RD(x)
#else
#define RD(x) x
#endif
#ifdef EXP_4W_DMA_NOP4
#define DMA16 dma16_saddr
#else
#define DMA16 dma16_saddr_settled
#endif

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// The accumulators live in a0 .. a255, OWNED by this file: hipcc cannot be made to keep 256 accumulator registers in AGPRs
// beside 200 VGPRs of fragments (builtin MFMAs: 2.8 v_accvgpr moves per MFMA and scratch spills in the main loop), so the
// MFMAs are asm statements naming their accumulator registers literally, as in attn.hip's flash_fwd64_kernel; the compiler
// keeps every C++ value in the 256 VGPRs (tests/test_build_cpu.py audits: no compiler-made AGPR use, no scratch).
#define G2V_A8(k) "a" #k "0", "a" #k "1", "a" #k "2", "a" #k "3", "a" #k "4", "a" #k "5", "a" #k "6", "a" #k "7", "a" #k "8", "a" #k "9"
__device__ __forceinline__ void acc_declare() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", G2V_A8(1), G2V_A8(2), G2V_A8(3), G2V_A8(4), G2V_A8(5), G2V_A8(6),
               G2V_A8(7), G2V_A8(8), G2V_A8(9), G2V_A8(10), G2V_A8(11), G2V_A8(12), G2V_A8(13), G2V_A8(14), G2V_A8(15), G2V_A8(16), G2V_A8(17),
               G2V_A8(18), G2V_A8(19), G2V_A8(20), G2V_A8(21), G2V_A8(22), G2V_A8(23), G2V_A8(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
template <int R> __device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "i"(R)); }
template <int R> __device__ __forceinline__ float acc_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R)); return v; }
__device__ __forceinline__ void wait32() { asm volatile("s_nop 15\n\ts_nop 15"); }
// a[R .. R+3] += B fragment . A fragment (16x16x32; the operands are swapped so that the accumulator holds C^T fragments)
template <int R> __device__ __forceinline__ void mfma_acc(const bf16x8& b, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(b), "v"(a), "i"(R), "i"(R + 3));
}
// the same into a VGPR accumulator (the ninth m-fragment of the 288-row tile: a0 .. a255 hold the first eight)
__device__ __forceinline__ void mfma_vacc(f32x4& c, const bf16x8& b, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(b), "v"(a));
}

template <int EPI, int MA0, int MA1>
__global__ __launch_bounds__(NT, 1) void gemm4w_kernel(W4Args a) {
  static_assert(MA0 + MA1 <= 9, "256 accumulator AGPRs + 32 VGPRs");
  acc_declare();
  constexpr int MT = MA0 + MA1;                            // 16-row m-fragments per wave
  constexpr int HB = 16 * MT;                              // rows per wave row
  constexpr int BMv = 2 * HB;                              // tile height
  constexpr int MAX_MA = MA0 > MA1 ? MA0 : MA1;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int N = a.N, K = a.K;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // ---- DMA geometry.  One wave-instruction fills 8 rows x 128 B (1 KiB, lane-linear): lane -> row srow, 16-B chunk scp;
  // the chunk it FETCHES is scp ^ srow (all staged rows have row & 7 == srow).
  const int srow = lane >> 3, scp = lane & 7;
  const int kcol = (scp ^ srow) << 3;
  // A quadrant q, piece i2 < MAq: linear row L = (4 i2 + w) * 8 of the quadrant's 32 MAq rows; wave row L / (16 MAq)
  // B half q, piece i2 < 4: group gq = w + 4 i2 -> cols (gq >> 3) * 128 + q * 64 + (gq & 7) * 8 + srow
  int a_row[2][MAX_MA], b_col[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i2 = 0; i2 < MAX_MA; ++i2) {
      const int MA = q ? MA1 : MA0;
      const int Lr = min((4 * i2 + w) * 8, 32 * MA - 8);
      a_row[q][i2] = (Lr / (16 * MA)) * HB + (q ? 16 * MA0 : 0) + Lr % (16 * MA);
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
      const int gq = w + 4 * i2;
      b_col[q][i2] = (gq >> 3) * 128 + q * 64 + (gq & 7) * 8;
    }

  struct TileCtx {
    int gi, m0, n0;
    uint32_t a_src[2][MAX_MA], b_src[2][4];                // per-lane byte offsets from the tile's first A row / W row
  };
  // tile id -> context: XCD-aware bijective remap, then supertile walk (as gemm_8p.hip)
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
      for (int i2 = 0; i2 < MAX_MA; ++i2)
        c.a_src[q][i2] = ((uint32_t)(min(c.m0 + a_row[q][i2] + srow, gM - 1) - c.m0) * (uint32_t)a.lda + kcol) * 2u;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
        c.b_src[q][i2] = ((uint32_t)(min(c.n0 + b_col[q][i2] + srow, N - 1) - c.n0) * (uint32_t)K + kcol) * 2u;
  };
  // piece i2 of half-tile (t, c) of the tile (Ab, Wb: wave-uniform byte pointers to the tile's first A row / W row)
  auto dma_piece = [&](int t, auto c_, auto i2_, const char* Ab, const char* Wb, const uint32_t (&as)[2][MAX_MA], const uint32_t (&bs)[2][4]) {
    constexpr int c = c_, i2 = i2_;
    const uint32_t slot = lds0 + (uint32_t)((t & 1) * KBUF_BYTES);
    if constexpr (c == 0) DMA16(Ab + (size_t)t * (BK * 2), as[0][i2], slot + (uint32_t)a_row[0][i2] * 128u);
    else if constexpr (c == 3) DMA16(Ab + (size_t)t * (BK * 2), as[1][i2], slot + (uint32_t)a_row[1][i2] * 128u);
    else DMA16(Wb + (size_t)t * (BK * 2), bs[c == 2][i2], slot + (uint32_t)(OP_BYTES + b_col[c == 2][i2] * 128));
  };
  // every piece of half-tile c of K-tile t (MA0, 4, 4, MA1 pieces per wave for c = 0 .. 3)
  auto stage_half = [&](int t, auto c_, const char* Ab, const char* Wb, const uint32_t (&as)[2][MAX_MA], const uint32_t (&bs)[2][4]) {
    constexpr int c = c_;
    constexpr int NP = c == 0 ? MA0 : (c == 3 ? MA1 : 4);
    sfor<NP>([&](auto i2) { dma_piece(t, c_, i2, Ab, Wb, as, bs); });
  };
  // s_waitcnt vmcnt(n): the instruction takes an immediate (vmcnt = bits 3:0 and 15:14; expcnt / lgkmcnt left at their maximum)
  auto wait_vm = [&](auto n_) {
    constexpr int n = n_;
    __builtin_amdgcn_s_waitcnt(0x0F70 | (n & 15) | ((n >> 4) << 14));
  };

  TileCtx cur;
  setup_tile(blockIdx.x, cur);
  bool prefetched = false;                                 // K-tile 0 of `cur` is already in flight / landed in slot 0
  for (int vt = blockIdx.x; vt < a.total; vt += gridDim.x) {
  const W4Group g = a.g[cur.gi];
  const int m0 = cur.m0, n0 = cur.n0, M = g.M;
  const char* Ab = uniform_ptr(reinterpret_cast<const char*>(g.A + (size_t)m0 * a.lda));
  const char* Wb = uniform_ptr(reinterpret_cast<const char*>(g.W + (size_t)n0 * K));
  const int nk = K / BK;

  // accumulator (i, j) = a[4 (8 i + j) .. + 3] holds a C^T fragment (operands swapped in the MFMA): register r of lane
  // (fr, fq) is C[m = i*16 + fr][n = j*16 + fq*4 + r]; j = 4 qb + jj.  Zeroed below, under the prologue's DMA latency.
  // m-fragment 8 (288-row tile only) lives in VGPRs.
  f32x4 acc8[8];
  if constexpr (MT > 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc8[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto acc_get = [&](auto i_, auto j_, auto r_) -> float {
    constexpr int i = i_, j = j_, r = r_;
    if constexpr (i < 8) return acc_read<4 * (8 * i + j) + r>();
    else return acc8[j][r];
  };

  // ---- fragment read offsets (k-step kk toggles byte bit 6)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fq ^ (fr & 7)) << 4;
  const int aoff = (wr * HB + fr) * 128 + sw;                        // + (qa*16*MA0 + i*16) * 128
  const int boff = OP_BYTES + (wc * 128 + fr) * 128 + sw;            // + (qb*64 + jj*16) * 128

  bf16x8 fa0[MA0][2], fa1[MA1][2], fb0[2][4][2], fb1[4][2];
  // fragment read number r of an operand (r = 2 * fragment + kk)
  auto rd_a0 = [&](const char* base, auto r_) { constexpr int r = r_; fa0[r >> 1][r & 1] = *reinterpret_cast<const bf16x8*>(base + ((aoff + (r >> 1) * 2048) ^ ((r & 1) << 6))); };
  auto rd_a1 = [&](const char* base, auto r_) { constexpr int r = r_; fa1[r >> 1][r & 1] = *reinterpret_cast<const bf16x8*>(base + ((aoff + (16 * MA0 + (r >> 1) * 16) * 128) ^ ((r & 1) << 6))); };
  auto rd_b0 = [&](const char* base, auto set_, auto r_) { constexpr int r = r_, S = set_; fb0[S][r >> 1][r & 1] = *reinterpret_cast<const bf16x8*>(base + ((boff + (r >> 1) * 2048) ^ ((r & 1) << 6))); };
  auto rd_b1 = [&](const char* base, auto r_) { constexpr int r = r_; fb1[r >> 1][r & 1] = *reinterpret_cast<const bf16x8*>(base + ((boff + (64 + (r >> 1) * 16) * 128) ^ ((r & 1) << 6))); };

  // MFMA number m of quadrant (QA, QB): kk-major, then m-fragment, then n-fragment; S = the B0 register set in use
  auto mfma_one = [&](auto qa_, auto qb_, auto set_, auto m_) {
    constexpr int QA = qa_, QB = qb_, S = set_, m = m_;
    constexpr int MAq = QA ? MA1 : MA0;
    constexpr int kk = m / (4 * MAq), i = (m % (4 * MAq)) / 4, jj = m % 4;
    constexpr int I = QA * MA0 + i, J = QB * 4 + jj, R = 4 * (8 * I + J);
    const bf16x8& av = [&]() -> const bf16x8& { if constexpr (QA == 0) return fa0[i][kk]; else return fa1[i][kk]; }();
    const bf16x8& bv = [&]() -> const bf16x8& { if constexpr (QB == 0) return fb0[S][jj][kk]; else return fb1[jj][kk]; }();
    if constexpr (I < 8) mfma_acc<R>(bv, av);
    else mfma_vacc(acc8[J], bv, av);
  };
  // one phase: NM MFMAs with NF fillers spread evenly between them (filler f follows MFMA floor-spaced), order pinned
  auto phase = [&](auto qa_, auto qb_, auto set_, auto nf_, auto&& filler) {
    constexpr int QA = qa_, NF = nf_;
    constexpr int NM = 8 * (QA ? MA1 : MA0);
    sfor<NM>([&](auto m_) {
      constexpr int m = m_;
      mfma_one(qa_, qb_, set_, m_);
      constexpr int f_lo = (m * NF + NM - 1) / NM, f_hi = ((m + 1) * NF + NM - 1) / NM;   // ceil(m NF / NM) .. ceil((m+1) NF / NM)
      if constexpr (f_hi > f_lo) {
        __builtin_amdgcn_sched_barrier(0);
        sfor<f_hi - f_lo>([&](auto d_) { filler(std::integral_constant<int, f_lo + decltype(d_)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  auto begin_phase = [&](auto barrier_) {
    __builtin_amdgcn_s_waitcnt(0xC07F);                    /* lgkmcnt(0) */
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(barrier_)::value) {
#ifndef EXP_4W_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // K-tile t.  SET: the B0 register set holding B0 of t (the other one receives B0 of t+1).  STEADY: K-tile t + 2 exists:
  // no per-piece conditions inside the MFMA stream.
  auto ktile = [&](int t, auto set_, auto steady_) {
    constexpr int S = set_;
    constexpr bool STEADY = steady_;
    const char* base = smem + (t & 1) * KBUF_BYTES;
    const char* nbase = smem + ((t + 1) & 1) * KBUF_BYTES;
    // EXP_4W_* (tools/gemm4w_ablate.sh): timing ablations only, wrong results; the shipped library defines none of them
    auto dma = [&](int tt, auto c_, auto i2) {             // piece i2 of half-tile c of K-tile tt
#ifndef EXP_4W_NO_DMA
      if (STEADY || tt < nk) dma_piece(tt, c_, i2, Ab, Wb, cur.a_src, cur.b_src);
#endif
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
    // phase 0: barrier | (A0, B0) | reads B1 of t | DMA A-q0 of t+2
    begin_phase(std::true_type{});
    phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, set_, std::integral_constant<int, 8 + MA0>{}, [&](auto f_) {
      constexpr int f = f_;
      if constexpr (f < 8) { RD(rd_b1(base, f_)); }
      else dma(t + 2, C0{}, std::integral_constant<int, f - 8>{});
    });
    // phase 1: (A0, B1) | reads A1 of t | DMA B-q0 of t+2
    begin_phase(std::false_type{});
    phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, set_, std::integral_constant<int, 2 * MA1 + 4>{}, [&](auto f_) {
      constexpr int f = f_;
      if constexpr (f < 2 * MA1) { RD(rd_a1(base, f_)); }
      else dma(t + 2, C1{}, std::integral_constant<int, f - 2 * MA1>{});
    });
    // phase 2: retire K-tile t+1 (the two youngest half-tiles - A-q0, B-q0 of t+2 - stay in flight) | barrier | (A1, B1) |
    // reads A0 of t+1 | DMA B-q1 of t+2
    __builtin_amdgcn_s_waitcnt(0xC07F);                    /* lgkmcnt(0) */
    if (STEADY || t + 2 < nk) wait_vm(std::integral_constant<int, MA0 + 4>{});
    else __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */
    __builtin_amdgcn_sched_barrier(0);
#ifndef EXP_4W_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
    phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, set_, std::integral_constant<int, 2 * MA0 + 4>{}, [&](auto f_) {
      constexpr int f = f_;
      if constexpr (f < 2 * MA0) { RD(rd_a0(nbase, f_)); }
      else dma(t + 2, C2{}, std::integral_constant<int, f - 2 * MA0>{});
    });
    // phase 3: (A1, B0) | reads B0 of t+1 into the other register set | DMA A-q1 of t+2
    begin_phase(std::false_type{});
    phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, set_, std::integral_constant<int, 8 + MA1>{}, [&](auto f_) {
      constexpr int f = f_;
      if constexpr (f < 8) { RD(rd_b0(nbase, std::integral_constant<int, S ^ 1>{}, f_)); }
      else dma(t + 2, C3{}, std::integral_constant<int, f - 8>{});
    });
  };

  // ---- prologue: K-tiles 0 and 1 in flight, K-tile 0 retired, its A0 / B0 fragments read
  if (!prefetched) sfor<4>([&](auto c) { stage_half(0, c, Ab, Wb, cur.a_src, cur.b_src); });
  sfor<4>([&](auto c) { stage_half(1, c, Ab, Wb, cur.a_src, cur.b_src); });   // nk >= 2 (g2v_gemm_8p_supported)
  sfor<32 * (MT < 8 ? MT : 8)>([&](auto r) { acc_zero<decltype(r)::value>(); });
  wait_vm(std::integral_constant<int, MT + 8>{});          // K-tile 1's four half-tiles stay in flight
  __builtin_amdgcn_s_barrier();
  sfor<2 * MA0>([&](auto r) { rd_a0(smem, r); });
  sfor<8>([&](auto r) { rd_b0(smem, std::integral_constant<int, 0>{}, r); });
  __builtin_amdgcn_s_waitcnt(0xC07F);                      /* lgkmcnt(0) */

  {
    int t = 0;
    for (; t + 3 < nk; t += 2) {                           // t + 1 + 2 < nk: both K-tiles are steady
      ktile(t, std::integral_constant<int, 0>{}, std::true_type{});
      ktile(t + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    for (; t < nk; t += 2) {
      ktile(t, std::integral_constant<int, 0>{}, std::false_type{});
      if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                      /* lgkmcnt(0) */
  __builtin_amdgcn_s_barrier();                            // the epilogue reuses the LDS the last reads came from
  wait32();                                                // the last MFMAs' results are in their registers before v_accvgpr_read

  // ---- the next tile of this workgroup: its first K-tile goes into ring slot 0 now (every wave is past its last LDS
  // read), and lands while the epilogue below runs out of the LDS above slot 0
  const int vn = vt + gridDim.x;
  const bool has_next = vn < a.total;
  if (has_next) {
    setup_tile(vn, cur);                                   // m0 / n0 / g of the tile being finished are in locals
    const W4Group& gn_ = a.g[cur.gi];
    const char* An = uniform_ptr(reinterpret_cast<const char*>(gn_.A + (size_t)cur.m0 * a.lda));
    const char* Wn = uniform_ptr(reinterpret_cast<const char*>(gn_.W + (size_t)cur.n0 * K));
    sfor<4>([&](auto c) { stage_half(0, c, An, Wn, cur.a_src, cur.b_src); });
  }

#ifdef EXP_4W_NO_EPI
  __builtin_amdgcn_s_barrier();
  prefetched = has_next;
  continue;
#endif
  // ------------------------------------------------------------------ epilogue (gemm_8p.hip's, for 4 waves x 128 columns)
  // Two passes over the m-fragments (i < I0, then the rest), each: (1) every lane rounds its accumulators to the bf16 Linear
  // output (bias, activation) and writes them, 4 consecutive columns = 8 bytes at a time, into a row-major bf16 image in LDS;
  // (2) the block walks that image in 16-byte chunks, 32 (16 for SwiGLU) consecutive lanes per output row: every global
  // access - output store, residual load - is a full 16-byte lane access on 512 (256) contiguous bytes per row.
  constexpr bool SWI = EPI == G2V_EPI_SWIGLU;
  constexpr int PITCH = OUT_PITCH;
  constexpr int I0 = (MT + 1) / 2;                         // m-fragments per pass
  constexpr int CPR = SWI ? 16 : 32;                       // 16-byte chunks per output row
  constexpr int RPI = NT / CPR;                            // rows per sweep
  char* const img = smem + KBUF_BYTES;
  int etid = tid;
  asm volatile("" : "+v"(etid));
  const int efr = etid & 15, efq = (etid >> 4) & 3;
  const int ch = etid % CPR, r0 = etid / CPR;
  const int gn = (SWI ? (n0 >> 1) : n0) + ch * 8;          // first of this lane's 8 output columns
  const bool round_gamma = a.flags & G2V_GEMM_GAMMA_ROUND_BF16;
  sfor<2>([&](auto h_) {
    constexpr int h = h_;
    const int irow0 = wr * 16 * I0 + efr;
    if constexpr (SWI) {
      sfor<I0>([&](auto ii_) {
        constexpr int ii = ii_, i = h * I0 + ii;
        if constexpr (i < MT) {
          sfor<4>([&](auto jp_) {
            constexpr int jp = jp_;
            float o[4];
            sfor<4>([&](auto r_) {
              constexpr int r = r_;
              float gt = bfround(acc_get(std::integral_constant<int, i>{}, std::integral_constant<int, 2 * jp>{}, r_));
              float up = bfround(acc_get(std::integral_constant<int, i>{}, std::integral_constant<int, 2 * jp + 1>{}, r_));
              float sl = bfround(siluf_(gt));
              o[r] = sl * up;
            });
            *reinterpret_cast<u32x2*>(img + (irow0 + ii * 16) * PITCH + (wc * 64 + jp * 16 + efq * 4) * 2) =
                u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          });
        }
      });
    } else {
      sfor<8>([&](auto j_) {
        constexpr int j = j_;
        const int cl = wc * 128 + j * 16 + efq * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
          u32x2 bb = *reinterpret_cast<const u32x2*>(g.bias + n0 + cl);
          bv[0] = bits2f_lo(bb[0]); bv[1] = bits2f_hi(bb[0]); bv[2] = bits2f_lo(bb[1]); bv[3] = bits2f_hi(bb[1]);
        }
        sfor<I0>([&](auto ii_) {
          constexpr int ii = ii_, i = h * I0 + ii;
          if constexpr (i < MT) {
            float o[4];
            sfor<4>([&](auto r_) {
              constexpr int r = r_;
              float v = bfround(acc_get(std::integral_constant<int, i>{}, j_, r_) + bv[r]);
              if constexpr (EPI == G2V_EPI_GELU) v = gelu_fast(v);
              if constexpr (EPI == G2V_EPI_QUICKGELU) {
                float u = bfround(1.702f * v);
                float sg = bfround(sigmoidf_(u));
                v = v * sg;
              }
              o[r] = v;
            });
            *reinterpret_cast<u32x2*>(img + (irow0 + ii * 16) * PITCH + cl * 2) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          }
        });
      });
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
    constexpr int SB = NIT % 5 == 0 ? 5 : (NIT % 4 == 0 ? 4 : (NIT % 3 == 0 ? 3 : 2));
    static_assert(NIT % SB == 0, "sweep batches");
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
  });
  prefetched = has_next;
  }
}

template <int EPI, int MA0, int MA1>
int launch_h(const W4Args& a, int total, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w_kernel<EPI, MA0, MA1>), hipFuncAttributeMaxDynamicSharedMemorySize,
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
  W4Args b = a;
  b.total = total;
  hipLaunchKernelGGL((gemm4w_kernel<EPI, MA0, MA1>), dim3(total < n_cu ? total : n_cu), dim3(NT), LDS_BYTES, s, b);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

template <int EPI>
int launch(const W4Args& a, int bm, int total, hipStream_t s) {
  if (bm == 288) return launch_h<EPI, 5, 4>(a, total, s);
#ifndef G2V_4W_FEW                                           // G2V_4W_FEW: tests/test_build_cpu.py audits a subset of the instantiations
  if (bm == 256) return launch_h<EPI, 4, 4>(a, total, s);
  if (bm == 224) return launch_h<EPI, 4, 3>(a, total, s);
  if (bm == 192) return launch_h<EPI, 4, 2>(a, total, s);
#endif
  if (bm == 160) return launch_h<EPI, 3, 2>(a, total, s);
  return launch_h<EPI, 2, 2>(a, total, s);
}

}  // namespace

// Same eligibility as gemm_8p.hip (g2v_gemm_8p_supported); the tile height is chosen there and handed over.
int g2v_gemm_4w_launch(const g2v_gemm_desc* d, int bm, const int* order, hipStream_t s) {
  W4Args a;
  a.ngroups = 0; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldres = d->ldres;
  a.tiles_n = d->N / BN; a.flags = d->flags;
  a.sn = a.tiles_n < 8 ? a.tiles_n : 8;
  a.sm = 32 / a.sn > 1 ? 32 / a.sn : 1;
  int total = 0;
  for (int i = 0; i < d->ngroups; ++i) {
    const g2v_gemm_group& sg = d->g[order[i]];
    if (sg.M <= 0) continue;
    W4Group& g = a.g[a.ngroups++];
    g.A = (const __bf16*)sg.A; g.W = (const __bf16*)sg.W; g.bias = (const __bf16*)sg.bias; g.C = sg.C; g.res = sg.res;
    g.gamma = (const float*)sg.gamma; g.M = sg.M; g.tile_start = total;
    total += ((sg.M + bm - 1) / bm) * a.tiles_n;
  }
  if (a.ngroups == 1) a.g[1] = a.g[0];
  if (total == 0) return G2V_OK;
  switch (d->epilogue) {
#ifndef G2V_4W_FEW
    case G2V_EPI_BF16: return launch<G2V_EPI_BF16>(a, bm, total, s);
    case G2V_EPI_GELU: return launch<G2V_EPI_GELU>(a, bm, total, s);
    case G2V_EPI_QUICKGELU: return launch<G2V_EPI_QUICKGELU>(a, bm, total, s);
    case G2V_EPI_RES_BF16: return launch<G2V_EPI_RES_BF16>(a, bm, total, s);
#endif
    case G2V_EPI_SWIGLU: return launch<G2V_EPI_SWIGLU>(a, bm, total, s);
    case G2V_EPI_RES_F32: return launch<G2V_EPI_RES_F32>(a, bm, total, s);
    default: return G2V_ERR_ARG;
  }
}

// Large-tile bf16 GEMM for the prefill shapes: (bm x 256 x 64) tile, 8 waves (2 M x 4 N), operands
// DMA'd global -> LDS with global_load_lds_dwordx4 (no VGPR staging), two LDS slots: tile k+1 is in flight
// while tile k is multiplied, one raw s_barrier per K-step (MI355X guide §5.5 T3 "minimum 2-phase").  Same
// math, epilogues and grouping as gemm.hip.  Measured (tools/bench_kernels.py): wins over the 128x128 kernel
// only on long-K, narrow-N shapes (down-proj 0.93 PF, decoder fc2 0.95 PF), where its higher flop/byte matters;
// a 3-slot ring with smaller tiles was L2-bound (87 % TCC miss) and is not kept.
//
// The row-tile height bm = 32*MT (MT = 1..9 m-tiles per wave, a template parameter) is chosen on the host so that the number of
// tiles lands just under a whole number of 256-CU rounds (a fixed tile height loses up to half the chip
// on the last round at M = 10 968).
//
// LDS image per operand tile: rows of 64 bf16 (128 B); one DMA wave-instruction fills 8 rows (1 KiB,
// lane-linear), so the bank-conflict swizzle (16-B chunk ^= row & 7) is applied to the per-lane SOURCE
// address and again on the ds_read_b128 fragment reads (guide rule 21).
#include "common.h"
#include "g2vlm_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int BN = 256, BK = 64, MAX_MT = 9, STAGES = 2;  // per-wave m-tiles; bm <= 2*9*16 = 288
constexpr int BM_MAX = 2 * MAX_MT * 16;
constexpr int A_BYTES = BM_MAX * 128;                    // 36 KiB
constexpr int B_BYTES = BN * 128;                        // 32 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;           // 68 KiB, x2 = 136 KiB (<= 160 KiB/CU)

struct BigGroup {
  const __bf16* A; const __bf16* W; const __bf16* bias; void* C; const void* res; const float* gamma;
  int M, tile_start, bm, pad;
};
struct BigArgs {
  BigGroup g[2];
  int ngroups, N, K, lda, ldc, ldres, tiles_n, flags, first_tile, sm, sn;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// wait until at most `keep` of this wave's DMA instructions are still in flight (keep is wave-uniform, 0..7)
__device__ __forceinline__ void wait_vm(int keep) {
  switch (keep) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  }
}

template <int EPI, int MT>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(BigArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware bijective remap (blocks sharing bid % 8 share an L2): contiguous tile ranges per XCD
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  }
  bid += a.first_tile;
  const int gi = (a.ngroups > 1 && bid >= a.g[1].tile_start) ? 1 : 0;
  const BigGroup g = a.g[gi];
  const int t = bid - g.tile_start;
  int tm, tn;
  if (a.sm <= 1) {
    tm = t / a.tiles_n; tn = t - tm * a.tiles_n;
  } else {
    // walk the tile grid supertile by supertile (sm x sn tiles, partial ones at the right/bottom edge): the ~32
    // tiles an XCD runs concurrently then share sm A-slabs and sn W-slabs per K-step in its 4 MiB L2
    const int tiles_m = (g.M + g.bm - 1) / g.bm;
    const int row_sz = a.sm * a.tiles_n;
    int sup_m = t / row_sz, r = t - sup_m * row_sz;
    int h = min(a.sm, tiles_m - sup_m * a.sm);             // tiles in this supertile row
    // within a supertile row, tiles are ordered supertile-column by supertile-column, each h x width
    int full_w = a.sn * h;
    int sup_n = r / full_w, p = r - sup_n * full_w;
    int width = min(a.sn, a.tiles_n - sup_n * a.sn);
    tm = sup_m * a.sm + p / width;
    tn = sup_n * a.sn + p % width;
  }
  const int bm = g.bm, m0 = tm * bm, n0 = tn * BN;
  const int M = g.M, N = a.N, K = a.K;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int row0 = wm * MT * 16;                           // bm == 32 * MT: both M-waves own MT m-tiles

  // ---- DMA source pointers.  group q of 8 rows is filled by wave (q & 7), instruction (q >> 3)
  const int a_groups = bm >> 3;                            // <= 20
  const int srow = lane >> 3, scp = lane & 7;
  const __bf16* asrc[5];
  const __bf16* bsrc[4];
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    int q = it * 8 + w;
    int row = q * 8 + srow;
    int gr = min(m0 + row, M - 1);
    asrc[it] = g.A + (size_t)gr * a.lda + ((scp ^ (row & 7)) << 3);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    int q = it * 8 + w;
    int row = q * 8 + srow;
    int gr = min(n0 + row, N - 1);
    bsrc[it] = g.W + (size_t)gr * K + ((scp ^ (row & 7)) << 3);
  }
  auto stage = [&](int slot, int k0) {
    char* sA = smem + slot * STAGE_BYTES;
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      int q = it * 8 + w;
      if (q < a_groups)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[it] + k0), (lds_ptr_t)(sA + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int q = it * 8 + w;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[it] + k0), (lds_ptr_t)(sB + q * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets for k-step 0 (k-step 1: chunk ^= 4  ->  byte ^= 64)
  const int fr = lane & 15, fq = lane >> 4;
  int aoff0;
  {
    int r = row0 + fr;                                     // + 16*i: (r & 7) unchanged
    aoff0 = r * 128 + ((fq ^ (r & 7)) << 4);
  }
  int boff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = wn * 64 + j * 16 + fr;
    boff[j] = c * 128 + ((fq ^ (c & 7)) << 4);
  }

  const int nk = K / BK;
  stage(0, 0);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt (issued one step ago) must have landed for every wave; own LDS reads of step kt-1 are complete
    // (lgkmcnt) before the barrier, so the other slot is free for the next DMA right after it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) stage(slot ^ 1, (kt + 1) * BK);
    const char* sA = smem + slot * STAGE_BYTES;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fb[4], fa[MT];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sB + (boff[j] ^ (kk << 6)));
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + ((aoff0 + i * 2048) ^ (kk << 6)));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    slot ^= 1;
  }

  // ------------------------------------------------------------------ epilogue
  const int rbase = m0 + row0 + (fq << 2);
  if constexpr (EPI == G2V_EPI_SWIGLU) {
    __bf16* C = reinterpret_cast<__bf16*>(g.C);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        int oc = ((n0 + wn * 64) >> 1) + jp * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int gm = rbase + i * 16 + r;
          if (gm >= M) continue;
          float gt = bfround(acc[i][2 * jp][r]);
          float up = bfround(acc[i][2 * jp + 1][r]);
          float s = bfround(siluf_(gt));
          C[(size_t)gm * a.ldc + oc] = f2bf(s * up);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int gn = n0 + wn * 64 + j * 16 + fr;
      float bv = g.bias ? bf2f(g.bias[gn]) : 0.f;
      float gam = (EPI == G2V_EPI_RES_F32 && g.gamma) ? g.gamma[gn] : 1.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int gm = rbase + i * 16 + r;
          if (gm >= M) continue;
          float v = bfround(acc[i][j][r] + bv);
          size_t o = (size_t)gm * a.ldc + gn;
          if constexpr (EPI == G2V_EPI_BF16) {
            reinterpret_cast<__bf16*>(g.C)[o] = f2bf(v);
          } else if constexpr (EPI == G2V_EPI_GELU) {
            reinterpret_cast<__bf16*>(g.C)[o] = f2bf(gelu_erf(v));
          } else if constexpr (EPI == G2V_EPI_QUICKGELU) {
            float u = bfround(1.702f * v);
            float s = bfround(sigmoidf_(u));
            reinterpret_cast<__bf16*>(g.C)[o] = f2bf(v * s);
          } else if constexpr (EPI == G2V_EPI_RES_F32) {
            if (g.gamma) {
              v = __fmul_rn(v, gam);
              if (a.flags & G2V_GEMM_GAMMA_ROUND_BF16) v = bfround(v);
            }
            float rv = g.res ? reinterpret_cast<const float*>(g.res)[(size_t)gm * a.ldres + gn] : 0.f;
            reinterpret_cast<float*>(g.C)[o] = __fadd_rn(rv, v);
          } else if constexpr (EPI == G2V_EPI_RES_BF16) {
            float rv = bf2f(reinterpret_cast<const __bf16*>(g.res)[(size_t)gm * a.ldres + gn]);
            reinterpret_cast<__bf16*>(g.C)[o] = f2bf(rv + v);
          }
        }
      }
    }
  }
}

inline int ceil32(int x) { return (x + 31) & ~31; }

template <int EPI, int MT>
int launch_mt(const BigArgs& a, int first, int count, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<EPI, MT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            STAGES * STAGE_BYTES) != hipSuccess) return G2V_ERR_LAUNCH;
    attr_set = true;
  }
  BigArgs b = a;
  b.first_tile = first;
  hipLaunchKernelGGL((gemm_big_kernel<EPI, MT>), dim3(count), dim3(512), STAGES * STAGE_BYTES, s, b);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// ONE launch for both groups with the large group's tile height: the small (und) group's few tiles are mostly
// padding rows, but they run on CUs the round leaves idle anyway, instead of a second latency-bound launch.
template <int EPI>
int launch(const BigArgs& a, int total, hipStream_t s) {
  switch (a.g[0].bm / 32) {
    case 1: return launch_mt<EPI, 1>(a, 0, total, s);
    case 2: return launch_mt<EPI, 2>(a, 0, total, s);
    case 3: return launch_mt<EPI, 3>(a, 0, total, s);
    case 4: return launch_mt<EPI, 4>(a, 0, total, s);
    case 5: return launch_mt<EPI, 5>(a, 0, total, s);
    case 6: return launch_mt<EPI, 6>(a, 0, total, s);
    case 7: return launch_mt<EPI, 7>(a, 0, total, s);
    case 8: return launch_mt<EPI, 8>(a, 0, total, s);
    default: return launch_mt<EPI, 9>(a, 0, total, s);
  }
}

}  // namespace

bool g2v_gemm_big_eligible(const g2v_gemm_desc* d) {
  if ((d->K % BK) || (d->N % BN) || (d->lda & 7)) return false;
  long rows = 0;
  for (int i = 0; i < d->ngroups; ++i) rows += d->g[i].M;
  // measured crossover (tools/bench_kernels.py): long K, narrow N (down-proj, fc2); elsewhere the 128x128 kernel wins
  return rows >= 256 && ((d->K >= 4096 && d->N <= 2048) || (d->K >= 1536 && d->N <= 2048));
}

int g2v_gemm_big_launch(const g2v_gemm_desc* d, hipStream_t s) {
  BigArgs a;
  a.ngroups = 0; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldres = d->ldres;
  a.tiles_n = d->N / BN; a.flags = d->flags;
  a.sn = a.tiles_n < 8 ? a.tiles_n : 8;
  a.sm = (d->flags & G2V_GEMM_SUPERTILE) ? (32 / a.sn > 1 ? 32 / a.sn : 1) : 1;
  // order groups large-first so the big group's tiles start every XCD range
  int order[2] = {0, 1};
  if (d->ngroups == 2 && d->g[1].M > d->g[0].M) { order[0] = 1; order[1] = 0; }
  int Ms[2] = {0, 0};
  for (int i = 0; i < d->ngroups; ++i) Ms[i] = d->g[order[i]].M;
  // the small group uses the same tile height; with M1 << bm it costs ceil(M1/bm) (usually 1) row of tiles
  int tiles1 = 0;
  if (d->ngroups == 2 && Ms[1] > 0) tiles1 = ((Ms[1] + BM_MAX - 1) / BM_MAX) * a.tiles_n;
  // big group: pick (rounds R, bm) minimising R * effective tile height, tiles <= R * 256 CUs
  int best_bm = BM_MAX; long best_cost = -1;
  for (int R = 1; R <= 256; ++R) {
    long budget = ((long)R * 256 - tiles1) / a.tiles_n;
    if (budget < 1) continue;
    int bm = ceil32((int)((Ms[0] + budget - 1) / budget));
    if (bm > BM_MAX) continue;
    long cost = (long)R * bm;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_bm = bm; }
    if (best_cost >= 0 && (long)R * 32 > best_cost) break;
  }
  int total = 0;
  for (int i = 0; i < d->ngroups; ++i) {
    const g2v_gemm_group& sg = d->g[order[i]];
    if (sg.M <= 0) continue;
    BigGroup& g = a.g[a.ngroups++];
    g.A = (const __bf16*)sg.A; g.W = (const __bf16*)sg.W; g.bias = (const __bf16*)sg.bias; g.C = sg.C; g.res = sg.res;
    g.gamma = (const float*)sg.gamma; g.M = sg.M; g.tile_start = total; g.pad = 0;
    g.bm = best_bm;
    total += ((sg.M + g.bm - 1) / g.bm) * a.tiles_n;
  }
  if (a.ngroups == 1) a.g[1] = a.g[0];
  if (total == 0) return G2V_OK;
  switch (d->epilogue) {
    case G2V_EPI_BF16: return launch<G2V_EPI_BF16>(a, total, s);
    case G2V_EPI_GELU: return launch<G2V_EPI_GELU>(a, total, s);
    case G2V_EPI_QUICKGELU: return launch<G2V_EPI_QUICKGELU>(a, total, s);
    case G2V_EPI_SWIGLU: return launch<G2V_EPI_SWIGLU>(a, total, s);
    case G2V_EPI_RES_F32: return launch<G2V_EPI_RES_F32>(a, total, s);
    case G2V_EPI_RES_BF16: return launch<G2V_EPI_RES_BF16>(a, total, s);
    default: return G2V_ERR_ARG;
  }
}

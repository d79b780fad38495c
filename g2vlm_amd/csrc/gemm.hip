// bf16 MFMA GEMM with fused epilogues, and an exact-fp32 MFMA GEMM for the fp32 head islands.
//
//   C[M,N] = epilogue( A[M,K] (bf16, row stride lda) x W[N,K]^T (bf16, nn.Linear layout) + bias[N] )
//
// Both operands are K-contiguous, so both LDS tiles are [rows][64 bf16] with the 16-byte-chunk XOR
// swizzle (chunk ^= row & 7) that makes the ds_read_b128 fragment reads of mfma_f32_16x16x32_bf16
// conflict-free (MI355X guide T2).  Block = 256 threads = 4 waves (2x2), tile 128x128x64, each wave
// 64x64 = 4x4 MFMA tiles.  Global->LDS staging goes through registers with the issue-early /
// write-late split (guide T14): tile t+1 is in flight while tile t is multiplied.
//
// "Grouped" launch: up to two problems that share N, K and epilogue (the und / geo experts of one
// MoT layer, reference modeling/g2vlm/qwen2vl.py:584-606, 655-658, 894-903) run in one grid.
//
// dtype flow replicated from the reference's autocast(bf16): bias is bf16, the
// Linear result is rounded to bf16 BEFORE any activation / layer-scale / residual.
#include "common.h"
#include "g2vlm_hip.h"
#include "gemm_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand

struct GemmGroup {
  const __bf16* A;
  const __bf16* W;
  const __bf16* bias;
  void* C;
  const void* res;
  const float* gamma;
  int M;
  int tile_start;
};

struct GemmArgs {
  GemmGroup g[2];
  int ngroups, N, K, lda, ldc, ldres, tiles_n, flags, sm, sn, g0_tiles;
};

__device__ __forceinline__ u32x4 ld_chunk(const __bf16* base, int row, int ld, int k, int K) {
  u32x4 z = {0, 0, 0, 0};
  if (k >= K) return z;
  return *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + k);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];
  char* sA = smem;
  char* sB = smem + TILE_BYTES;

  int bid = blockIdx.x;
  int gi = (a.ngroups > 1 && bid >= a.g[1].tile_start) ? 1 : 0;
  const GemmGroup g = a.g[gi];
  int t = bid - g.tile_start;
  int tm, tn;
  if (a.sm > 1 && gi == 0) {
    // XCD-aware order for the large group: blocks b, b+8, ... share an XCD (4 MiB L2); give each XCD a contiguous
    // range of the tile sequence and walk it in sm x sn supertiles so concurrent tiles share A/W slabs in that L2
    const int nt = a.g0_tiles;
    int xcd = t & 7, qn = nt >> 3, rn = nt & 7;
    t = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (t >> 3);
    const int tiles_m = (g.M + BM - 1) / BM;
    const int row_sz = a.sm * a.tiles_n;
    int sup_m = t / row_sz, r = t - sup_m * row_sz;
    int h = min(a.sm, tiles_m - sup_m * a.sm);
    int full_w = a.sn * h;
    int sup_n = r / full_w, p = r - sup_n * full_w;
    int width = min(a.sn, a.tiles_n - sup_n * a.sn);
    tm = sup_m * a.sm + p / width;
    tn = sup_n * a.sn + p % width;
  } else {
    tm = t / a.tiles_n; tn = t - tm * a.tiles_n;
  }
  int m0 = tm * BM, n0 = tn * BN;
  const int M = g.M, N = a.N, K = a.K;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;

  // staging map: 4 chunks of A and 4 of W per thread; 8 consecutive threads cover one 128-B row
  int srow[4], sc[4], arow[4], brow[4], soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int id = tid + 256 * i;
    srow[i] = id >> 3;
    sc[i] = id & 7;
    arow[i] = min(m0 + srow[i], M - 1);
    brow[i] = min(n0 + srow[i], N - 1);
    soff[i] = srow[i] * 128 + ((sc[i] ^ (srow[i] & 7)) << 4);
  }
  u32x4 ra[4], rb[4];
  auto stage_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = ld_chunk(g.A, arow[i], a.lda, k0 + sc[i] * 8, K);
      rb[i] = ld_chunk(g.W, brow[i], K, k0 + sc[i] * 8, K);
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(sA + soff[i]) = ra[i];
      *reinterpret_cast<u32x4*>(sB + soff[i]) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) for k-step 0; k-step 1 flips chunk bit 2 (chunk += 4)
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = wm * 64 + i * 16 + (lane & 15);
    aoff[i] = r * 128 + ((((lane >> 4)) ^ (r & 7)) << 4);
    int c = wn * 64 + i * 16 + (lane & 15);
    boff[i] = c * 128 + ((((lane >> 4)) ^ (c & 7)) << 4);
  }

  const int nk = (K + BK - 1) / BK;
  stage_load(0);
  stage_write();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage_load((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(sA + (aoff[i] ^ (kk << 6)));
        fb[i] = *reinterpret_cast<const bf16x8*>(sB + (boff[i] ^ (kk << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      stage_write();
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ epilogue
  const int rbase = m0 + wm * 64 + ((lane >> 4) << 2);
  const int cl = lane & 15;
  if constexpr (EPI == G2V_EPI_SWIGLU) {
    // W rows interleave gate/up in blocks of 16 output columns: MFMA tile j even = gate, odd = up
    __bf16* C = reinterpret_cast<__bf16*>(g.C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        int oc = ((n0 + wn * 64) >> 1) + jp * 16 + cl;
        if (oc >= (N >> 1)) continue;
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
    return;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int gn = n0 + wn * 64 + j * 16 + cl;
      if (gn >= N) continue;
      float bv = g.bias ? bf2f(g.bias[gn]) : 0.f;
      float gam = (EPI == G2V_EPI_RES_F32 && g.gamma) ? g.gamma[gn] : 1.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int gm = rbase + i * 16 + r;
          if (gm >= M) continue;
          float v = bfround(acc[i][j][r] + bv);   // the Linear's bf16 output
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
            // out(fp32) = res(fp32) + [bf16](linear * gamma)
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

// ------------------------------------------------------------------------------------ fp32 GEMM
// C = act(A[M,K] x W[N,K]^T + bias) (+ res), everything fp32, products accumulated by
// v_mfma_f32_16x16x4_f32 = an exact k-ordered fp32 fma chain (guide §3 "FP32-input MFMA").
// Tile 128x128x16; LDS holds both operands k-major ([16][144] floats) so the per-lane fragment
// reads (row = lane&15, k = lane>>4) are bank-conflict-free.
constexpr int FBK = 16, FLD = 144;

struct GemmF32Args {
  const float* A; const float* W; const float* bias; float* C; const float* res;
  int M, N, K, lda, ldc, ldres, tiles_n, relu;
};

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmF32Args a) {
  __shared__ __attribute__((aligned(16))) float sA[FBK * FLD];
  __shared__ __attribute__((aligned(16))) float sB[FBK * FLD];
  int tm = blockIdx.x / a.tiles_n, tn = blockIdx.x - tm * a.tiles_n;
  int m0 = tm * 128, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  // staging: 128 rows x 16 floats = 512 float4 chunks per operand -> 2 per thread
  int srow[2], sk[2], ar[2], br[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int id = tid + 256 * i;
    srow[i] = id >> 2; sk[i] = (id & 3) * 4;
    ar[i] = min(m0 + srow[i], a.M - 1);
    br[i] = min(n0 + srow[i], a.N - 1);
  }
  f32x4 ra[2], rb[2];
  auto ld4 = [&](const float* base, int row, int ld, int k) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* p = base + (size_t)row * ld + k;
    if (k + 3 < a.K) v = *reinterpret_cast<const f32x4*>(p);
    else { for (int e = 0; e < 4; ++e) if (k + e < a.K) v[e] = p[e]; }
    return v;
  };
  auto stage_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { ra[i] = ld4(a.A, ar[i], a.lda, k0 + sk[i]); rb[i] = ld4(a.W, br[i], a.K, k0 + sk[i]); }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { sA[(sk[i] + e) * FLD + srow[i]] = ra[i][e]; sB[(sk[i] + e) * FLD + srow[i]] = rb[i][e]; }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (a.K + FBK - 1) / FBK;
  stage_load(0); stage_write(); __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage_load((kt + 1) * FBK);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float fa[4], fb[4];
      int kq = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = sA[kq * FLD + wm * 64 + i * 16 + (lane & 15)];
        fb[i] = sB[kq * FLD + wn * 64 + i * 16 + (lane & 15)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) { stage_write(); __syncthreads(); }
  }
  const int rbase = m0 + wm * 64 + ((lane >> 4) << 2), cl = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int gn = n0 + wn * 64 + j * 16 + cl;
    if (gn >= a.N) continue;
    float bv = a.bias ? a.bias[gn] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int gm = rbase + i * 16 + r;
        if (gm >= a.M) continue;
        float v = acc[i][j][r] + bv;
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.res) v = a.res[(size_t)gm * a.ldres + gn] + v;
        a.C[(size_t)gm * a.ldc + gn] = v;
      }
  }
}

template <int EPI>
int launch_bf16(const GemmArgs& a, int total_tiles, hipStream_t s) {
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(total_tiles), dim3(256), 0, s, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

}  // namespace

extern "C" int g2v_gemm_bf16(const g2v_gemm_desc* d, void* stream) {
  if (!d || d->ngroups < 1 || d->ngroups > 2 || d->N <= 0 || d->K <= 0 || (d->K & 7) || (d->lda & 7)) return G2V_ERR_ARG;
  if (d->epilogue == G2V_EPI_SWIGLU && (d->N & 31)) return G2V_ERR_ARG;
  const int force = d->flags & (G2V_GEMM_FORCE_SMALL_TILE | G2V_GEMM_FORCE_BIG_TILE | G2V_GEMM_FORCE_8P);
  if (force == 0 && g2v_gemm_skinny_eligible(d)) {
    for (int i = 0; i < d->ngroups; ++i)
      if (d->g[i].M < 0 || !d->g[i].W || !d->g[i].C || (d->g[i].M > 0 && !d->g[i].A)) return G2V_ERR_ARG;
    if (d->epilogue == G2V_EPI_RES_BF16)
      for (int i = 0; i < d->ngroups; ++i) if (d->g[i].M > 0 && !d->g[i].res) return G2V_ERR_ARG;
    return g2v_gemm_skinny_launch(d, (hipStream_t)stream);
  }
  if (g2v_gemm_8p_supported(d) && (force == G2V_GEMM_FORCE_8P || (force == 0 && g2v_gemm_8p_preferred(d)))) {
    for (int i = 0; i < d->ngroups; ++i)
      if (d->g[i].M < 0 || !d->g[i].W || !d->g[i].C || (d->g[i].M > 0 && !d->g[i].A)) return G2V_ERR_ARG;
    if (d->epilogue == G2V_EPI_RES_BF16)
      for (int i = 0; i < d->ngroups; ++i) if (!d->g[i].res) return G2V_ERR_ARG;
    return g2v_gemm_8p_launch(d, (hipStream_t)stream);
  }
  if (!(d->flags & G2V_GEMM_FORCE_SMALL_TILE) && ((d->flags & G2V_GEMM_FORCE_BIG_TILE) ? (d->K % 64 == 0 && d->N % 256 == 0) : g2v_gemm_big_eligible(d))) {
    for (int i = 0; i < d->ngroups; ++i)
      if (d->g[i].M < 0 || !d->g[i].W || !d->g[i].C || (d->g[i].M > 0 && !d->g[i].A)) return G2V_ERR_ARG;
    return g2v_gemm_big_launch(d, (hipStream_t)stream);
  }
  GemmArgs a;
  a.ngroups = d->ngroups; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldres = d->ldres;
  a.tiles_n = (d->N + BN - 1) / BN; a.flags = d->flags;
  int total = 0;
  for (int i = 0; i < 2; ++i) {
    const g2v_gemm_group& s = d->g[i < d->ngroups ? i : 0];
    GemmGroup& g = a.g[i];
    g.A = (const __bf16*)s.A; g.W = (const __bf16*)s.W; g.bias = (const __bf16*)s.bias; g.C = s.C;
    g.res = s.res; g.gamma = (const float*)s.gamma; g.M = s.M; g.tile_start = total;
    if (i < d->ngroups) {
      if (s.M < 0 || !s.W || !s.C || (s.M > 0 && !s.A)) return G2V_ERR_ARG;
      if (d->epilogue == G2V_EPI_RES_BF16 && !s.res) return G2V_ERR_ARG;
      total += ((s.M + BM - 1) / BM) * a.tiles_n;
    }
  }
  a.sm = 1; a.sn = 1;
  // 96 resident tiles per XCD (3 blocks x 32 CUs): +3..8 % over row-major order on the C3 shapes (profiles/)
  a.sn = a.tiles_n < 12 ? a.tiles_n : 12; a.sm = 96 / a.sn > 1 ? 96 / a.sn : 1;
  if (a.ngroups == 2 && a.g[1].M == 0) a.ngroups = 1;
  if (a.ngroups == 2 && a.g[0].M == 0) { a.g[0] = a.g[1]; a.g[0].tile_start = 0; a.ngroups = 1; }
  if (total == 0) return G2V_OK;
  a.g0_tiles = a.ngroups > 1 ? a.g[1].tile_start : total;
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case G2V_EPI_BF16: return launch_bf16<G2V_EPI_BF16>(a, total, s);
    case G2V_EPI_GELU: return launch_bf16<G2V_EPI_GELU>(a, total, s);
    case G2V_EPI_QUICKGELU: return launch_bf16<G2V_EPI_QUICKGELU>(a, total, s);
    case G2V_EPI_SWIGLU: return launch_bf16<G2V_EPI_SWIGLU>(a, total, s);
    case G2V_EPI_RES_F32: return launch_bf16<G2V_EPI_RES_F32>(a, total, s);
    case G2V_EPI_RES_BF16: return launch_bf16<G2V_EPI_RES_BF16>(a, total, s);
    default: return G2V_ERR_ARG;
  }
}

extern "C" int g2v_gemm_f32(const void* A, const void* W, const void* bias, void* C, const void* res,
                            int M, int N, int K, int lda, int ldc, int ldres, int relu, void* stream) {
  if (!A || !W || !C || M < 0 || N <= 0 || K <= 0 || (lda & 3) || (K & 3)) return G2V_ERR_ARG;
  if (M == 0) return G2V_OK;
  GemmF32Args a{(const float*)A, (const float*)W, (const float*)bias, (float*)C, (const float*)res,
                M, N, K, lda, ldc, ldres, (N + 127) / 128, relu};
  int tiles = ((M + 127) / 128) * a.tiles_n;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

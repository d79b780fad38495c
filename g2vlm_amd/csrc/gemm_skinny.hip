// Skinny-M bf16 GEMM (M <= 64 rows: the text-prefix prefill of G2VLM.recon, reference g2vlm.py:701-733, and short
// prompt chunks of chat_with_recon): C[M,N] = epilogue(x[M,K] . W[N,K]^T).  With so few rows the Linear is a
// weight-streaming problem (HBM-bound, 2*N*K bytes), not an MFMA one: the tiled kernels would put 8 rows into
// 128/256-row tiles on N/128 CUs and take 30-160 us per launch where the weight stream needs 1-12 us.
//
// One wave = one 16-column group of W x one K-slice; the S waves of a workgroup split K and are summed through LDS in
// slice order (deterministic).  A wave streams its W rows straight into MFMA B fragments (guide: "GEMV / M <= 16 decode
// weights: load straight to VGPRs, deep unroll, late vmcnt"): per 64-deep k-step a lane loads 32 contiguous bytes of
// its row, four lanes cover one 128-byte line, and the two 16x16x32 MFMAs of the step take the two 16-byte halves -
// the k index inside a step is permuted identically for x and W, which a dot product does not see.  x (<= 64 x K bf16)
// is re-read by every wave from L2.
#include "common.h"
#include "g2vlm_hip.h"
#include "gemm_internal.h"

namespace {

struct SkArgs {
  const __bf16* A; const __bf16* W; const __bf16* bias; void* C; const void* res; const float* gamma;
  int M, N, K, lda, ldc, ldres, flags, S;
};

template <int EPI, int MTS>
__global__ __launch_bounds__(1024) void gemm_skinny_kernel(SkArgs a) {
  constexpr int CG = EPI == G2V_EPI_SWIGLU ? 2 : 1;       // column groups per workgroup (gate + up for SwiGLU)
  extern __shared__ __attribute__((aligned(16))) float red[];   // [S][CG][MTS][256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w % CG, s = w / CG;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = (blockIdx.x * CG + cg) * 16;
  const int nks = a.K >> 6;                                // 64-deep k-steps
  const int ks0 = (int)((long)s * nks / a.S), ks1 = (int)((long)(s + 1) * nks / a.S);

  const __bf16* wp = a.W + (size_t)(n0 + fr) * a.K + fq * 16;
  const __bf16* xp[MTS];
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt) xp[mt] = a.A + (size_t)min(mt * 16 + fr, a.M - 1) * a.lda + fq * 16;

  f32x4 acc[MTS];
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int U = MTS == 1 ? 4 : 2;                      // k-steps in flight per wave
  int ks = ks0;
  for (; ks + U <= ks1; ks += U) {
    bf16x8 wb[U][2], xa[U][MTS][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (ks + u) << 6;
      wb[u][0] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k));
      wb[u][1] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k + 8));
#pragma unroll
      for (int mt = 0; mt < MTS; ++mt) {
        xa[u][mt][0] = *reinterpret_cast<const bf16x8*>(xp[mt] + k);
        xa[u][mt][1] = *reinterpret_cast<const bf16x8*>(xp[mt] + k + 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[u][mt][h], wb[u][h], acc[mt], 0, 0, 0);
  }
  for (; ks < ks1; ++ks) {
    const int k = ks << 6;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 wv = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + k + 8 * h));
#pragma unroll
      for (int mt = 0; mt < MTS; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(xp[mt] + k + 8 * h), wv, acc[mt], 0, 0, 0);
    }
  }

  // ---- K-slice partials -> LDS; register r of lane (fr, fq) is C[m = mt*16 + fq*4 + r][n = n0 + fr]
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((s * CG + cg) * MTS + mt) * 256 + (fq * 4 + r) * 16 + fr] = acc[mt][r];
  __syncthreads();

  // ---- one thread per output element: sum the slices in order, then the Linear's epilogue (as gemm.hip)
  constexpr int OUT = MTS * 256;                           // outputs per column group
  for (int e = tid; e < OUT; e += blockDim.x) {
    const int mt = e >> 8, idx = e & 255, m = mt * 16 + (idx >> 4), nl = idx & 15;
    if (m >= a.M) continue;
    float v0 = 0.f, v1 = 0.f;
    for (int ss = 0; ss < a.S; ++ss) {
      v0 += red[((ss * CG + 0) * MTS + mt) * 256 + idx];
      if constexpr (CG == 2) v1 += red[((ss * CG + 1) * MTS + mt) * 256 + idx];
    }
    if constexpr (EPI == G2V_EPI_SWIGLU) {
      const int oc = blockIdx.x * 16 + nl;
      float gt = bfround(v0), up = bfround(v1);
      float sl = bfround(siluf_(gt));
      reinterpret_cast<__bf16*>(a.C)[(size_t)m * a.ldc + oc] = f2bf(sl * up);
    } else {
      const int gn = blockIdx.x * 16 + nl;
      float v = bfround(v0 + (a.bias ? bf2f(a.bias[gn]) : 0.f));
      const size_t o = (size_t)m * a.ldc + gn;
      if constexpr (EPI == G2V_EPI_BF16) {
        reinterpret_cast<__bf16*>(a.C)[o] = f2bf(v);
      } else if constexpr (EPI == G2V_EPI_GELU) {
        reinterpret_cast<__bf16*>(a.C)[o] = f2bf(gelu_fast(v));
      } else if constexpr (EPI == G2V_EPI_QUICKGELU) {
        float u = bfround(1.702f * v);
        float sg = bfround(sigmoidf_(u));
        reinterpret_cast<__bf16*>(a.C)[o] = f2bf(v * sg);
      } else if constexpr (EPI == G2V_EPI_RES_F32) {
        if (a.gamma) {
          v = __fmul_rn(v, a.gamma[gn]);
          if (a.flags & G2V_GEMM_GAMMA_ROUND_BF16) v = bfround(v);
        }
        float rv = a.res ? reinterpret_cast<const float*>(a.res)[(size_t)m * a.ldres + gn] : 0.f;
        reinterpret_cast<float*>(a.C)[o] = __fadd_rn(rv, v);
      } else if constexpr (EPI == G2V_EPI_RES_BF16) {
        float rv = bf2f(reinterpret_cast<const __bf16*>(a.res)[(size_t)m * a.ldres + gn]);
        reinterpret_cast<__bf16*>(a.C)[o] = f2bf(rv + v);
      }
    }
  }
}

template <int EPI, int MTS>
int launch_sk(const SkArgs& a, hipStream_t s) {
  constexpr int CG = EPI == G2V_EPI_SWIGLU ? 2 : 1;
  const int blocks = a.N / (16 * CG);
  const size_t lds = (size_t)a.S * CG * MTS * 256 * sizeof(float);
  hipLaunchKernelGGL((gemm_skinny_kernel<EPI, MTS>), dim3(blocks), dim3(64 * CG * a.S), lds, s, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

template <int EPI>
int launch_epi(const SkArgs& a, hipStream_t s) {
  if (a.M <= 16) return launch_sk<EPI, 1>(a, s);
  if (a.M <= 32) return launch_sk<EPI, 2>(a, s);
  return launch_sk<EPI, 4>(a, s);
}

}  // namespace

bool g2v_gemm_skinny_eligible(const g2v_gemm_desc* d) {
  int live = 0, M = 0;
  for (int i = 0; i < d->ngroups; ++i)
    if (d->g[i].M > 0) { ++live; M = d->g[i].M; }
  if (live != 1 || M > 64) return false;
  if ((d->K & 63) || (d->lda & 7) || (d->N & (d->epilogue == G2V_EPI_SWIGLU ? 31 : 15))) return false;
  for (int i = 0; i < d->ngroups; ++i)
    if (d->g[i].M > 0 && ((reinterpret_cast<uintptr_t>(d->g[i].A) | reinterpret_cast<uintptr_t>(d->g[i].W)) & 15)) return false;
  return true;
}

int g2v_gemm_skinny_launch(const g2v_gemm_desc* d, hipStream_t s) {
  const g2v_gemm_group* sg = nullptr;
  for (int i = 0; i < d->ngroups; ++i)
    if (d->g[i].M > 0) sg = &d->g[i];
  if (!sg) return G2V_OK;
  SkArgs a{(const __bf16*)sg->A, (const __bf16*)sg->W, (const __bf16*)sg->bias, sg->C, sg->res, (const float*)sg->gamma,
           sg->M, d->N, d->K, d->lda, d->ldc, d->ldres, d->flags, 1};
  // K-slices per column group: enough waves to keep ~8 per CU streaming, each slice at least 2 k-steps, <= 16 waves / workgroup
  const int cg = d->epilogue == G2V_EPI_SWIGLU ? 2 : 1;
  const int groups = d->N / 16, nks = d->K / 64;
  int S = (2048 + groups - 1) / groups;
  if (S > 16 / cg) S = 16 / cg;
  if (S > nks / 2) S = nks / 2;
  if (S < 1) S = 1;
  a.S = S;
  switch (d->epilogue) {
    case G2V_EPI_BF16: return launch_epi<G2V_EPI_BF16>(a, s);
    case G2V_EPI_GELU: return launch_epi<G2V_EPI_GELU>(a, s);
    case G2V_EPI_QUICKGELU: return launch_epi<G2V_EPI_QUICKGELU>(a, s);
    case G2V_EPI_SWIGLU: return launch_epi<G2V_EPI_SWIGLU>(a, s);
    case G2V_EPI_RES_F32: return launch_epi<G2V_EPI_RES_F32>(a, s);
    case G2V_EPI_RES_BF16: return launch_epi<G2V_EPI_RES_BF16>(a, s);
    default: return G2V_ERR_ARG;
  }
}

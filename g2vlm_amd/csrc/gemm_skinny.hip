// Skinny-M bf16 GEMM (M <= 64 rows: the text-prefix prefill of G2VLM.recon, reference g2vlm.py:701-733, short prompt
// chunks of chat_with_recon, and every Linear of the batched decode step): C[M,N] = epilogue(x[M,K] . W[N,K]^T).  With so
// few rows the Linear is a weight-streaming problem (HBM-bound, 2*N*K bytes), not an MFMA one: the tiled kernels would
// put 8 rows into 128/256-row tiles on N/128 CUs and take 30-160 us per launch where the weight stream needs 1-12 us.
//
// One wave = one 16-column group of W x one K-slice.  A wave streams its W rows straight into MFMA B fragments (guide:
// "GEMV / M <= 16 decode weights: load straight to VGPRs, deep unroll, late vmcnt"): per 64-deep k-step a lane loads 32
// contiguous bytes of its row, four lanes cover one 128-byte line, and the two 16x16x32 MFMAs of the step take the two
// 16-byte halves - the k index inside a step is permuted identically for x and W, which a dot product does not see.
// x (<= 64 x K bf16) is re-read by every wave from L2.
//
// The launch is latency-bound, not bandwidth-bound, unless (a) a wave's whole K-slice is in flight at once - slices are
// cut to <= U k-steps so the loop is ONE trip of U x 32-byte loads per lane - and (b) there are enough waves to cover
// the chip even when N is small (o / down projections: 96 column groups).  So K is split S ways inside a workgroup
// (summed through LDS) and KS ways across workgroups (grid.y): every workgroup writes its fp32 partial tile to the
// caller's workspace, takes a ticket on its column group, and the last one to arrive sums the KS partials in slice order
// (deterministic) and applies the Linear's epilogue.  Tickets reset themselves.
#include "common.h"
#include "g2vlm_hip.h"
#include "gemm_internal.h"

namespace {

struct SkArgs {
  const __bf16* A; const __bf16* W; const __bf16* bias; void* C; const void* res; const float* gamma;
  float* ws; int* tickets;
  int M, N, K, lda, ldc, ldres, flags, S, KS;
};

template <int EPI, int MTS>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkArgs a) {
  constexpr int CG = EPI == G2V_EPI_SWIGLU ? 2 : 1;       // column groups per workgroup (gate + up for SwiGLU)
  constexpr int U = MTS == 1 ? 6 : (MTS == 2 ? 4 : 2);     // k-steps in flight per wave (one trip when the slice fits)
  extern __shared__ __attribute__((aligned(16))) float red[];   // [S][CG][MTS][256]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w % CG, s = w / CG;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = (blockIdx.x * CG + cg) * 16;
  const int nks = a.K >> 6;                                // 64-deep k-steps
  const int T = a.S * a.KS, t = blockIdx.y * a.S + s;      // this wave's K-slice of T
  const int ks0 = (int)((long)t * nks / T), ks1 = (int)((long)(t + 1) * nks / T);

  const __bf16* wp = a.W + (size_t)(n0 + fr) * a.K + fq * 16;
  const __bf16* xp[MTS];
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt) xp[mt] = a.A + (size_t)min(mt * 16 + fr, a.M - 1) * a.lda + fq * 16;

  f32x4 acc[MTS];
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // epilogue operands (bias, residual, layer scale) of this thread's first PF output elements, fetched now: read after
  // the reduction they would add one more memory round trip to a kernel that is a chain of three or four
  constexpr int OUT = MTS * 256;                           // outputs per column group
  constexpr int PF = 2;
  float pf_bias[PF], pf_res[PF], pf_gam[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    pf_bias[i] = 0.f; pf_res[i] = 0.f; pf_gam[i] = 1.f;
    const int e = tid + i * (int)blockDim.x;
    const int m = (e >> 8) * 16 + ((e & 255) >> 4), gn = blockIdx.x * 16 + (e & 15);
    if (e < OUT && m < a.M) {
      if constexpr (EPI != G2V_EPI_SWIGLU) {
        if (a.bias) pf_bias[i] = bf2f(a.bias[gn]);
      }
      if constexpr (EPI == G2V_EPI_RES_F32) {
        if (a.gamma) pf_gam[i] = a.gamma[gn];
        if (a.res) pf_res[i] = reinterpret_cast<const float*>(a.res)[(size_t)m * a.ldres + gn];
      }
      if constexpr (EPI == G2V_EPI_RES_BF16) pf_res[i] = bf2f(reinterpret_cast<const __bf16*>(a.res)[(size_t)m * a.ldres + gn]);
    }
  }

  for (int ks = ks0; ks < ks1; ks += U) {
    bf16x8 wb[U][2], xa[U][MTS][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(ks + u, ks1 - 1) << 6;             // past the slice: a valid address, the MFMA is skipped
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
      if (ks + u < ks1) {                                  // wave-uniform
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int mt = 0; mt < MTS; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[u][mt][h], wb[u][h], acc[mt], 0, 0, 0);
      }
  }

  // ---- K-slice partials -> LDS; register r of lane (fr, fq) is C[m = mt*16 + fq*4 + r][n = n0 + fr]
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((s * CG + cg) * MTS + mt) * 256 + (fq * 4 + r) * 16 + fr] = acc[mt][r];
  __syncthreads();

  // the workgroup's own slices, in order
  auto wg_sum = [&](int e, float& v0, float& v1) {
    const int mt = e >> 8, idx = e & 255;
    v0 = v1 = 0.f;
    for (int ss = 0; ss < a.S; ++ss) {
      v0 += red[((ss * CG + 0) * MTS + mt) * 256 + idx];
      if constexpr (CG == 2) v1 += red[((ss * CG + 1) * MTS + mt) * 256 + idx];
    }
  };
  if (a.KS > 1) {
    // ---- cross-workgroup slices: partial tile -> workspace, ticket, the last arrival finishes
    float* mine = a.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (CG * OUT);
    for (int e = tid; e < OUT; e += blockDim.x) {
      float v0, v1;
      wg_sum(e, v0, v1);
      // agent-scope atomic stores / loads (sc1: write-through / L2-bypassing across the 8 XCDs) carry the partials; no
      // release / acquire fence is used - a fence here is a whole-L2 writeback + invalidate per workgroup (measured:
      // 34 us per launch instead of 8) - the stores are simply complete (vmcnt(0)) before the ticket is taken
      __hip_atomic_store(mine + e, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (CG == 2) __hip_atomic_store(mine + OUT + e, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(a.tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.KS - 1;
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(a.tickets + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- one thread per output element: sum the slices in order, then the Linear's epilogue (as gemm.hip)
  int it = 0;
  for (int e = tid; e < OUT; e += blockDim.x, ++it) {
    const int mt = e >> 8, idx = e & 255, m = mt * 16 + (idx >> 4), nl = idx & 15;
    if (m >= a.M) continue;
    float v0 = 0.f, v1 = 0.f;
    if (a.KS > 1) {
      for (int y = 0; y < a.KS; ++y) {
        const float* p = a.ws + ((size_t)y * gridDim.x + blockIdx.x) * (CG * OUT);
        v0 += __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (CG == 2) v1 += __hip_atomic_load(p + OUT + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      wg_sum(e, v0, v1);
    }
    if constexpr (EPI == G2V_EPI_SWIGLU) {
      const int oc = blockIdx.x * 16 + nl;
      float gt = bfround(v0), up = bfround(v1);
      float sl = bfround(siluf_(gt));
      reinterpret_cast<__bf16*>(a.C)[(size_t)m * a.ldc + oc] = f2bf(sl * up);
    } else {
      const int gn = blockIdx.x * 16 + nl;
      float bias_v, res_v = 0.f, gam_v = 1.f;
      if (it < PF) {
        bias_v = it == 0 ? pf_bias[0] : pf_bias[1]; res_v = it == 0 ? pf_res[0] : pf_res[1]; gam_v = it == 0 ? pf_gam[0] : pf_gam[1];
      } else {
        bias_v = a.bias ? bf2f(a.bias[gn]) : 0.f;
        if constexpr (EPI == G2V_EPI_RES_F32) {
          if (a.gamma) gam_v = a.gamma[gn];
          if (a.res) res_v = reinterpret_cast<const float*>(a.res)[(size_t)m * a.ldres + gn];
        }
        if constexpr (EPI == G2V_EPI_RES_BF16) res_v = bf2f(reinterpret_cast<const __bf16*>(a.res)[(size_t)m * a.ldres + gn]);
      }
      float v = bfround(v0 + bias_v);
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
          v = __fmul_rn(v, gam_v);
          if (a.flags & G2V_GEMM_GAMMA_ROUND_BF16) v = bfround(v);
        }
        reinterpret_cast<float*>(a.C)[o] = __fadd_rn(res_v, v);
      } else if constexpr (EPI == G2V_EPI_RES_BF16) {
        reinterpret_cast<__bf16*>(a.C)[o] = f2bf(res_v + v);
      }
    }
  }
}

template <int EPI, int MTS>
int launch_sk(const SkArgs& a, hipStream_t s) {
  constexpr int CG = EPI == G2V_EPI_SWIGLU ? 2 : 1;
  const int blocks = a.N / (16 * CG);
  const size_t lds = (size_t)a.S * CG * MTS * 256 * sizeof(float);
  hipLaunchKernelGGL((gemm_skinny_kernel<EPI, MTS>), dim3(blocks, a.KS), dim3(64 * CG * a.S), lds, s, a);
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
           nullptr, nullptr, sg->M, d->N, d->K, d->lda, d->ldc, d->ldres, d->flags, 1, 1};
  // K-slices: T = S (inside a workgroup, <= 4 per column group) x KS (across workgroups).  Enough slices that a wave's
  // slice is one trip of U k-steps, and enough waves (~1024) to cover the chip when N is small; at least 2 k-steps each.
  const int cg = d->epilogue == G2V_EPI_SWIGLU ? 2 : 1;
  const int mts = sg->M <= 16 ? 1 : (sg->M <= 32 ? 2 : 4);
  const int U = mts == 1 ? 6 : (mts == 2 ? 4 : 2);
  const int groups = d->N / 16, blocks = groups / cg, nks = d->K / 64;
  int T = (nks + U - 1) / U;
  const int fill = (1024 + groups - 1) / groups;
  // the cross-workgroup stage costs a store -> ticket -> load chain (~3 us): worth it only for long K (down projection,
  // 13 vs 20 us); short-K Linears keep all slices in one workgroup (qkv / o: 8 vs 8-10 us)
  const bool cross = nks >= 64;
  if (cross && T < fill) T = fill;
  if (T > nks / 2) T = nks / 2;
  if (T < 1) T = 1;
  const int smax = cross ? 4 : 8 / cg;
  int S = T < smax ? T : smax;
  int KS = cross ? (T + S - 1) / S : 1;
  // the cross-workgroup stage needs the caller's workspace: tickets (int32 per block, zeroed once) then fp32 partials
  // (the ticket area has a FIXED size: partials of one shape must never land where another shape keeps its tickets)
  constexpr size_t tick_bytes = 64 << 10;
  if ((size_t)blocks * 4 > tick_bytes) KS = 1;
  const size_t part_bytes = (size_t)blocks * cg * mts * 256 * 4;
  if (cross) {
    const size_t avail = d->workspace && d->workspace_bytes > (int64_t)tick_bytes ? (size_t)d->workspace_bytes - tick_bytes : 0;
    const int fit = (int)(avail / part_bytes);
    if (KS > fit) KS = fit;
    if (KS <= 1) {                                         // no (or too small a) workspace: all slices inside the workgroup
      KS = 1;
      S = T < 16 / cg ? T : 16 / cg;
      if (64 * cg * S > 512) S = 512 / (64 * cg);
    } else {
      a.tickets = reinterpret_cast<int*>(d->workspace);
      a.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->workspace) + tick_bytes);
    }
  }
  a.S = S; a.KS = KS;
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

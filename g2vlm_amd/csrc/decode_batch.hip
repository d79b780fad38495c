// Batched decode step (B = 2..8 scenes, one token each): the persistent-grid GEMVs of decode_layer.hip with B activation
// vectors per weight pass.
//
// The reference decodes one scene at a time (modeling/g2vlm/g2vlm.py:1086-1135, B == 1 asserted at :1006 / :1137); the
// batched step streams the 3.09 GB of weights ONCE for all scenes.  Its Linears used to go through the skinny MFMA GEMM
// behind separate RMSNorm launches (8 kernels per layer, 70 us per layer at B = 2 against 40 us at B = 1); here they are
// the batch-1 kernels' structure - 256 blocks, every wave an equal contiguous share of the rows, all loads of a batch
// issued before the first FMA, the norm / SwiGLU / residual fused - with the weight registers reused for B dot products:
//   * gemv_pgb_kernel (K <= 1536: qkv, o, gate/up, lm_head): a wave covers the whole K of its rows; the B activation rows
//     are normalised ONCE per block (wave b takes row b: one memory round trip for all rows) into an LDS strip and every
//     wave reads its fragments from there; the o projection reads its bf16 rows straight from global memory.
//   * gemv_pgk_kernel (K > 1536: down, K = 8960): the K axis is cut over the 8 waves of a block (1120 elements each), a
//     block owns ceil(N / 256) rows, partial dot products meet in LDS and are summed in wave order.  The activation
//     fragment of a wave is then 12 registers per scene instead of 72, which is what lets B = 8 fit.
// Results equal the batch-1 kernels' up to the order of the fp32 partial sums (gemv_pgk splits K by wave).
#include "common.h"
#include "decode_util.h"
#include "g2vlm_hip.h"

namespace {

// XMODE 0: x bf16 [B, K].  XMODE 1: x fp32 [B, K] residual stream, Qwen2RMSNorm(norm_w, eps) applied per row on the fly.
// ACT: gate/up rows interleaved per 16 (weights.interleave_gate_up), out bf16 [B, N / 2] = bf16(bf16(silu(g)) * u).
// Otherwise out bf16 [B, N] = bf16(W x + bias), or res fp32 [B, N] += that.
template <int XMODE, bool ACT, int NB, int RB>
__global__ __launch_bounds__(512) void gemv_pgb_kernel(const void* xin, const float* norm_w, float eps, const __bf16* W,
                                                       const __bf16* bias, __bf16* out, float* res, int B, int N, int K, int uq, int ur) {
  constexpr int KCH = 3, ROWS = ACT ? 2 * RB : RB;
  static_assert(NB * RB <= 32, "the batch's values are reduced together");
  __shared__ __attribute__((aligned(16))) uint32_t sx[XMODE == 1 ? NB * 768 : 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gw = blockIdx.x * 8 + w;
  const int lo = gw * uq + min(gw, ur), hi = lo + uq + (gw < ur ? 1 : 0);   // may be empty: the wave still stages its x row
  const int nch = K >> 3;
  const int No = ACT ? N / 2 : N;
  auto row_of = [&](int u, int half) { return ACT ? 32 * (u >> 4) + (u & 15) + 16 * half : u; };
  // the NB x RB values of a batch are reduced together (reduce_transpose): lane l ends up with value number l >> SH = lb RB + lr
  constexpr int NV = NB * RB, VP = g2v_pow2_ge(NV), SH = 6 - g2v_log2(VP);
  const int vi = lane >> SH;
  const bool rep = (lane & ((1 << SH) - 1)) == 0 && vi < NV;
  const int lb = vi / RB, lr = vi - lb * RB;               // the (scene, unit) this lane finishes

  u32x4 ww[ROWS][KCH];
  unsigned short bnext = 0;
  float rnext = 0.f;
  auto issue = [&](int u0) {
    const int nrow = min(RB, hi - u0);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (r < nrow) {
#pragma unroll
        for (int h = 0; h < (ACT ? 2 : 1); ++h) {
          const u32x4* wp = reinterpret_cast<const u32x4*>(W + (size_t)row_of(u0 + r, h) * K);
#pragma unroll
          for (int j = 0; j < KCH; ++j) ww[(ACT ? 2 * r + h : r)][j] = __builtin_nontemporal_load(wp + min(lane + 64 * j, nch - 1));
        }
      }
    }
    if constexpr (!ACT) {
      const int n = min(u0 + lr, hi - 1), b = min(lb, B - 1);
      if (bias) bnext = reinterpret_cast<const unsigned short*>(bias)[n];
      if (res) rnext = res[(size_t)b * N + n];
    }
  };

  uint32_t xp[NB][KCH][4];
  if constexpr (XMODE == 0) {
    u32x4 xv[NB][KCH];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
        xv[b][j] = reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(xin) + (size_t)min(b, B - 1) * K)[min(lane + 64 * j, nch - 1)];
    if (lo < hi) issue(lo);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) xp[b][j][e] = lane + 64 * j < nch ? xv[b][j][e] : 0u;
  } else {
    // wave b normalises row b into the strip (rows past B: the last row again, never stored)
    const float* xf = reinterpret_cast<const float*>(xin) + (size_t)min(w, B - 1) * K;
    f32x4 a[KCH][2], nwv[KCH][2];
    if (w < NB) {
#pragma unroll
      for (int j = 0; j < KCH; ++j) {
        const int c = min(lane + 64 * j, nch - 1);
        a[j][0] = *reinterpret_cast<const f32x4*>(xf + 8 * c);
        a[j][1] = *reinterpret_cast<const f32x4*>(xf + 8 * c + 4);
        nwv[j][0] = *reinterpret_cast<const f32x4*>(norm_w + 8 * c);
        nwv[j][1] = *reinterpret_cast<const f32x4*>(norm_w + 8 * c + 4);
      }
    }
    if (lo < hi) issue(lo);
    if (w < NB) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < KCH; ++j) {
        if (lane + 64 * j < nch) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ss += a[j][0][e] * a[j][0][e] + a[j][1][e] * a[j][1][e];
        }
      }
      ss = wave_sum_dpp(ss);
      const float rstd = 1.0f / sqrtf(ss / (float)K + eps);
#pragma unroll
      for (int j = 0; j < KCH; ++j) {
        if (lane + 64 * j < nch) {
          u32x4 p;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            p[e] = pack_bf16x2(__fmul_rn(nwv[j][0][2 * e], __fmul_rn(a[j][0][2 * e], rstd)),
                               __fmul_rn(nwv[j][0][2 * e + 1], __fmul_rn(a[j][0][2 * e + 1], rstd)));
            p[2 + e] = pack_bf16x2(__fmul_rn(nwv[j][1][2 * e], __fmul_rn(a[j][1][2 * e], rstd)),
                                   __fmul_rn(nwv[j][1][2 * e + 1], __fmul_rn(a[j][1][2 * e + 1], rstd)));
          }
          *reinterpret_cast<u32x4*>(&sx[w * 768 + 4 * (lane + 64 * j)]) = p;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j) {
        const u32x4 v = lane + 64 * j < nch ? *reinterpret_cast<const u32x4*>(&sx[b * 768 + 4 * (lane + 64 * j)]) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) xp[b][j][e] = v[e];
      }
  }

  for (int u0 = lo; u0 < hi; u0 += RB) {
    const int nrow = min(RB, hi - u0);
    const float bcur = __uint_as_float((uint32_t)bnext << 16), rcur = rnext;
    float acc[NB][ROWS];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[b][r] = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if ((ACT ? r / 2 : r) < nrow) {
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b][r] = dot2(ww[r][j][e], xp[b][j][e], acc[b][r]);
      }
    }
    if (u0 + RB < hi) issue(u0 + RB);                        // next batch in flight under this batch's reductions
    float v;
    {
      float va[VP], vb[ACT ? VP : 1];
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        va[i] = i < NV ? acc[i / RB][ACT ? 2 * (i % RB) : i % RB] : 0.f;
        if constexpr (ACT) vb[i] = i < NV ? acc[i / RB][2 * (i % RB) + 1] : 0.f;
      }
      v = reduce_transpose<VP>(va, lane);
      if constexpr (ACT) {
        const float u = reduce_transpose<VP>(vb, lane);
        v = bfround(siluf_(bfround(v))) * bfround(u);
      }
    }
    if (rep && lb < B && lr < nrow) {
      const int n = u0 + lr;
      if constexpr (ACT) {
        out[(size_t)lb * No + n] = f2bf(v);
      } else {
        v = bfround(v + bcur);
        if (res) res[(size_t)lb * N + n] = rcur + v;
        else out[(size_t)lb * N + n] = f2bf(v);
      }
    }
  }
}

// Long-K form: out / res [B, N] from x bf16 [B, K], W [N, K]; block `blk` owns rows [blk per, (blk + 1) per), wave w the
// 16-byte chunks [w CW, (w + 1) CW) of K (CW <= 192).
template <int NB>
__global__ __launch_bounds__(512) void gemv_pgk_kernel(const __bf16* x, const __bf16* W, const __bf16* bias, __bf16* out, float* res,
                                                       int B, int N, int K, int per, int CW) {
  constexpr int R = 6, KCH = 3, G = NB < 4 ? NB : 4;       // scenes per accumulation group
  __shared__ float part[8][NB][R];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
  const int nch = K >> 3;
  const int c0 = w * CW, cend = min(c0 + CW, nch);
  const int row_lo = blockIdx.x * per, row_hi = min(row_lo + per, N);
  uint32_t xp[NB][KCH][4];
  u32x4 ww[R][KCH];
  auto issue = [&](int r0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u32x4* wp = reinterpret_cast<const u32x4*>(W + (size_t)min(r0 + r, N - 1) * K);
#pragma unroll
      for (int j = 0; j < KCH; ++j) ww[r][j] = __builtin_nontemporal_load(wp + min(c0 + lane + 64 * j, nch - 1));
    }
  };
  {
    u32x4 xv[NB][KCH];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j) xv[b][j] = reinterpret_cast<const u32x4*>(x + (size_t)min(b, B - 1) * K)[min(c0 + lane + 64 * j, nch - 1)];
    if (row_lo < row_hi) issue(row_lo);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) xp[b][j][e] = c0 + lane + 64 * j < cend ? xv[b][j][e] : 0u;
  }
  for (int r0 = row_lo; r0 < row_hi; r0 += R) {
#pragma unroll
    for (int g0 = 0; g0 < NB; g0 += G) {
      float acc[G][R];
#pragma unroll
      for (int b = 0; b < G; ++b)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[b][r] = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < G; ++b) acc[b][r] = dot2(ww[r][j][e], xp[g0 + b][j][e], acc[b][r]);
      if (g0 + G >= NB && r0 + R < row_hi) issue(r0 + R);   // the weights are free after the last group's dots
      {
        constexpr int NV = G * R, VP = g2v_pow2_ge(NV), SH = 6 - g2v_log2(VP);
        float va[VP];
#pragma unroll
        for (int i = 0; i < VP; ++i) va[i] = i < NV ? acc[i / R][i % R] : 0.f;
        const float s = reduce_transpose<VP>(va, lane);
        const int vi = lane >> SH;
        if ((lane & ((1 << SH) - 1)) == 0 && vi < NV) part[w][g0 + vi / R][vi - (vi / R) * R] = s;
      }
    }
    __syncthreads();
    if (tid < NB * R) {
      const int b = tid / R, r = tid - b * R, n = r0 + r;
      if (b < B && n < row_hi) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += part[k][b][r];
        const float v = bfround(s + (bias ? bf2f(bias[n]) : 0.f));
        if (res) res[(size_t)b * N + n] += v;
        else out[(size_t)b * N + n] = f2bf(v);
      }
    }
    __syncthreads();
  }
}

template <int XMODE, bool ACT, int NB>
int launch_pgb(int rb, hipStream_t s, const void* x, const float* nw, float eps, const __bf16* W, const __bf16* bias, __bf16* out, float* res,
               int B, int N, int K, int uq, int ur) {
#define G2V_PGB(RB_)                                                                                                         \
  hipLaunchKernelGGL((gemv_pgb_kernel<XMODE, ACT, NB, RB_>), dim3(256), dim3(512), 0, s, x, nw, eps, W, bias, out, res, B, N, K, uq, ur)
  if constexpr (NB == 8) {
    if (rb <= 1) G2V_PGB(1); else G2V_PGB(2);
  } else if constexpr (NB == 4) {
    if (rb <= 1) G2V_PGB(1); else if (rb <= 2) G2V_PGB(2); else G2V_PGB(3);
  } else {
    if (rb <= 1) G2V_PGB(1); else if (rb <= 2) G2V_PGB(2); else if (rb <= 3) G2V_PGB(3); else G2V_PGB(5);
  }
#undef G2V_PGB
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

template <int NB>
int dispatch_pgb(bool norm, bool act, int rb, hipStream_t s, const void* x, const float* nw, float eps, const __bf16* W, const __bf16* bias,
                 __bf16* out, float* res, int B, int N, int K, int uq, int ur) {
  if (norm && act) return launch_pgb<1, true, NB>(rb, s, x, nw, eps, W, bias, out, res, B, N, K, uq, ur);
  if (norm) return launch_pgb<1, false, NB>(rb, s, x, nw, eps, W, bias, out, res, B, N, K, uq, ur);
  return launch_pgb<0, false, NB>(rb, s, x, nw, eps, W, bias, out, res, B, N, K, uq, ur);
}

}  // namespace

// Y[B, N] = X[B, K] . W[N, K]^T for B = 1..8 decode rows, with the fused forms of g2v_gemv_pg (same argument meaning; x, out
// and res are row-major with leading dimensions K, N (N / 2 for act) and N).  The grid is 256 blocks of 8 waves whatever N.
extern "C" int g2v_gemv_pg_batch(const void* x, const void* norm_w, float eps, const void* W, const void* bias, void* out, void* res,
                                 int B, int N, int K, int act, void* stream) {
  if (!x || !W || (!out && !res) || B <= 0 || B > 8 || N <= 0 || K <= 0 || (K & 7) || K > 12288) return G2V_ERR_ARG;
  if (act && ((N & 31) || !out || res || !norm_w)) return G2V_ERR_ARG;
  const int nch = K / 8;
  if (norm_w && nch > 192) return G2V_ERR_ARG;               // the fused norm stages whole rows: hidden-size K
  hipStream_t s = (hipStream_t)stream;
  const __bf16 *Wp = (const __bf16*)W, *bp = (const __bf16*)bias;
  const int nb = B <= 2 ? 2 : (B <= 4 ? 4 : 8);
  if (nch > 192) {
    const int per = (N + 255) / 256, CW = (nch + 7) / 8;
    if (nb == 2) hipLaunchKernelGGL(gemv_pgk_kernel<2>, dim3(256), dim3(512), 0, s, (const __bf16*)x, Wp, bp, (__bf16*)out, (float*)res, B, N, K, per, CW);
    else if (nb == 4) hipLaunchKernelGGL(gemv_pgk_kernel<4>, dim3(256), dim3(512), 0, s, (const __bf16*)x, Wp, bp, (__bf16*)out, (float*)res, B, N, K, per, CW);
    else hipLaunchKernelGGL(gemv_pgk_kernel<8>, dim3(256), dim3(512), 0, s, (const __bf16*)x, Wp, bp, (__bf16*)out, (float*)res, B, N, K, per, CW);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  const int U = act ? N / 2 : N, waves = 256 * 8;
  const int uq = U / waves, ur = U % waves;
  const int per_wave = uq + (ur ? 1 : 0);
  const int rb_cap = nb == 8 ? 2 : (nb == 4 ? 3 : 5);
  int rb = per_wave;
  if (rb > rb_cap) {                                         // several equal batches rather than a full one and a remainder
    const int nbat = (per_wave + rb_cap - 1) / rb_cap;
    rb = (per_wave + nbat - 1) / nbat;
  }
  if (nb == 2 && rb == 4) rb = 5;                            // instantiated batch sizes: 1, 2, 3, 5
  const float* nwp = (const float*)norm_w;
  if (nb == 2) return dispatch_pgb<2>(norm_w != nullptr, act != 0, rb, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, B, N, K, uq, ur);
  if (nb == 4) return dispatch_pgb<4>(norm_w != nullptr, act != 0, rb, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, B, N, K, uq, ur);
  return dispatch_pgb<8>(norm_w != nullptr, act != 0, rb, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, B, N, K, uq, ur);
}

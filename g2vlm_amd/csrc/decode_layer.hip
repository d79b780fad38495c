// Batch-1 decode step, second generation: every kernel is sized to the chip instead of to the problem.
//
// The step (reference modeling/g2vlm/g2vlm.py:1086-1135: 28 x {q/k/v/o/gate/up/down Linear at M = 1, flash_attn_varlen_func
// with q_len 1} + lm_head) is HBM-bound: 3.09 GB of weights + 28 672 B x Lk of K/V per token.  One CU can pull about
// 1/256 of the HBM rate (MI355X guide: ~10 B/clk/CU), so a kernel streams at the full rate only while ALL 256 CUs hold an
// equal share of its bytes.  The first-generation kernels (decode.hip) size their grids by the problem - 86 workgroups for
// the split-KV attention at Lk = 11 k, 2240 / 768 / 1024 row blocks for the GEMVs - and run at 0.6-3.6 TB/s.  Here:
//   * GEMV: a persistent grid of 256 blocks; the N rows are cut into 256 x NW equal contiguous shares, one per wave.
//     A wave covers the whole K of its rows (lane l takes 16-byte chunks l, l + 64, ...): no LDS, no barrier, the
//     activation vector (normalised on the fly for the fused RMSNorm form) lives in registers for the wave's lifetime.
//   * attention: 256 blocks per scene, each an equal share of ONE kv head's keys; the 4 waves of a block merge their
//     online-softmax partials through LDS, so the combine reads 128 partials per head instead of 172.
// Tried and dropped (profiles/r02d_*): pulling the next Linear's weights into the 256 MiB Infinity Cache under the
// latency-bound kernels.  From the Infinity Cache the 55 MB gate/up GEMV takes 10.4 us instead of 12.1 (a bare read of the
// 55 MB takes 9.9): these kernels are bound by launch ramp and per-CU load throughput, not by HBM, and a prefetch branch in the
// step's graph serialises (+0.6 ms per token).
#include "common.h"
#include "decode_util.h"
#include "decode_attn_pg.h"
#include "g2vlm_hip.h"

#ifdef G2V_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
extern "C" int g2v_debug_stamps(void* buf) { g_stamp_buf = (unsigned long long*)buf; return 0; }
#endif

namespace {

// ---- persistent GEMV ---------------------------------------------------------------------------------------------------
// XMODE 0: x bf16[K].   XMODE 1: x = bf16(w_norm * (res * rsqrt(mean(res^2) + eps)))  (Qwen2RMSNorm of the fp32 residual stream).
// ACT: W is a gate/up matrix interleaved per 16 rows (weights.interleave_gate_up); a unit is the pair (gate row, up row) of
//      one output and the kernel writes bf16(bf16(silu(g)) * u).  Otherwise a unit is one row: out[n] = bf16(W[n].x + bias)
//      or res[n] += that.
// KCH = ceil(K / 512) chunk steps per lane; RB = units per batch (all of a batch's loads are issued before the first FMA).
// The activation is kept packed (bf16 pairs) and multiplied with v_dot2c_f32_bf16: 8 FLOP-pairs per 16-byte load cost 4
// instructions, so even K = 8960 (18 chunk steps, 72 registers of x) leaves the wave waiting on memory, not on the VALU.
template <int XMODE, bool ACT, int KCH, int RB>
__global__ __launch_bounds__(512) void gemv_pg_kernel(const void* xin, const float* norm_w, float eps, const __bf16* W,
                                                      const __bf16* bias, __bf16* out, float* res, int N, int K, int uq, int ur G2V_STAMP_ARG) {
  constexpr int ROWS = ACT ? 2 * RB : RB;
  G2V_STAMP_RT(10);
  G2V_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int nwb = blockDim.x >> 6;
  // wave gw of nw takes units [gw uq + min(gw, ur), +uq (+1 if gw < ur)): U = nw uq + ur, split by the host (no division here)
  const int gw = blockIdx.x * nwb + (threadIdx.x >> 6);
  const int lo = gw * uq + min(gw, ur), hi = lo + uq + (gw < ur ? 1 : 0);
  if (lo >= hi) return;
  const int nch = K >> 3;
  auto row_of = [&](int u, int half) { return ACT ? 32 * (u >> 4) + (u & 15) + 16 * half : u; };

  // Every load of the wave's first batch - weights, bias / residual words, activation, norm weights - is issued before
  // anything waits: the kernel is ONE memory round trip, a reduction and a store (a second dependent round trip, e.g.
  // the norm weights fetched behind the rstd reduction, costs ~1 us on a 4 us kernel)
  u32x4 ww[ROWS][KCH];
  unsigned short bnext = 0;                                 // raw bf16 bits: converting at load time would wait for the load
  float rnext = 0.f;
  auto issue = [&](int u0) {
    const int nrow = min(RB, hi - u0);                      // wave-uniform
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
      const int n = min(u0 + lane, hi - 1);
      if (bias) bnext = reinterpret_cast<const unsigned short*>(bias)[n];
      if (res) rnext = res[n];
    }
  };

  // activation fragment: chunk lane + 64 j, j < KCH (chunks past K/8 are zero).  Its loads go out FIRST: vmcnt retires in
  // issue order, so the norm's reduction can run on them while the (younger) weight loads are still in flight.
  uint32_t xp[KCH][4];
  if constexpr (XMODE == 0) {
    u32x4 xv[KCH];
#pragma unroll
    for (int j = 0; j < KCH; ++j) xv[j] = reinterpret_cast<const u32x4*>(xin)[min(lane + 64 * j, nch - 1)];
    issue(lo);
    G2V_STAMP(1);
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) xp[j][e] = lane + 64 * j < nch ? xv[j][e] : 0u;
    }
  } else {
    const float* xf = reinterpret_cast<const float*>(xin);
    f32x4 a[KCH][2], nwv[KCH][2];
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
      const int c = min(lane + 64 * j, nch - 1);
      a[j][0] = *reinterpret_cast<const f32x4*>(xf + 8 * c);
      a[j][1] = *reinterpret_cast<const f32x4*>(xf + 8 * c + 4);
      nwv[j][0] = *reinterpret_cast<const f32x4*>(norm_w + 8 * c);
      nwv[j][1] = *reinterpret_cast<const f32x4*>(norm_w + 8 * c + 4);
    }
    issue(lo);
    G2V_STAMP(1);
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
      const bool live = lane + 64 * j < nch;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t p0 = pack_bf16x2(__fmul_rn(nwv[j][0][2 * e], __fmul_rn(a[j][0][2 * e], rstd)),
                                        __fmul_rn(nwv[j][0][2 * e + 1], __fmul_rn(a[j][0][2 * e + 1], rstd)));
        const uint32_t p1 = pack_bf16x2(__fmul_rn(nwv[j][1][2 * e], __fmul_rn(a[j][1][2 * e], rstd)),
                                        __fmul_rn(nwv[j][1][2 * e + 1], __fmul_rn(a[j][1][2 * e + 1], rstd)));
        xp[j][e] = live ? p0 : 0u;
        xp[j][2 + e] = live ? p1 : 0u;
      }
    }
  }

  G2V_STAMP(2);
  for (int u0 = lo; u0 < hi; u0 += RB) {
    const int nrow = min(RB, hi - u0);
    if constexpr (KCH > 8) {                                 // long rows: one batch fills the register file, no look-ahead
      if (u0 > lo) issue(u0);
    }
    const float bcur = __uint_as_float((uint32_t)bnext << 16), rcur = rnext;
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      acc[r] = 0.f;
      if ((ACT ? r / 2 : r) < nrow) {
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r] = dot2(ww[r][j][e], xp[j][e], acc[r]);
      }
    }
    if constexpr (KCH <= 8) {
      if (u0 + RB < hi) issue(u0 + RB);                    // next batch in flight under this batch's reduction
    }
    if (u0 == lo) G2V_STAMP(3);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum_dpp(acc[r]);
    if (u0 == lo) G2V_STAMP(4);
    if constexpr (ACT) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (lane == r) v = bfround(siluf_(bfround(acc[2 * r]))) * bfround(acc[2 * r + 1]);
      if (lane < nrow) out[u0 + lane] = f2bf(v);
    } else {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (lane == r) v = acc[r];
      if (lane < nrow) {
        const int n = u0 + lane;
        v = bfround(v + bcur);
        if (res) res[n] = rcur + v;
        else out[n] = f2bf(v);
      }
    }
  }
  G2V_STAMP(5);
  G2V_STAMP_RT(11);
}

// ---- persistent split-KV attention (head_dim 128), q/k-norm + mRoPE + cache append folded in ------------------------------
// grid (NBH, Hkv, scenes).  The key axis is cut by the cache CAPACITY, not by the current length: block (b, kvh, z) owns
// keys [b S, (b+1) S), S = ceil(cap / NBH), of kv head kvh and its 4 waves a quarter each, so every address is known at
// launch and ALL loads of the kernel - the wave's first 32-key batch of K and V, the step's q / k / v rows, the RoPE row,
// the norm weights and the length word - leave together: one memory round trip, then arithmetic, then one store.  Keys at
// or past the length are masked after the fact: their scores to -inf, their V rows to zero (cache rows past the length
// may hold anything, NaN included).  With the cache sized in 4096-row buckets 75-100 % of the blocks have keys.
//
// A wave runs alone on its SIMD, so every dependent LDS round trip or cross-lane shuffle is exposed latency (the first
// form of this kernel - scores by 16x16 MFMA, softmax by 16-lane shuffles, P.V by fp32 FMAs fed from an LDS strip - spent
// 17 000 of its 26 000 cycles in such chains, profiles/r02d_decode_stamps.txt).  This form is the prefill kernel's
// (attn.hip) at M = the kv head's G <= 8 query heads:
//   S^T[key][head] = K . Q^T on mfma_f32_32x32x16_bf16 (A = 32 K rows straight from global, B = the normalised query heads
//   from a wave-private LDS strip, columns past G duplicate head G - 1): a lane owns ONE head and 16 of the batch's 32 keys in
//   registers, so the softmax is in-register plus one v_permlane32_swap; the S accumulator is the B operand of
//   O^T += V^T . P^T (guide §3, permuted-k order); V^T fragments come by ds_read_b64_tr_b16 from the V batch staged in LDS in
//   the dual-use image of attn.hip (written with ds_write_b128 from the coalesced loads).  16 MFMAs per 32 keys, no shuffle.
// The four waves' (m, l, o) are merged through LDS and ONE partial per (query head, block) goes to the workspace:
// ws[((z Hq + head) NBH + b) 130 + {m, l, o[128]}].  The arithmetic of the new token's q / k (norm, rotation, rounding) is
// qknorm_mrope_cache_kernel's (norm_rope.hip), its 16-lane sum in the same order (v_add with DPP row_ror 8, 4, 2, 1 == the
// xor butterfly 8, 4, 2, 1): the appended K row and the scores are bit-identical to the separate kernels.
__global__ __launch_bounds__(256, 2) void decode_attn_pg_kernel(AttnArgs a G2V_STAMP_ARG) {
  __shared__ AttnLds lds;
  decode_attn_pg_body<false, false, 4>(a, lds, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, threadIdx.x G2V_STAMP_PASS_DEV);
}

// out[z][h][d] = sum_b O_b e^(m_b - M) / sum_b l_b e^(m_b - M) over the NBH <= 128 block partials of a head.
// grid (Hq, scenes); 1024 threads = 128 d x 8 groups of 16 consecutive partials.  All loads of a thread - the (m, l) pair
// of partial `tid` and its 16 O words - are issued before the first use (one memory round trip; a loop of dependent loads
// over the partials made this kernel 11.7 us, as long as the attention itself); reductions by DPP / readlane, the 16 weights
// of a group by four 16-byte LDS reads.
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, dpp_f<0x128>(x)); x = fmaxf(x, dpp_f<0x124>(x)); x = fmaxf(x, dpp_f<0x122>(x)); x = fmaxf(x, dpp_f<0x121>(x));
  return x;
}

__global__ __launch_bounds__(1024) void decode_combine_pg_kernel(const float* ws, __bf16* out, int NBH) {
  __shared__ float sm[2];
  __shared__ __attribute__((aligned(16))) float sf[128];
  __shared__ float sL[2], sO[8][128];
  const int h = blockIdx.x, z = blockIdx.y, tid = threadIdx.x, d = tid & 127, g = tid >> 7;
  const float* p = ws + ((size_t)z * gridDim.x + h) * NBH * 130;
  float ov[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) ov[k] = p[min(16 * g + k, NBH - 1) * 130 + 2 + d];
  float m = -INFINITY, l = 0.f;
  if (tid < NBH) { m = p[tid * 130]; l = p[tid * 130 + 1]; }
  if (tid < 128) {                                          // waves 0 and 1 hold the (m, l) pairs
    float mx = row16_max(m);
    mx = fmaxf(fmaxf(readlane_f(mx, 0), readlane_f(mx, 16)), fmaxf(readlane_f(mx, 32), readlane_f(mx, 48)));
    if ((tid & 63) == 0) sm[tid >> 6] = mx;
  }
  __syncthreads();
  const float M = fmaxf(sm[0], sm[1]);
  if (tid < 128) {
    const float f = m == -INFINITY ? 0.f : __expf(m - M);
    sf[tid] = f;
    const float lw = wave_sum_dpp(l * f);
    if ((tid & 63) == 0) sL[tid >> 6] = lw;
  }
  __syncthreads();
  float O = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const f32x4 f4 = *reinterpret_cast<const f32x4*>(&sf[16 * g + 4 * k4]);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (16 * g + 4 * k4 + e < NBH) O = fmaf(ov[4 * k4 + e], f4[e], O);
  }
  sO[g][d] = O;
  __syncthreads();
  if (g == 0) {
    const float Lt = sL[0] + sL[1];
    float Ot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) Ot += sO[k][d];
    out[((size_t)z * gridDim.x + h) * 128 + d] = f2bf(Ot / Lt);
  }
}

template <int XMODE, bool ACT, int KCH>
int gemv_pg_launch_rb(int rb, int blocks, int threads, hipStream_t s, const void* x, const float* nw, float eps, const __bf16* W,
                      const __bf16* bias, __bf16* out, float* res, int N, int K) {
  const int U = ACT ? N / 2 : N, waves = blocks * (threads / 64);
  const int uq = U / waves, ur = U % waves;
#define G2V_PG(RB_)                                                                                                      \
  hipLaunchKernelGGL((gemv_pg_kernel<XMODE, ACT, KCH, RB_>), dim3(blocks), dim3(threads), 0, s, x, nw, eps, W, bias, out, res, N, K, uq, ur G2V_STAMP_PASS)
  if constexpr (KCH > 8) {                                   // long K: one row per batch (18 loads per lane; two rows spill)
    G2V_PG(1);
  } else if constexpr (ACT) {
    if (rb <= 1) G2V_PG(1); else if (rb <= 2) G2V_PG(2); else if (rb <= 3) G2V_PG(3); else if (rb <= 4) G2V_PG(4); else G2V_PG(5);
  } else {
    if (rb <= 1) G2V_PG(1); else if (rb <= 2) G2V_PG(2); else if (rb <= 3) G2V_PG(3); else if (rb <= 4) G2V_PG(4);
    else if (rb <= 5) G2V_PG(5); else if (rb <= 6) G2V_PG(6); else G2V_PG(8);
  }
#undef G2V_PG
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

}  // namespace

// y = W[N,K] . x, batch-1 decode form (reference g2vlm.py:1086-1125, every nn.Linear of the und expert at q_len 1).
//   norm_w != NULL: x is the fp32 residual stream and Qwen2RMSNorm(norm_w, eps) is applied on the fly (modeling_qwen2_vl.py:496-501);
//                   else x is bf16[K].
//   act != 0: W = gate/up interleaved per 16 rows, N = 2F rows; out bf16[F] = bf16(bf16(silu(g)) * u) (modeling_qwen2_vl.py:519-521).
//   res != NULL: res[n] (f32) += bf16(y[n] + bias[n]); else out[n] = that.
// K % 8 == 0, K <= 9216.  The grid is 256 blocks whatever N: see the header of this file.
extern "C" int g2v_gemv_pg(const void* x, const void* norm_w, float eps, const void* W, const void* bias, void* out, void* res, int N,
                           int K, int act, void* stream) {
  if (!x || !W || (!out && !res) || N <= 0 || K <= 0 || (K & 7) || K > 9216) return G2V_ERR_ARG;
  if (act && ((N & 31) || !out || res || !norm_w)) return G2V_ERR_ARG;      // the activation form is the MLP's first half: norm fused
  const int kch = (K / 8 + 63) / 64;
  if (norm_w && kch > 3) return G2V_ERR_ARG;                 // the fused norm keeps the fp32 row in registers: hidden-size K
  const int U = act ? N / 2 : N;
  // waves per block: the count (3..8) that splits the units most evenly over 256 blocks; a wave then takes ceil(U / waves)
  // units.  Ties go to MORE waves for a streaming kernel (loads in flight per CU) and to FEWER, fatter waves for a small one
  // (< 48 KB per CU: all of it is in flight either way, and 1536 waves take ~1.4 us to dispatch - half of a 3 us kernel,
  // profiles/r02f_decode_stamps.txt: wave life 2.1 us, kernel span 3.5 us)
  const int rb_cap0 = kch > 3 ? 1 : (act ? 5 : 8);
  const bool small = (double)N * K * 2.0 / 256.0 < 48.0 * 1024.0;
  int best = 4;
  double best_imb = 1e30;
  for (int t = 0; t < 6; ++t) {
    const int nwb = small ? 3 + t : 8 - t;
    const long nw = 256L * nwb;
    const double per = (double)U / nw;
    const double imb = per >= 1.0 ? (double)((U + nw - 1) / nw) / per : 1.0 / per;
    if (small && (U + nw - 1) / nw > rb_cap0 && best_imb < 1e29) continue;   // a small kernel is ONE batch per wave
    if (imb < best_imb - 1e-9) { best_imb = imb; best = nwb; }
  }
  const long nw = 256L * best;
  const int per_wave = (int)((U + nw - 1) / nw);
  const int rb_cap = rb_cap0;                                // registers: ROWS x KCH x 4 per batch
  int rb = per_wave;
  if (rb > rb_cap) {                                         // several equal batches rather than a full one and a remainder
    const int nb = (per_wave + rb_cap - 1) / rb_cap;
    rb = (per_wave + nb - 1) / nb;
  }
  hipStream_t s = (hipStream_t)stream;
  const float* nwp = (const float*)norm_w;
  const __bf16 *Wp = (const __bf16*)W, *bp = (const __bf16*)bias;
  const int threads = 64 * best;
  if (norm_w) {
    if (act) return gemv_pg_launch_rb<1, true, 3>(rb, 256, threads, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, N, K);
    return gemv_pg_launch_rb<1, false, 3>(rb, 256, threads, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, N, K);
  }
  if (kch <= 3) return gemv_pg_launch_rb<0, false, 3>(rb, 256, threads, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, N, K);
  return gemv_pg_launch_rb<0, false, 18>(rb, 256, threads, s, x, nwp, eps, Wp, bp, (__bf16*)out, (float*)res, N, K);
}

extern "C" int64_t g2v_decode_attn_pg_workspace(int Hq, int Hkv, int batch) {
  if (Hq <= 0 || Hkv <= 0 || Hkv > 128 || batch <= 0) return 0;
  return (int64_t)batch * Hq * (256 / Hkv > 128 ? 128 : 256 / Hkv) * 130 * 4;
}

// The decode step's attention, persistent-grid form of g2v_decode_attn_fused (same arguments, same results up to the order
// of the fp32 partial sums): see decode_attn_pg_kernel.  workspace >= g2v_decode_attn_pg_workspace(Hq, Hkv, batch) bytes.
extern "C" int g2v_decode_attn_pg(const void* qkv, const void* q_norm_w, const void* k_norm_w, float eps, int und_rounding,
                                  const void* cos, const void* sin, void* k_cache, void* v_cache, void* out, const void* Lk_dev,
                                  int batch, int64_t scene_rows, int max_len, int Hq, int Hkv, float scale, void* workspace,
                                  void* stream) {
  if (!qkv || !q_norm_w || !k_norm_w || !cos || !sin || !k_cache || !v_cache || !out || !workspace || !Lk_dev || batch <= 0 ||
      batch > 65535 || max_len <= 0 || scene_rows < max_len || Hq <= 0 || Hkv <= 0 || Hkv > 128 || Hq % Hkv || Hq / Hkv > GMAX)
    return G2V_ERR_ARG;
  // 2..128 partials per head (the combine reads <= 128): 256 blocks per scene for one or two scenes; from three scenes on
  // fewer, longer shares (>= 512 blocks in all, >= 8 per kv head) - at B = 8 a block with 96 keys spends its life in the
  // prologue (3.0 TB/s), one with 384 keys streams three batches per wave behind it
  const int nbh1 = 256 / Hkv > 128 ? 128 : 256 / Hkv;
  const int nbhb = 512 / (Hkv * batch) < 8 ? 8 : 512 / (Hkv * batch);
  const int nbh = nbhb < nbh1 ? nbhb : nbh1;
  AttnArgs a{(const __bf16*)qkv, (const float*)q_norm_w, (const float*)k_norm_w, (const float*)cos, (const float*)sin, eps, und_rounding,
             (__bf16*)k_cache, (__bf16*)v_cache, (float*)workspace, (const int*)Lk_dev, Hq, Hkv, scale, (long)scene_rows, max_len, (max_len + nbh - 1) / nbh,
             ((max_len + nbh - 1) / nbh + 3) / 4};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(decode_attn_pg_kernel, dim3(nbh, Hkv, batch), dim3(256), 0, s, a G2V_STAMP_PASS);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_combine_pg_kernel, dim3(Hq, batch), dim3(1024), 0, s, (const float*)workspace, (__bf16*)out, nbh);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

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
//   * prefetch: a copy-nothing kernel that pulls a byte range through L2 into the 256 MiB Infinity Cache; launched on a
//     side branch of the step's graph it fills the HBM idle time under the latency-bound kernels (attention, combine,
//     o-proj) with the NEXT Linear's weights.
#include "common.h"
#include "g2vlm_hip.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;

__device__ __forceinline__ float dot2(uint32_t w, uint32_t x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf2_t*>(&w), *reinterpret_cast<bf2_t*>(&x), acc, false);
}

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
                                                      const __bf16* bias, __bf16* out, float* res, int N, int K) {
  constexpr int ROWS = ACT ? 2 * RB : RB;
  const int lane = threadIdx.x & 63;
  const int nwb = blockDim.x >> 6;
  const long gw = (long)blockIdx.x * nwb + (threadIdx.x >> 6), nw = (long)gridDim.x * nwb;
  const int U = ACT ? N / 2 : N;
  const int lo = (int)(gw * U / nw), hi = (int)((gw + 1) * U / nw);
  if (lo >= hi) return;
  const int nch = K >> 3;
  auto row_of = [&](int u, int half) { return ACT ? 32 * (u >> 4) + (u & 15) + 16 * half : u; };

  // first batch of weight loads goes out before the activation is touched
  u32x4 ww[ROWS][KCH];
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
  };
  issue(lo);

  // activation fragment: chunk lane + 64 j, j < KCH (chunks past K/8 are zero)
  uint32_t xp[KCH][4];
  if constexpr (XMODE == 0) {
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
      const int c = lane + 64 * j;
      u32x4 v = reinterpret_cast<const u32x4*>(xin)[min(c, nch - 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) xp[j][e] = c < nch ? v[e] : 0u;
    }
  } else {
    const float* xf = reinterpret_cast<const float*>(xin);
    f32x4 a[KCH][2];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
      const int c = lane + 64 * j;
      const bool live = c < nch;
      a[j][0] = *reinterpret_cast<const f32x4*>(xf + 8 * min(c, nch - 1));
      a[j][1] = *reinterpret_cast<const f32x4*>(xf + 8 * min(c, nch - 1) + 4);
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += a[j][0][e] * a[j][0][e] + a[j][1][e] * a[j][1][e];
      }
    }
    ss = wave_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / (float)K + eps);
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
      const int c = lane + 64 * j;
      const f32x4 wa = *reinterpret_cast<const f32x4*>(norm_w + 8 * min(c, nch - 1));
      const f32x4 wb = *reinterpret_cast<const f32x4*>(norm_w + 8 * min(c, nch - 1) + 4);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t p0 = pack_bf16x2(__fmul_rn(wa[2 * e], __fmul_rn(a[j][0][2 * e], rstd)), __fmul_rn(wa[2 * e + 1], __fmul_rn(a[j][0][2 * e + 1], rstd)));
        const uint32_t p1 = pack_bf16x2(__fmul_rn(wb[2 * e], __fmul_rn(a[j][1][2 * e], rstd)), __fmul_rn(wb[2 * e + 1], __fmul_rn(a[j][1][2 * e + 1], rstd)));
        xp[j][e] = c < nch ? p0 : 0u;
        xp[j][2 + e] = c < nch ? p1 : 0u;
      }
    }
  }

  for (int u0 = lo; u0 < hi; u0 += RB) {
    const int nrow = min(RB, hi - u0);
    if constexpr (KCH > 8) {                                 // long rows: one batch fills the register file, no look-ahead
      if (u0 > lo) issue(u0);
    }
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
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
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
        v = bfround(v + (bias ? bf2f(bias[n]) : 0.f));
        if (res) res[n] = res[n] + v;
        else out[n] = f2bf(v);
      }
    }
  }
}

// ---- persistent split-KV attention (head_dim 128), q/k-norm + mRoPE + cache append folded in ------------------------------
// grid (NBH, Hkv, scenes): block (b, kvh, z) owns keys [b Lk / NBH, (b+1) Lk / NBH) of kv head kvh; its 4 waves take a
// quarter each and walk it in batches of 32 keys: all K / V loads of a batch first (16 x 16 B per lane), scores by MFMA
// 16x16x32 (A = the kv head's G <= 8 query heads padded to 16 rows, B = K rows straight from global), online softmax over
// the wave's batches, P.V by fp32 FMAs with p passed through a wave-private LDS strip.  The four waves' (m, l, o) are merged
// through LDS and ONE partial per (query head, block) goes to the workspace: ws[((z Hq + head) NBH + b) 130 + {m, l, o[128]}].
// The arithmetic of the new token's q / k (norm, rotation, rounding) is qknorm_mrope_cache_kernel's, instruction for
// instruction (norm_rope.hip), so the appended K row and the scores are bit-identical to the separate kernels.
constexpr int KB = 32, GMAX = 8;

struct AttnArgs {
  const __bf16* qkv; const float* qw; const float* kw; const float* cs; const float* sn; float eps; int und_rounding;
  __bf16* kc; __bf16* vc; float* ws; const int* Lk_dev; int Hq, Hkv; float scale; long scene_rows;
};

__global__ __launch_bounds__(256, 2) void decode_attn_pg_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sq[4][GMAX + 1][128];   // per wave: normalised q heads + the new k
  __shared__ float sp[4][GMAX * KB];                                      // per wave: p[h][key] of the current batch
  __shared__ float sal[4][GMAX];                                          // per wave: rescale factor per head
  __shared__ float wm[4][GMAX], wl[4][GMAX];
  __shared__ __attribute__((aligned(16))) float wo[4][GMAX][128];
  const int z = blockIdx.z, kvh = blockIdx.y, NBH = gridDim.x;
  const int Hq = a.Hq, Hkv = a.Hkv, G = Hq / Hkv;
  const int Lk = a.Lk_dev[z];
  const __bf16* q = a.qkv + (size_t)z * (Hq + 2 * Hkv) * 128;
  __bf16* kc = a.kc + (size_t)z * a.scene_rows * Hkv * 128;
  __bf16* vc = a.vc + (size_t)z * a.scene_rows * Hkv * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const size_t row_stride = (size_t)Hkv * 128;
  const int blo = (int)((long)blockIdx.x * Lk / NBH), bhi = (int)((long)(blockIdx.x + 1) * Lk / NBH);
  const int wlo = blo + (int)((long)w * (bhi - blo) / 4), whi = blo + (int)((long)(w + 1) * (bhi - blo) / 4);
  const bool has_new = whi == Lk && whi > wlo;              // this wave's range ends with the new token's row (wave-uniform)

  float m_run[4], l_run[4];                                 // heads 4 fg + r, as the MFMA leaves the scores
  float acc[GMAX][8];                                       // dims 8 fr .. 8 fr + 7, keys 4 i + fg
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
  for (int h = 0; h < GMAX; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;

  if (wlo < whi) {
    // ---- first batch's loads, then the query side while they fly
    bf16x8 kf[2][4];
    u32x4 vv[8];
    auto load_batch = [&](int k0, int nk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        if (16 * kb < nk) {
          const __bf16* kp = kc + (size_t)(k0 + min(16 * kb + fr, nk - 1)) * row_stride + kvh * 128 + 8 * fg;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8*>(kp + 32 * ks);
        }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (4 * i < nk) vv[i] = *reinterpret_cast<const u32x4*>(vc + (size_t)(k0 + min(4 * i + fg, nk - 1)) * row_stride + kvh * 128 + 8 * fr);
    };
    load_batch(wlo, min(KB, whi - wlo));
    u32x4 vnew = {0u, 0u, 0u, 0u};
    if (has_new) vnew = *reinterpret_cast<const u32x4*>(q + (size_t)(Hq + Hkv + kvh) * 128 + 8 * fr);
    {
      const float* cs = a.cs + (size_t)z * 128;
      const float* sn = a.sn + (size_t)z * 128;
      const int n_items = G + (has_new ? 1 : 0);
      const int j = lane & 15;
      for (int it0 = 0; it0 < n_items; it0 += 4) {           // 16 lanes per head, 4 heads per pass
        const int item = min(it0 + (lane >> 4), n_items - 1);
        const bool isq = item < G;
        const __bf16* src = q + (size_t)(isq ? kvh * G + item : Hq + kvh) * 128 + 4 * j;
        const u32x2 x0r = *reinterpret_cast<const u32x2*>(src), x1r = *reinterpret_cast<const u32x2*>(src + 64);
        float x0[4] = {bits2f_lo(x0r[0]), bits2f_hi(x0r[0]), bits2f_lo(x0r[1]), bits2f_hi(x0r[1])};
        float x1[4] = {bits2f_lo(x1r[0]), bits2f_hi(x1r[0]), bits2f_lo(x1r[1]), bits2f_hi(x1r[1])};
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += x0[e] * x0[e] + x1[e] * x1[e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = 1.0f / sqrtf(ss / 128.f + a.eps);
        const float* wp = (isq ? a.qw : a.kw) + 4 * j;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 64);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 4 * j), c1 = *reinterpret_cast<const f32x4*>(cs + 64 + 4 * j);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sn + 4 * j), s1 = *reinterpret_cast<const f32x4*>(sn + 64 + 4 * j);
        float o0[4], o1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float n0 = __fmul_rn(x0[e], rstd), n1 = __fmul_rn(x1[e], rstd);
          if (a.und_rounding) { n0 = bfround(n0); n1 = bfround(n1); }
          n0 = __fmul_rn(w0[e], n0); n1 = __fmul_rn(w1[e], n1);
          o0[e] = __fadd_rn(__fmul_rn(n0, c0[e]), __fmul_rn(-n1, s0[e]));
          o1[e] = __fadd_rn(__fmul_rn(n1, c1[e]), __fmul_rn(n0, s1[e]));
        }
        const u32x2 p0 = {pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3])}, p1 = {pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
        *reinterpret_cast<u32x2*>(&sq[w][item][4 * j]) = p0;
        *reinterpret_cast<u32x2*>(&sq[w][item][64 + 4 * j]) = p1;
        if (!isq) {                                          // the new token's K row -> cache row Lk - 1 of this scene
          __bf16* krow = kc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 4 * j;
          *reinterpret_cast<u32x2*>(krow) = p0;
          *reinterpret_cast<u32x2*>(krow + 64) = p1;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    bf16x8 qa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(&sq[w][min(fr, G - 1)][32 * ks + 8 * fg]);
    if (has_new && fg == 0) *reinterpret_cast<u32x4*>(vc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 8 * fr) = vnew;

    for (int k0 = wlo; k0 < whi; k0 += KB) {
      const int nk = min(KB, whi - k0);
      if (has_new && k0 + nk == whi) {
        // the batch that ends with the new row: the loads above read whatever the cache row held BEFORE this step (maybe NaN);
        // every lane whose key is at or past it (clamped duplicates included) takes the fresh row instead
        const int new_local = Lk - 1 - k0;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          if (16 * kb + fr >= new_local) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8*>(&sq[w][G][32 * ks + 8 * fg]);
          }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (4 * i + fg >= new_local) vv[i] = vnew;
      }
      // ---- scores: register r of lane (fr, fg) is S[head 4 fg + r][key 16 kb + fr]
      f32x4 S[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        S[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (16 * kb < nk) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) S[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kf[kb][ks], S[kb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          S[kb][r] = (16 * kb + fr < nk) ? S[kb][r] * a.scale : -INFINITY;
          mx = fmaxf(mx, S[kb][r]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float m_new = fmaxf(m_run[r], mx);             // finite: nk >= 1
        const float alpha = expf(m_run[r] - m_new);          // first batch: exp(-inf) = 0
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const float p = expf(S[kb][r] - m_new);
          S[kb][r] = p;
          sum += p;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        l_run[r] = l_run[r] * alpha + sum;
        m_run[r] = m_new;
        const int h = 4 * fg + r;
        if (h < G) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) sp[w][h * KB + 16 * kb + fr] = S[kb][r];
          if (fr == 0) sal[w][h] = alpha;
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      // ---- o[h][d] = alpha o[h][d] + sum_key p[h][key] V[key][d]
      const bool rescale = k0 > wlo;                          // wave-uniform
#pragma unroll
      for (int h = 0; h < GMAX; ++h)
        if (h < G && rescale) {
          const float al = sal[w][h];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h][e] *= al;
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (4 * i < nk) {
          const int key = 4 * i + fg;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] = bits2f_lo(vv[i][e]); v[2 * e + 1] = bits2f_hi(vv[i][e]); }
#pragma unroll
          for (int h = 0; h < GMAX; ++h)
            if (h < G) {
              const float p = sp[w][h * KB + key];            // 0 for key >= nk (exp(-inf))
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(p, v[e], acc[h][e]);
            }
        }
      }
      __builtin_amdgcn_wave_barrier();                       // every lane is done with sp / sal before the next batch rewrites them
      if (k0 + KB < whi) load_batch(k0 + KB, min(KB, whi - k0 - KB));
    }
  }
  // ---- the wave's result to LDS: (m, l) per head from the lanes that hold them, o summed over the four key sub-groups
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int h = 4 * fg + r;
    if (h < G && fr == 0) { wm[w][h] = m_run[r]; wl[w][h] = l_run[r]; }
  }
#pragma unroll
  for (int h = 0; h < GMAX; ++h)
    if (h < G) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[h][e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[h][e] = v;
      }
      if (fg == 0) {
        *reinterpret_cast<f32x4*>(&wo[w][h][8 * fr]) = f32x4{acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
        *reinterpret_cast<f32x4*>(&wo[w][h][8 * fr + 4]) = f32x4{acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
      }
    }
  __syncthreads();
  // ---- merge the four waves: one partial per (head, block)
  for (int idx = tid; idx < G * 128; idx += 256) {
    const int h = idx >> 7, d = idx & 127;
    float M = fmaxf(fmaxf(wm[0][h], wm[1][h]), fmaxf(wm[2][h], wm[3][h]));
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = wm[k][h] == -INFINITY ? 0.f : expf(wm[k][h] - M);
      L = fmaf(wl[k][h], f, L);
      O = fmaf(wo[k][h][d], f, O);
    }
    float* o = a.ws + (((size_t)z * Hq + kvh * G + h) * NBH + blockIdx.x) * 130;
    if (d == 0) { o[0] = M; o[1] = L; }
    o[2 + d] = O;
  }
}

// out[z][h][d] = sum_b O_b e^(m_b - M) / sum_b l_b e^(m_b - M) over the NBH block partials of a head.
// grid (Hq, scenes); 512 threads = 128 d x 4 partial groups; the m / l loads are shared by the 128 d-threads of a group.
__global__ __launch_bounds__(512) void decode_combine_pg_kernel(const float* ws, __bf16* out, int NBH) {
  __shared__ float sm[8], sL[4], sO[4][128];
  const int h = blockIdx.x, z = blockIdx.y, tid = threadIdx.x, d = tid & 127, g = tid >> 7;
  const float* p = ws + ((size_t)z * gridDim.x + h) * NBH * 130;
  float mx = -INFINITY;
  for (int b = tid; b < NBH; b += 512) mx = fmaxf(mx, p[b * 130]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) sm[tid >> 6] = mx;
  __syncthreads();
  float M = sm[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) M = fmaxf(M, sm[k]);
  float L = 0.f, O = 0.f;
  for (int b = g; b < NBH; b += 4) {
    const float m = p[b * 130];
    const float f = m == -INFINITY ? 0.f : expf(m - M);
    L = fmaf(p[b * 130 + 1], f, L);
    O = fmaf(p[b * 130 + 2 + d], f, O);
  }
  if (d == 0) sL[g] = L;
  sO[g][d] = O;
  __syncthreads();
  if (g == 0) {
    const float Lt = (sL[0] + sL[1]) + (sL[2] + sL[3]);
    const float Ot = (sO[0][d] + sO[1][d]) + (sO[2][d] + sO[3][d]);
    out[((size_t)z * gridDim.x + h) * 128 + d] = f2bf(Ot / Lt);
  }
}

// ---- prefetch: read a byte range and drop it.  The lines stay in the Infinity Cache (256 MiB, memory side) for the kernel
// that streams them next; the xor chain keeps the loads alive without a store.
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* p, long n16, uint32_t* sink) {
  uint32_t acc = 0;
  const long stride = (long)gridDim.x * 256 * 8;
  for (long i = (long)blockIdx.x * 256 * 8 + threadIdx.x; i < n16; i += stride) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[min(i + 256 * k, n16 - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
  }
  if (acc == 0x9E3779B9u && sink) sink[0] = acc;            // practically never: the loads just must not be dead code
}

template <int XMODE, bool ACT, int KCH>
int gemv_pg_launch_rb(int rb, int blocks, int threads, hipStream_t s, const void* x, const float* nw, float eps, const __bf16* W,
                      const __bf16* bias, __bf16* out, float* res, int N, int K) {
#define G2V_PG(RB_)                                                                                                      \
  hipLaunchKernelGGL((gemv_pg_kernel<XMODE, ACT, KCH, RB_>), dim3(blocks), dim3(threads), 0, s, x, nw, eps, W, bias, out, res, N, K)
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
  // waves per block: the count (3..8) that splits the units most evenly over 256 blocks; a wave then takes ceil(U / waves) units
  int best = 4;
  double best_imb = 1e30;
  for (int nwb = 8; nwb >= 3; --nwb) {
    const long nw = 256L * nwb;
    const double per = (double)U / nw;
    const double imb = per >= 1.0 ? (double)((U + nw - 1) / nw) / per : 1.0 / per;
    if (imb < best_imb - 1e-9) { best_imb = imb; best = nwb; }
  }
  const long nw = 256L * best;
  const int per_wave = (int)((U + nw - 1) / nw);
  const int rb_cap = kch > 3 ? 1 : (act ? 5 : 8);            // registers: ROWS x KCH x 4 per batch
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
  if (Hq <= 0 || Hkv <= 0 || Hkv > 256 || batch <= 0) return 0;
  return (int64_t)batch * Hq * (256 / Hkv) * 130 * 4;
}

// The decode step's attention, persistent-grid form of g2v_decode_attn_fused (same arguments, same results up to the order
// of the fp32 partial sums): see decode_attn_pg_kernel.  workspace >= g2v_decode_attn_pg_workspace(Hq, Hkv, batch) bytes.
extern "C" int g2v_decode_attn_pg(const void* qkv, const void* q_norm_w, const void* k_norm_w, float eps, int und_rounding,
                                  const void* cos, const void* sin, void* k_cache, void* v_cache, void* out, const void* Lk_dev,
                                  int batch, int64_t scene_rows, int Hq, int Hkv, float scale, void* workspace, void* stream) {
  if (!qkv || !q_norm_w || !k_norm_w || !cos || !sin || !k_cache || !v_cache || !out || !workspace || !Lk_dev || batch <= 0 ||
      batch > 65535 || scene_rows <= 0 || Hq <= 0 || Hkv <= 0 || Hkv > 256 || Hq % Hkv || Hq / Hkv > GMAX) return G2V_ERR_ARG;
  const int nbh = 256 / Hkv;
  AttnArgs a{(const __bf16*)qkv, (const float*)q_norm_w, (const float*)k_norm_w, (const float*)cos, (const float*)sin, eps, und_rounding,
             (__bf16*)k_cache, (__bf16*)v_cache, (float*)workspace, (const int*)Lk_dev, Hq, Hkv, scale, (long)scene_rows};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(decode_attn_pg_kernel, dim3(nbh, Hkv, batch), dim3(256), 0, s, a);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_combine_pg_kernel, dim3(Hq, batch), dim3(512), 0, s, (const float*)workspace, (__bf16*)out, nbh);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// Pull `bytes` at p (16-byte aligned) through the cache hierarchy with `blocks` workgroups and discard them: a hint for
// the kernel that streams the range next (see the header of this file).  Never needed for correctness.
extern "C" int g2v_prefetch(const void* p, int64_t bytes, int blocks, void* stream) {
  if (!p || bytes < 0 || blocks <= 0 || ((uintptr_t)p & 15)) return G2V_ERR_ARG;
  if (bytes < 16) return G2V_OK;
  hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)p, (long)(bytes / 16), (uint32_t*)nullptr);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

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
#include "g2vlm_hip.h"

// In-kernel stamps (diagnostic build only: -DG2V_STAMPS, tools/decode_stamps.py; the shipped library executes none).  Lane 0
// of every wave stores s_memtime at up to 8 points of the kernel plus s_memrealtime at its start and end.
#ifdef G2V_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
extern "C" int g2v_debug_stamps(void* buf) { g_stamp_buf = (unsigned long long*)buf; return 0; }
#define G2V_STAMP_ARG , unsigned long long* dbg
#define G2V_STAMP_PASS , g_stamp_buf
#define G2V_STAMP(i)                                                                                                   \
  do {                                                                                                                 \
    if (dbg) {                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      unsigned long long t__;                                                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if ((threadIdx.x & 63) == 0) dbg[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12 + (i)] = t__; \
    }                                                                                                                  \
  } while (0)
#define G2V_STAMP_RT(i)                                                                                                \
  do {                                                                                                                 \
    if (dbg) {                                                                                                         \
      unsigned long long t__;                                                                                          \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                                  \
      if ((threadIdx.x & 63) == 0) dbg[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 12 + (i)] = t__; \
    }                                                                                                                  \
  } while (0)
#else
#define G2V_STAMP_ARG
#define G2V_STAMP_PASS
#define G2V_STAMP(i)
#define G2V_STAMP_RT(i)
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
constexpr int KB = 32, GMAX = 8;

struct AttnArgs {
  const __bf16* qkv; const float* qw; const float* kw; const float* cs; const float* sn; float eps; int und_rounding;
  __bf16* kc; __bf16* vc; float* ws; const int* Lk_dev; int Hq, Hkv; float scale; long scene_rows; int cap, S, SW;
};

// byte offset of 16-byte chunk `ch` of row `row` in the dual-use LDS image (attn.hip lds_off, guide T10 layout (a))
__device__ __forceinline__ int v_img_off(int row, int ch) {
  return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

__global__ __launch_bounds__(256, 2) void decode_attn_pg_kernel(AttnArgs a G2V_STAMP_ARG) {
  __shared__ __attribute__((aligned(16))) __bf16 sq[4][GMAX + 1][128];   // per wave: normalised q heads + the new k
  __shared__ __attribute__((aligned(16))) char sv[4][KB * 256];           // per wave: the V batch, dual-use image
  __shared__ float wm[4][GMAX], wl[4][GMAX];
  __shared__ __attribute__((aligned(16))) float wo[4][GMAX][128];
  G2V_STAMP_RT(10);
  G2V_STAMP(0);
  const int z = blockIdx.z, kvh = blockIdx.y, NBH = gridDim.x;
  const int Hq = a.Hq, Hkv = a.Hkv, G = Hq / Hkv;
  const __bf16* q = a.qkv + (size_t)z * (Hq + 2 * Hkv) * 128;
  __bf16* kc = a.kc + (size_t)z * a.scene_rows * Hkv * 128;
  __bf16* vc = a.vc + (size_t)z * a.scene_rows * Hkv * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;                 // MFMA 32x32: row / column index, k half
  const int fr = lane & 15, fg = lane >> 4;
  const int row_stride = Hkv * 128;                          // elements
  const int S = a.S, SW = a.SW;                             // keys per block / per wave, by capacity (host: ceil(cap / NBH), ceil(S / 4))
  const int wlo = blockIdx.x * S + w * SW;
  const int wcap = min(min(wlo + SW, (int)(blockIdx.x + 1) * S), a.cap);   // end of this wave's range if the cache were full
  const int last_row = a.cap - 1;
  const __bf16* kbase = kc + kvh * 128 + 8 * hh;            // A operand: lane (r32, hh) takes K[key r32][16 ks + 8 hh ..]
  const __bf16* vbase = vc + kvh * 128 + 8 * fr;            // staging: lane (fg, fr) takes V[row 4 i + fg][8 fr ..]

  // ---- every load first, the step's own rows before the cache (vmcnt retires in issue order: the norms below run while
  // K / V are in flight).  Rows are clamped to the cache block (always mapped); what lies past the length is masked below.
  const int j = lane & 15;
  u32x2 x0r[3], x1r[3];
#pragma unroll
  for (int ps = 0; ps < 3; ++ps) {
    const int item = min(4 * ps + (lane >> 4), G);          // G = the new token's k row
    const __bf16* src = q + (size_t)(item < G ? kvh * G + item : Hq + kvh) * 128 + 4 * j;
    x0r[ps] = *reinterpret_cast<const u32x2*>(src);
    x1r[ps] = *reinterpret_cast<const u32x2*>(src + 64);
  }
  const u32x4 vnew = *reinterpret_cast<const u32x4*>(q + (size_t)(Hq + Hkv + kvh) * 128 + 8 * fr);
  const float* cs = a.cs + (size_t)z * 128;
  const float* sn = a.sn + (size_t)z * 128;
  const f32x4 qw0 = *reinterpret_cast<const f32x4*>(a.qw + 4 * j), qw1 = *reinterpret_cast<const f32x4*>(a.qw + 64 + 4 * j);
  const f32x4 kw0 = *reinterpret_cast<const f32x4*>(a.kw + 4 * j), kw1 = *reinterpret_cast<const f32x4*>(a.kw + 64 + 4 * j);
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 4 * j), c1 = *reinterpret_cast<const f32x4*>(cs + 64 + 4 * j);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sn + 4 * j), s1 = *reinterpret_cast<const f32x4*>(sn + 64 + 4 * j);
  bf16x8 kf[8];
  u32x4 vv[8];
  auto load_batch = [&](int k0) {
    const __bf16* kp = kbase + (size_t)min(k0 + r32, last_row) * row_stride;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
#pragma unroll
    for (int i = 0; i < 8; ++i) vv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)min(k0 + 4 * i + fg, last_row) * row_stride);
  };
  load_batch(wlo);
  const int Lk = a.Lk_dev[z];
  G2V_STAMP(1);

  const int whi = min(wcap, Lk);                            // the wave's real range is [wlo, whi)
  const bool has_new = wlo < whi && whi == Lk;              // it ends with the new token's row (wave-uniform)

  // ---- q / k norm + rotation of the step's rows: every wave, unconditionally (the loads above must not end up behind the
  // wait for the length word; only the STORES of the new k row depend on it)
#pragma unroll
  for (int ps = 0; ps < 3; ++ps) {
    if (4 * ps < G + 1) {                                   // uniform over the launch
      const int c = 4 * ps + (lane >> 4);
      const int item = min(c, G);                           // what this group loaded above
      const bool isq = item < G;
      const u32x2 a0 = x0r[ps], a1 = x1r[ps];
      float x0[4] = {bits2f_lo(a0[0]), bits2f_hi(a0[0]), bits2f_lo(a0[1]), bits2f_hi(a0[1])};
      float x1[4] = {bits2f_lo(a1[0]), bits2f_hi(a1[0]), bits2f_lo(a1[1]), bits2f_hi(a1[1])};
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) ss += x0[e] * x0[e] + x1[e] * x1[e];
      ss = row16_sum(ss);
      const float rstd = 1.0f / sqrtf(ss / 128.f + a.eps);
      const f32x4 w0 = isq ? qw0 : kw0, w1 = isq ? qw1 : kw1;
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
      if (c <= G) {                                          // strip rows 0..G (row G is only read by the wave that owns the new row)
        *reinterpret_cast<u32x2*>(&sq[w][item][4 * j]) = p0;
        *reinterpret_cast<u32x2*>(&sq[w][item][64 + 4 * j]) = p1;
      }
      if (c == G && has_new) {                               // the new token's K row -> cache row Lk - 1 of this scene
        __bf16* krow = kc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 4 * j;
        *reinterpret_cast<u32x2*>(krow) = p0;
        *reinterpret_cast<u32x2*>(krow + 64) = p1;
      }
    }
  }
  G2V_STAMP(2);
  float m_run = -INFINITY, l_run = 0.f;                      // this lane's head (column r32), raw-score units / its half's keys
  f32x16 O[4];                                               // O^T[d = 32 blk + row][head r32]
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
  const float c2 = a.scale * 1.4426950408889634f;            // p = 2^((s - m) c2)

  if (wlo < whi) {
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // the strip is written and read by this wave only
    __builtin_amdgcn_wave_barrier();
    bf16x8 qf[8];                                            // B operand: Q^T[d = 16 ks + 8 hh + j][head r32]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(&sq[w][min(r32, G - 1)][16 * ks + 8 * hh]);
    if (has_new && fg == 0) *reinterpret_cast<u32x4*>(vc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 8 * fr) = vnew;
    char* sV = sv[w];
    // V^T fragment addresses (attn.hip): row 16 s + 8 jj + 4 hh + tq, chunk 4 d + t_ch -> v_lb[jj] + 2048 (2 s + jj) + 512 d
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int t_ch = 2 * ((lane >> 4) & 1) + (tp >> 1);
    int v_lb[2];
    v_lb[0] = 64 * (4 * hh + tq) + 16 * (t_ch ^ hh) + 8 * (tp & 1);
    v_lb[1] = v_lb[0] ^ 32;

    for (int k0 = wlo; k0 < whi; k0 += KB) {
      const int nk = min(KB, whi - k0);
      if (has_new && k0 + nk == whi) {
        // the batch that ends with the new row: the loads above read whatever the cache row held BEFORE this step
        const int new_local = Lk - 1 - k0;
        if (r32 == new_local) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(&sq[w][G][16 * ks + 8 * hh]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (4 * i + fg == new_local) vv[i] = vnew;
      }
      // ---- V batch -> LDS image (rows at or past nk as zeros: 0 x NaN must not reach the MFMA)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + fg;
        const u32x4 val = row < nk ? vv[i] : u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(sV + v_img_off(row, fr)) = val;
      }
      // ---- S^T = K . Q^T: register e of lane (r32, hh) is S[key (e & 3) + 8 (e >> 2) + 4 hh][head r32]
      f32x16 Sx;
#pragma unroll
      for (int e = 0; e < 16; ++e) Sx[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], Sx, 0, 0, 0);
      if (k0 + KB < whi) {                                   // next batch's K / V under this batch's softmax and P.V
        // (kf and vv are free: the MFMAs above have issued, the LDS stores have read vv)
        load_batch(k0 + KB);
      }
      float rmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * hh;
        Sx[e] = key < nk ? Sx[e] : -INFINITY;
        rmax = fmaxf(rmax, Sx[e]);
      }
      {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
        rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float m_new = fmaxf(m_run, rmax);                // finite: nk >= 1
      if (k0 > wlo) {                                        // wave-uniform: a second batch rescales what the first left
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) O[d][e] *= alpha;
      }
      m_run = m_new;
      const float mc = m_new * c2;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(Sx[e], c2, -mc));      // masked keys: exp2(-inf) = 0
        Sx[e] = pv;
        psum += pv;
      }
      l_run += psum;
      bf16x8 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf[s2][jj] = f2bf(Sx[8 * s2 + jj]);
      // ---- O^T += V^T . P^T
      __builtin_amdgcn_s_waitcnt(0xC07F);                    // this wave's V stores have landed (wave-private image)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s2 = i >> 2, d = i & 3;
        union { struct { s16x4 a, b; } s; bf16x8 v; } uu;
        uu.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + v_lb[0] + 2048 * (2 * s2) + 512 * d));
        uu.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + v_lb[1] + 2048 * (2 * s2 + 1) + 512 * d));
        O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uu.v, pf[s2], O[d], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();                       // the reads are issued before the next batch's stores (same wave, in order)
      if (k0 == wlo) G2V_STAMP(3);
    }
  }
  G2V_STAMP(4);
  // ---- the wave's result to LDS: lanes r32 < G hold head r32; the two halves hold disjoint d rows and partial l
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    if (r32 < G) {
      if (hh == 0) { wm[w][r32] = m_run * a.scale; wl[w][r32] = l_tot; }      // natural-log units, as the combine expects
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(&wo[w][r32][32 * d + 8 * g + 4 * hh]) = f32x4{O[d][4 * g], O[d][4 * g + 1], O[d][4 * g + 2], O[d][4 * g + 3]};
    }
  }
  G2V_STAMP(5);
  __syncthreads();
  G2V_STAMP(6);
  // ---- merge the four waves: one partial per (head, block)
  for (int idx = tid; idx < G * 128; idx += 256) {
    const int h = idx >> 7, d = idx & 127;
    float M = fmaxf(fmaxf(wm[0][h], wm[1][h]), fmaxf(wm[2][h], wm[3][h]));
    float L = 0.f, Ov = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = wm[k][h] == -INFINITY ? 0.f : __expf(wm[k][h] - M);
      L = fmaf(wl[k][h], f, L);
      Ov = fmaf(wo[k][h][d], f, Ov);
    }
    float* o = a.ws + (((size_t)z * Hq + kvh * G + h) * NBH + blockIdx.x) * 130;
    if (d == 0) { o[0] = M; o[1] = L; }
    o[2 + d] = Ov;
  }
  G2V_STAMP(7);
  G2V_STAMP_RT(11);
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

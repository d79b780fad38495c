// The batch-1 decode step as ONE kernel: 28 x {qkv, attention, combine, o, gate/up, down} + lm_head behind grid-wide
// barriers instead of kernel boundaries (reference loop body modeling/g2vlm/g2vlm.py:1088-1125).
//
// Why: the captured graph of decode_layer.hip runs ~170 dependent kernels per token; the three small ones of a layer cost
// 4.8 us each (launch + argument fetch + one memory round trip + reduce + store) and the two streaming ones pay a ~1.2 us
// ramp - the weight stream stops 170 times per token.  A phase boundary INSIDE one kernel costs 3.5 us including the
// activation hand-over (tools/grid_barrier_bench.py: two-level arrival, activations exchanged with agent-scope sc1 stores /
// loads instead of L2 write-back / invalidate fences, which alone cost 6 us), and - the point - the NEXT phase's weight
// loads are already in flight while the barrier is being crossed: weights do not depend on activations.
//
// Structure: 256 workgroups (one per CU: 8 waves holding > 128 VGPRs each cannot share a CU) stay resident for the whole
// step.  Every phase gives each workgroup an equal contiguous share of the rows (so each CU pulls 1/256 of the bytes), the
// arithmetic of every phase is the arithmetic of the per-phase kernels of decode_layer.hip - same dot order, same
// reductions, same rounding points - so the logits are BIT-IDENTICAL to that path (tests/test_kernels_gpu.py compares them).
// Data that crosses a barrier (residual stream, qkv row, attention partials, attention output, MLP activation) is written
// and read with sc1 accesses (xch_load / xch_store); weights, norm weights, the RoPE row and the KV cache rows of earlier
// steps come from earlier launches and take the normal path.  The new token's K / V row is appended by the one wave that
// owns it and used from registers in this step, as in the per-phase kernel.
//
// Barrier: groups of 32 workgroups (workgroup & 7: what an XCD holds under round-robin dispatch) count arrivals on their own
// word; the last arriver of a group adds 32 to the global word every waiter polls.  Words are zeroed by the host before each
// launch.  Every spin is bounded: a workgroup that gives up raises `err`, stops waiting at all later barriers, and the grid
// drains (the step's result is then garbage and the host falls back to the per-phase kernels).
#include "common.h"
#include "decode_util.h"
#include "decode_attn_pg.h"
#include "g2vlm_hip.h"

namespace {

struct MkLayer {
  const __bf16* qkv_w; const __bf16* qkv_b; const __bf16* o_w; const __bf16* gu_w; const __bf16* down_w;
  const float* ln1; const float* ln2; const float* qn; const float* kn;
  __bf16* kc; __bf16* vc;
};

struct MkArgs {
  const MkLayer* layers; int n_layers;
  float* x; __bf16* qkv; __bf16* ao; __bf16* act; float* ws;
  const float* cs; const float* sn; const int* Lk_dev;
  const float* fnorm; const __bf16* lm_head; __bf16* logits; int vocab;
  unsigned* bar; int* err; unsigned long long* dbg;
  int H, Hq, Hkv, F; float eps, scale; int und_rounding; long scene_rows; int cap, S, SW, nbh;
};

constexpr int SPIN_LIMIT = 1 << 21;

struct MkShared {
  AttnLds attn;
  float sm[2], sL[2];
  __attribute__((aligned(16))) float sf[128];
  float sO[8][128];
  int dead;
};

__device__ __forceinline__ unsigned long long rt_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__device__ __forceinline__ void grid_barrier(const MkArgs& a, MkShared& sh, unsigned& epoch) {
  const bool stamp = a.dbg && blockIdx.x == 0 && threadIdx.x == 0;
  ++epoch;
  if (stamp) a.dbg[4 * epoch - 3] = rt_now();               // work of the phase issued (wave 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's exchange stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0 && !sh.dead) {
    if (stamp) a.dbg[4 * epoch - 2] = rt_now();             // the whole workgroup is done
    const unsigned per = gridDim.x >> 3;
    unsigned* grp = a.bar + 32 + 32 * (blockIdx.x & 7);
    const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * per) __hip_atomic_fetch_add(a.bar, per, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamp) a.dbg[4 * epoch - 1] = rt_now();             // arrival acknowledged
    const unsigned target = epoch * gridDim.x;
    bool ok = false;
    for (int it = 0; it < SPIN_LIMIT; ++it) {
      if (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) { sh.dead = 1; *a.err = 1; }
    if (stamp) a.dbg[4 * epoch] = rt_now();                 // released
  }
  __syncthreads();
}

// ---- activation fragments (chunk lane + 64 j of the K axis, packed bf16 pairs), exactly gemv_pg_kernel's -----------------
template <int KCH>
__device__ __forceinline__ void x_frag_bf16(const __bf16* x, int nch, int lane, uint32_t (&xp)[KCH][4]) {
  u32x4 xv[KCH];
#pragma unroll
  for (int j = 0; j < KCH; ++j) xv[j] = xch_load<false, u32x4>(x, 16 * min(lane + 64 * j, nch - 1));
#pragma unroll
  for (int j = 0; j < KCH; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) xp[j][e] = lane + 64 * j < nch ? xv[j][e] : 0u;
}

template <int KCH>
__device__ __forceinline__ void x_frag_norm(const float* xf, const float* norm_w, int K, float eps, int lane, uint32_t (&xp)[KCH][4]) {
  const int nch = K >> 3;
  f32x4 a[KCH][2], nwv[KCH][2];
#pragma unroll
  for (int j = 0; j < KCH; ++j) {
    const int c = min(lane + 64 * j, nch - 1);
    a[j][0] = xch_load<false, f32x4>(xf, 32 * c);
    a[j][1] = xch_load<false, f32x4>(xf, 32 * c + 16);
    nwv[j][0] = *reinterpret_cast<const f32x4*>(norm_w + 8 * c);
    nwv[j][1] = *reinterpret_cast<const f32x4*>(norm_w + 8 * c + 4);
  }
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

// this wave's units of a phase with U units: workgroup b owns [b per, (b + 1) per), per = ceil(U / grid), cut over its 8 waves
__device__ __forceinline__ void wave_units(int U, int w, int& lo, int& hi) {
  const int per = (U + (int)gridDim.x - 1) / (int)gridDim.x;
  const int base = min((int)blockIdx.x * per, U), nb = min(per, U - base);
  lo = base + (w * nb) / 8;
  hi = base + ((w + 1) * nb) / 8;
}

// weights of one batch of RB units (ACT: gate + up row per unit) -> registers; issued as early as the caller can
template <bool ACT, int KCH, int RB>
__device__ __forceinline__ void w_issue(const __bf16* W, int K, int nch, int lane, int u0, int hi, u32x4 (&ww)[ACT ? 2 * RB : RB][KCH]) {
  const int nrow = min(RB, hi - u0);
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    if (r < nrow) {
#pragma unroll
      for (int h = 0; h < (ACT ? 2 : 1); ++h) {
        const int u = u0 + r;
        const int row = ACT ? 32 * (u >> 4) + (u & 15) + 16 * h : u;
        const u32x4* wp = reinterpret_cast<const u32x4*>(W + (size_t)row * K);
#pragma unroll
        for (int j = 0; j < KCH; ++j) ww[(ACT ? 2 * r + h : r)][j] = __builtin_nontemporal_load(wp + min(lane + 64 * j, nch - 1));
      }
    }
  }
}

// one batch: dots, reductions; lane r < nrow ends up with its unit's value(s)
template <bool ACT, int KCH, int RB>
__device__ __forceinline__ void w_dot(const u32x4 (&ww)[ACT ? 2 * RB : RB][KCH], const uint32_t (&xp)[KCH][4], int nrow, int lane, float& v0, float& v1) {
  constexpr int ROWS = ACT ? 2 * RB : RB;
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
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum_dpp(acc[r]);
  v0 = 0.f; v1 = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r)
    if (lane == r) { v0 = acc[ACT ? 2 * r : r]; v1 = ACT ? acc[2 * r + 1] : 0.f; }
}

__device__ __forceinline__ void xch_store_bf16(__bf16* base, int idx, float v) {
  const __bf16 b = f2bf(v);
  __builtin_amdgcn_raw_buffer_store_b16(*reinterpret_cast<const unsigned short*>(&b), xch_rsrc(base), 2 * idx, 0, 16);
}

// out[n] = bf16(bfround(W[n] . x + bias[n]))           (qkv, lm_head: x = RMSNorm(residual))
template <int RB>
__device__ __attribute__((noinline)) void phase_linear_norm(const MkArgs& a, const float* xin, const float* norm_w, const __bf16* W, const __bf16* bias, __bf16* out,
                                                  int N, bool out_xch, int lane, int w) {
  const int K = a.H, nch = K >> 3;
  int lo, hi;
  wave_units(N, w, lo, hi);
  u32x4 ww[RB][3];
  if (lo < hi) w_issue<false, 3, RB>(W, K, nch, lane, lo, hi, ww);
  if (lo >= hi) return;
  uint32_t xp[3][4];
  x_frag_norm<3>(xin, norm_w, K, a.eps, lane, xp);
  for (int u0 = lo; u0 < hi; u0 += RB) {
    const int nrow = min(RB, hi - u0);
    float v, unused;
    w_dot<false, 3, RB>(ww, xp, nrow, lane, v, unused);
    if (u0 + RB < hi) w_issue<false, 3, RB>(W, K, nch, lane, u0 + RB, hi, ww);
    if (lane < nrow) {
      const int n = u0 + lane;
      v = bfround(v + (bias ? bf2f(bias[n]) : 0.f));
      if (out_xch) xch_store_bf16(out, n, v);
      else out[n] = f2bf(v);
    }
  }
}

// x[n] += bfround(W[n] . xin)                           (o projection, down projection), xin bf16 [K]
template <int KCH>
__device__ __attribute__((noinline)) void phase_linear_res(const MkArgs& a, const __bf16* xin, const float* res_in, float* res_out, const __bf16* W, int N, int K, int lane,
                                                 int w) {
  const int nch = K >> 3;
  int lo, hi;
  wave_units(N, w, lo, hi);
  if (lo >= hi) return;
  u32x4 ww[1][KCH];
  w_issue<false, KCH, 1>(W, K, nch, lane, lo, hi, ww);
  uint32_t xp[KCH][4];
  x_frag_bf16<KCH>(xin, nch, lane, xp);
  for (int u0 = lo; u0 < hi; ++u0) {
    float v, unused;
    const float rcur = lane == 0 ? res_in[u0] : 0.f;
    w_dot<false, KCH, 1>(ww, xp, 1, lane, v, unused);
    if (u0 + 1 < hi) w_issue<false, KCH, 1>(W, K, nch, lane, u0 + 1, hi, ww);
    if (lane == 0) xch_store<true>(res_out, 4 * u0, rcur + bfround(v));
  }
}

// act[u] = bf16(bfround(silu(bfround(g))) * bfround(u)), (g, u) = the unit's gate / up rows . RMSNorm(x)
__device__ __attribute__((noinline)) void phase_gate_up(const MkArgs& a, const float* xin, const float* norm_w, const __bf16* W, __bf16* act, int lane, int w) {
  constexpr int RB = 5;
  const int K = a.H, nch = K >> 3;
  int lo, hi;
  wave_units(a.F, w, lo, hi);
  if (lo >= hi) return;
  u32x4 ww[2 * RB][3];
  w_issue<true, 3, RB>(W, K, nch, lane, lo, hi, ww);
  uint32_t xp[3][4];
  x_frag_norm<3>(xin, norm_w, K, a.eps, lane, xp);
  for (int u0 = lo; u0 < hi; u0 += RB) {
    const int nrow = min(RB, hi - u0);
    float g, u;
    w_dot<true, 3, RB>(ww, xp, nrow, lane, g, u);
    if (u0 + RB < hi) w_issue<true, 3, RB>(W, K, nch, lane, u0 + RB, hi, ww);
    if (lane < nrow) xch_store_bf16(act, u0 + lane, bfround(siluf_(bfround(g))) * bfround(u));
  }
}

// decode_combine_pg_kernel's arithmetic on 512 threads: thread (d, g) sums the groups g and g + 4 of 16 consecutive partials
__device__ __attribute__((noinline)) void phase_combine(const MkArgs& a, MkShared& sh, __bf16* ao, int h, int tid) {
  const int NBH = a.nbh, d = tid & 127, g = tid >> 7;
  const float* p = a.ws + (size_t)h * NBH * 130;
  float ov[2][16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int k = 0; k < 16; ++k) ov[q][k] = xch_load<true, float>(p, 4 * (min(16 * (g + 4 * q) + k, NBH - 1) * 130 + 2 + d));
  float m = -INFINITY, l = 0.f;
  if (tid < NBH) { m = xch_load<true, float>(p, 4 * (tid * 130)); l = xch_load<true, float>(p, 4 * (tid * 130 + 1)); }
  if (tid < 128) {
    float mx = m;
    mx = fmaxf(mx, dpp_f<0x128>(mx)); mx = fmaxf(mx, dpp_f<0x124>(mx)); mx = fmaxf(mx, dpp_f<0x122>(mx)); mx = fmaxf(mx, dpp_f<0x121>(mx));
    mx = fmaxf(fmaxf(readlane_f(mx, 0), readlane_f(mx, 16)), fmaxf(readlane_f(mx, 32), readlane_f(mx, 48)));
    if ((tid & 63) == 0) sh.sm[tid >> 6] = mx;
  }
  __syncthreads();
  const float M = fmaxf(sh.sm[0], sh.sm[1]);
  if (tid < 128) {
    const float f = m == -INFINITY ? 0.f : __expf(m - M);
    sh.sf[tid] = f;
    const float lw = wave_sum_dpp(l * f);
    if ((tid & 63) == 0) sh.sL[tid >> 6] = lw;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int gg = g + 4 * q;
    float O = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const f32x4 f4 = *reinterpret_cast<const f32x4*>(&sh.sf[16 * gg + 4 * k4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (16 * gg + 4 * k4 + e < NBH) O = fmaf(ov[q][4 * k4 + e], f4[e], O);
    }
    sh.sO[gg][d] = O;
  }
  __syncthreads();
  if (g == 0) {
    const float Lt = sh.sL[0] + sh.sL[1];
    float Ot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) Ot += sh.sO[k][d];
    xch_store_bf16(ao, h * 128 + d, Ot / Lt);
  }
}

__global__ __launch_bounds__(512) void decode_step_mk_kernel(MkArgs a) {
  __shared__ MkShared sh;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) sh.dead = 0;
  __syncthreads();
  unsigned epoch = 0;
  if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[0] = rt_now();
  const int nqkv = (a.Hq + 2 * a.Hkv) * 128, nq = a.Hq * 128;
  // Everything a LATER phase reads with ordinary (L2-cached) loads has its own address per layer and phase: an address is
  // written once per launch (sc1: through to memory) before any workgroup reads it, so no L2 can hold a stale copy, and of the 32
  // workgroups of an XCD only the first to ask goes to memory.  (Read with sc1 loads from one shared buffer instead, each of
  // 1536 waves pulled the 18 KB activation through the fabric: the down projection took 15.5 us.)
  for (int li = 0; li < a.n_layers; ++li) {
    const MkLayer L = a.layers[li];
    const float* x0 = a.x + (size_t)(2 * li) * a.H;          // residual stream entering the layer
    float* x1 = a.x + (size_t)(2 * li + 1) * a.H;            // after attention
    float* x2 = a.x + (size_t)(2 * li + 2) * a.H;            // after the MLP
    __bf16* qkv = a.qkv + (size_t)li * nqkv;
    __bf16* ao = a.ao + (size_t)li * nq;
    __bf16* act = a.act + (size_t)li * a.F;
    // ---- q / k / v
    phase_linear_norm<2>(a, x0, L.ln1, L.qkv_w, L.qkv_b, qkv, nqkv, true, lane, w);
    grid_barrier(a, sh, epoch);
    // ---- split-KV attention: workgroup -> (kv head, key share)
    if ((int)blockIdx.x < a.nbh * a.Hkv) {
      AttnArgs at{qkv, L.qn, L.kn, a.cs, a.sn, a.eps, a.und_rounding, L.kc, L.vc, a.ws, a.Lk_dev, a.Hq, a.Hkv, a.scale, a.scene_rows,
                  a.cap, a.S, a.SW};
      decode_attn_pg_body<true, false, 8>(at, sh.attn, (int)blockIdx.x % a.nbh, (int)blockIdx.x / a.nbh, 0, a.nbh, tid);
    }
    grid_barrier(a, sh, epoch);
    // ---- combine: one workgroup per query head (the partials are few and read once: sc1 loads from one buffer)
    if ((int)blockIdx.x < a.Hq) phase_combine(a, sh, ao, blockIdx.x, tid);
    grid_barrier(a, sh, epoch);
    // ---- o projection + residual
    phase_linear_res<3>(a, ao, x0, x1, L.o_w, a.H, nq, lane, w);
    grid_barrier(a, sh, epoch);
    // ---- gate / up + SwiGLU
    phase_gate_up(a, x1, L.ln2, L.gu_w, act, lane, w);
    grid_barrier(a, sh, epoch);
    // ---- down projection + residual
    phase_linear_res<18>(a, act, x1, x2, L.down_w, a.H, a.F, lane, w);
    grid_barrier(a, sh, epoch);
  }
  // ---- final norm + lm_head (the logits are read by the next launch: plain stores)
  phase_linear_norm<8>(a, a.x + (size_t)(2 * a.n_layers) * a.H, a.fnorm, a.lm_head, nullptr, a.logits, a.vocab, false, lane, w);
  if (a.dbg && blockIdx.x == 0 && tid == 0) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    a.dbg[4 * epoch + 1] = t;
  }
}

}  // namespace

static unsigned long long* g_mk_dbg = nullptr;
// diagnostic (tools/decode_mk_phases.py): device buffer of >= 4 * (6 * n_layers) + 2 words that workgroup 0 fills with 100 MHz stamps
extern "C" int g2v_debug_mk_stamps(void* buf) { g_mk_dbg = (unsigned long long*)buf; return 0; }

extern "C" int64_t g2v_decode_step_mk_layer_bytes(void) { return (int64_t)sizeof(MkLayer); }

// One decode step (every layer + lm_head) in one launch: see the header of this file.  `layers`: device array of n_layers
// records {qkv_w, qkv_b, o_w, gate_up_w, down_w, ln1, ln2, q_norm, k_norm, k_cache, v_cache} (11 pointers each);
// x f32 [(2 n_layers + 1) H]: row 0 holds the token's embedding, row v the residual stream after v half-layers; qkv / ao / act:
// one row PER LAYER ([n_layers] x (Hq + 2 Hkv) 128 / Hq 128 / F) - every hand-over has its own address within a launch;
// workspace >= g2v_decode_attn_pg_workspace bytes; barrier: >= 1280 bytes of device words ZEROED before
// every call; err: device int, raised when a workgroup gave up waiting (the result is then invalid).
// Shapes: H <= 1536, F <= 9216, head_dim 128; the grid is 256 workgroups and must be fully resident.
extern "C" int g2v_decode_step_mk(const void* layers, int n_layers, void* x, void* qkv, void* ao, void* act, void* workspace, const void* cos,
                                  const void* sin, const void* Lk_dev, const void* final_norm_w, const void* lm_head, void* logits, int vocab,
                                  void* barrier, void* err, int H, int Hq, int Hkv, int F, float eps, float scale, int und_rounding,
                                  int64_t scene_rows, int max_len, void* stream) {
  if (!layers || n_layers <= 0 || !x || !qkv || !ao || !act || !workspace || !cos || !sin || !Lk_dev || !final_norm_w || !lm_head || !logits ||
      !barrier || !err || vocab <= 0 || H <= 0 || (H & 7) || H > 1536 || F <= 0 || (F & 7) || F > 9216 || Hq <= 0 || Hkv <= 0 || Hq % Hkv ||
      Hq / Hkv > GMAX || Hq > 256 || Hq * 128 > 1536 || max_len <= 0 || scene_rows < max_len)
    return G2V_ERR_ARG;
  const int nbh = 256 / Hkv > 128 ? 128 : 256 / Hkv;         // as g2v_decode_attn_pg at batch 1
  if (nbh * Hkv > 256) return G2V_ERR_ARG;
  MkArgs a{(const MkLayer*)layers, n_layers, (float*)x, (__bf16*)qkv, (__bf16*)ao, (__bf16*)act, (float*)workspace, (const float*)cos,
           (const float*)sin, (const int*)Lk_dev, (const float*)final_norm_w, (const __bf16*)lm_head, (__bf16*)logits, vocab,
           (unsigned*)barrier, (int*)err, g_mk_dbg, H, Hq, Hkv, F, eps, scale, und_rounding, (long)scene_rows, max_len, (max_len + nbh - 1) / nbh,
           ((max_len + nbh - 1) / nbh + 3) / 4, nbh};
  hipLaunchKernelGGL(decode_step_mk_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

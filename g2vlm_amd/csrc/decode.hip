// Batch-1 decode kernels (HBM-bound): weight-streaming GEMV, split-KV GQA attention, SwiGLU.
// Replace the per-token path of G2VLM.generate_text (reference modeling/g2vlm/g2vlm.py:1086-1135):
// 28 x {q/k/v/o/gate/up/down Linear at M=1, flash_attn_varlen_func with q_len 1} + lm_head.
// Weights and K/V rows go straight to VGPRs with 16-byte loads (guide: "GEMV / M<=16: neither LDS nor
// glds"), one wave per output row, fp32 accumulation, one bf16 rounding at the Linear's output.
#include "common.h"
#include "g2vlm_hip.h"

namespace {

__device__ __forceinline__ float dot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc = fmaf(bits2f_lo(w[e]), bits2f_lo(x[e]), acc);
    acc = fmaf(bits2f_hi(w[e]), bits2f_hi(x[e]), acc);
  }
  return acc;
}

// Weight-streaming GEMV, one block (4 waves) per RB output rows, the 4 waves split K and reduce through LDS.
// The activation vector is produced on the fly from one of three sources so the tiny norm / activation kernels of
// the decode step disappear (they were ~85 of ~310 launches per token):
//   XMODE 0: x bf16[K]
//   XMODE 1: x = bf16( w_norm[k] * (res[k] * rsqrt(mean(res^2)+eps)) )   Qwen2RMSNorm of the fp32 residual stream
//   XMODE 2: x = bf16( bf16(silu(g[k])) * u[k] ) from the gate/up GEMV output (interleaved per 16)
// Weights are streamed once -> non-temporal loads (guide "nt-weights").
// ACT (RB == 8 only): W is a gate/up matrix interleaved per 16 rows; a block takes 4 gate rows and their 4 up rows and
// writes the 4 SwiGLU activations bf16(bf16(silu(g)) * u) directly, so the down-projection reads a ready vector instead
// of every one of its 768 blocks re-deriving all 8960 activations (that recomputation was 6 of its 12 us).
template <int XMODE, int RB, int U, bool ACT = false>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const void* xin, const float* norm_w, float eps, const __bf16* W,
                                                        const __bf16* bias, __bf16* out, float* res, int N, int K) {
  __shared__ float red[4][RB];
  __shared__ float s_rstd;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * RB;
  const int nch = K >> 3;
  // the first batch of weight loads goes out before anything else (the fused norm's reduction below then overlaps it)
  const __bf16* wrow[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    int n = n0 + r;
    if constexpr (ACT) n = 32 * (blockIdx.x >> 2) + 4 * (blockIdx.x & 3) + (r & 3) + 16 * (r >> 2);
    wrow[r] = W + (size_t)min(n, N - 1) * K;
  }
  float rstd = 1.f;
  if constexpr (XMODE == 1) {
    const float* xf = reinterpret_cast<const float*>(xin);
    float ss = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xf + k);
      ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[w][0] = ss;
    __syncthreads();
    if (tid == 0) s_rstd = 1.0f / sqrtf(((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) / (float)K + eps);
    __syncthreads();
    rstd = s_rstd;
  }
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
  // U chunks x RB rows of 16-byte weight loads are in flight per lane before the first FMA (the stream is latency-bound
  // otherwise: one load per lane per trip of a non-unrolled loop)
  for (int c0 = tid; c0 < nch; c0 += 256 * U) {
    u32x4 ww[U][RB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = min(c0 + 256 * u, nch - 1);
#pragma unroll
      for (int r = 0; r < RB; ++r) ww[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r]) + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = min(c0 + 256 * u, nch - 1);
      const bool live = c0 + 256 * u < nch;
      float xv[8];
      if constexpr (XMODE == 0) {
        u32x4 xx = reinterpret_cast<const u32x4*>(xin)[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[2 * e] = bits2f_lo(xx[e]); xv[2 * e + 1] = bits2f_hi(xx[e]); }
      } else if constexpr (XMODE == 1) {
        const float* xf = reinterpret_cast<const float*>(xin) + c * 8;
        f32x4 a = *reinterpret_cast<const f32x4*>(xf), b = *reinterpret_cast<const f32x4*>(xf + 4);
        f32x4 wa = *reinterpret_cast<const f32x4*>(norm_w + c * 8), wb = *reinterpret_cast<const f32x4*>(norm_w + c * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[e] = bfround(__fmul_rn(wa[e], __fmul_rn(a[e], rstd)));
          xv[4 + e] = bfround(__fmul_rn(wb[e], __fmul_rn(b[e], rstd)));
        }
      } else {
        const __bf16* gu = reinterpret_cast<const __bf16*>(xin);
        const int k0 = c * 8, blk = k0 >> 4, j = k0 & 15;           // 8 consecutive k inside one 16-block
        u32x4 gg = *reinterpret_cast<const u32x4*>(gu + 32 * blk + j);
        u32x4 uu = *reinterpret_cast<const u32x4*>(gu + 32 * blk + 16 + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[2 * e] = bfround(bfround(siluf_(bits2f_lo(gg[e]))) * bits2f_lo(uu[e]));
          xv[2 * e + 1] = bfround(bfround(siluf_(bits2f_hi(gg[e]))) * bits2f_hi(uu[e]));
        }
      }
      if (!live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[r] = fmaf(bits2f_lo(ww[u][r][e]), xv[2 * e], acc[r]);
          acc[r] = fmaf(bits2f_hi(ww[u][r][e]), xv[2 * e + 1], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    float s = wave_sum(acc[r]);
    if (lane == 0) red[w][r] = s;
  }
  __syncthreads();
  if constexpr (ACT) {
    if (tid < 4) {
      const float g = bfround((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
      const float u = bfround((red[0][tid + 4] + red[1][tid + 4]) + (red[2][tid + 4] + red[3][tid + 4]));
      out[16 * (blockIdx.x >> 2) + 4 * (blockIdx.x & 3) + tid] = f2bf(bfround(siluf_(g)) * u);
    }
  } else if (tid < RB && n0 + tid < N) {
    int n = n0 + tid;
    float v = bfround(((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + (bias ? bf2f(bias[n]) : 0.f));
    if (res) res[n] = res[n] + v;
    else out[n] = f2bf(v);
  }
}

__global__ void swiglu_bf16_kernel(const __bf16* gu, __bf16* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b = i >> 4, j = i & 15;                     // gate/up interleaved per 16 outputs (weights.interleave_gate_up)
  float g = bf2f(gu[32 * b + j]), u = bf2f(gu[32 * b + 16 + j]);
  out[i] = f2bf(bfround(siluf_(g)) * u);
}

// ---- split-KV decode attention, head_dim 128 -------------------------------------------------------
// grid (chunks, Hkv); block 256 = 4 waves; chunk = 64 keys; all G = Hq/Hkv query heads of the kv head
// share each K/V row read.  Partials (m, l, o[128]) per (q head, chunk) go to the fp32 workspace and
// are merged by decode_combine_kernel.
constexpr int DCH = 64, GMAX = 8;

// One WAVE per 64-key chunk, four chunks per block, no barrier anywhere: (1) every K and V load of the chunk is issued up
// front (32 x 16 B per lane); (2) scores S[head][key] by MFMA 16x16x32 - A = the kv head's G <= 8 query heads padded to 16
// rows, B = K rows read straight from global in fragment layout - so a lane owns one key and four heads; (3) the chunk's
// softmax statistics by 16-lane xor-shuffles; (4) p goes through a wave-private LDS strip to the lanes that hold the V rows
// (lane = 4 keys x 8 dims), fp32 FMAs, two xor-shuffles over the four key sub-groups.  Partials as before.
//
// Batched decode (gridDim.z scenes sharing the weights, one KV cache each): scene z reads q row z, the cache block at
// z * scene_rows rows, its own length Lk_dev[z] and its own workspace slab.
//
// FUSED: the step's q/k-norm + mRoPE + cache write (qknorm_mrope_cache_kernel, reference qwen2vl.py:596-634) happen here:
// `q` is the RAW fused qkv row [Hq + 2 Hkv heads] of the new token.  While its K/V chunk loads are in flight every wave
// normalises and rotates the G query heads of its kv head - 16 lanes per head, lane j holding dims 4j..4j+3 and their
// rotate_half partners 64+4j.., the arithmetic and reduction order of qknorm_mrope_cache_kernel, so the result is
// bit-identical to it - into a wave-private LDS strip and reads its MFMA A fragments from there; the one wave whose chunk
// holds the new token's cache row Lk - 1 does the same for the new k, takes the new v, uses both and appends them to the
// cache.  One launch and one dependent memory round trip less per layer than the separate kernel.
struct FusedArgs {
  const float* qw; const float* kw; const float* cs; const float* sn; float eps; int und_rounding;
};

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void decode_attn_kernel(const __bf16* q, __bf16* kc, __bf16* vc, float* ws,
                                                          int Lk_arg, const int* Lk_dev, int Hq, int Hkv, float scale,
                                                          long scene_rows, long ws_scene, FusedArgs fa) {
  // Lk comes from device memory when the step is replayed from a HIP graph (grid sized for the cache capacity)
  const int z = blockIdx.z;
  const int Lk = Lk_dev ? Lk_dev[z] : Lk_arg;
  q += (size_t)z * (FUSED ? (Hq + 2 * Hkv) : Hq) * 128;
  kc += (size_t)z * scene_rows * Hkv * 128;
  vc += (size_t)z * scene_rows * Hkv * 128;
  ws += (size_t)z * ws_scene;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chunk = blockIdx.x * 4 + w;
  if (chunk * DCH >= Lk) return;                           // wave-uniform; nothing below synchronises across waves
  __shared__ float sp[4][GMAX * DCH];
  const int G = Hq / Hkv, kvh = blockIdx.y, nchunks = (Lk + DCH - 1) / DCH;
  const int k0 = chunk * DCH, nk = min(DCH, Lk - k0);
  const int fr = lane & 15, fg = lane >> 4;
  const size_t row_stride = (size_t)Hkv * 128;
  // ---- all loads first
  bf16x8 qa[4], kf[4][4];
  [[maybe_unused]] u32x4 vnew;
  [[maybe_unused]] const int new_local = Lk - 1 - k0;      // the new token's key index inside this chunk, if 0 <= . < 64
  [[maybe_unused]] const bool has_new = new_local >= 0 && new_local < DCH;     // wave-uniform
  if constexpr (!FUSED) {
    const __bf16* qp = q + (size_t)(kvh * G + min(fr, G - 1)) * 128 + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
  } else {
    if (has_new) vnew = *reinterpret_cast<const u32x4*>(q + (size_t)(Hq + Hkv + kvh) * 128 + 8 * fr);
  }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const __bf16* kp = kc + (size_t)(k0 + min(16 * kb + fr, nk - 1)) * row_stride + kvh * 128 + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8*>(kp + 32 * ks);
  }
  u32x4 vv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    vv[i] = *reinterpret_cast<const u32x4*>(vc + (size_t)(k0 + min(4 * i + fg, nk - 1)) * row_stride + kvh * 128 + 8 * fr);
  if constexpr (FUSED) {
    __shared__ __attribute__((aligned(16))) __bf16 sq[4][GMAX + 1][128];     // per wave: G normalised q heads + the new k
    const float* cs = fa.cs + (size_t)z * 128;
    const float* sn = fa.sn + (size_t)z * 128;
    const int n_items = G + (has_new ? 1 : 0);
    const int j = lane & 15;
    for (int it0 = 0; it0 < n_items; it0 += 4) {           // 16 lanes per head, 4 heads per pass
      const int item = min(it0 + (lane >> 4), n_items - 1);  // surplus groups repeat the last item (same bytes)
      const bool isq = item < G;
      const __bf16* src = q + (size_t)(isq ? kvh * G + item : Hq + kvh) * 128 + 4 * j;
      const u32x2 a = *reinterpret_cast<const u32x2*>(src), b = *reinterpret_cast<const u32x2*>(src + 64);
      float x0[4] = {bits2f_lo(a[0]), bits2f_hi(a[0]), bits2f_lo(a[1]), bits2f_hi(a[1])};
      float x1[4] = {bits2f_lo(b[0]), bits2f_hi(b[0]), bits2f_lo(b[1]), bits2f_hi(b[1])};
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) ss += x0[e] * x0[e] + x1[e] * x1[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      const float rstd = 1.0f / sqrtf(ss / 128.f + fa.eps);
      const float* wp = (isq ? fa.qw : fa.kw) + 4 * j;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 64);
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 4 * j), c1 = *reinterpret_cast<const f32x4*>(cs + 64 + 4 * j);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sn + 4 * j), s1 = *reinterpret_cast<const f32x4*>(sn + 64 + 4 * j);
      float o0[4], o1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float n0 = __fmul_rn(x0[e], rstd), n1 = __fmul_rn(x1[e], rstd);
        if (fa.und_rounding) { n0 = bfround(n0); n1 = bfround(n1); }
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
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // the strip is written and read by this wave only
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(&sq[w][min(fr, G - 1)][32 * ks + 8 * fg]);
    if (has_new) {
      // the new row is the chunk's last key (nk - 1 == new_local); the loads above clamped every key >= nk to it and read
      // whatever the cache row held BEFORE this step (uninitialised memory: possibly NaN, and 0 x NaN would poison the P.V
      // sums of the masked keys).  So every lane whose key is >= new_local takes the fresh row: K row of key 16 kb + fr in
      // kf[kb], V row of key 4 i + fg in vv[i].
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        if (16 * kb + fr >= new_local) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = *reinterpret_cast<const bf16x8*>(&sq[w][G][32 * ks + 8 * fg]);
        }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (4 * i + fg >= new_local) vv[i] = vnew;
      if (fg == 0) *reinterpret_cast<u32x4*>(vc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 8 * fr) = vnew;
    }
  }
  // ---- scores: register r of lane (fr, fg) is S[head 4 fg + r][key 16 kb + fr]
  f32x4 S[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    S[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) S[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kf[kb][ks], S[kb], 0, 0, 0);
  }
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      S[kb][r] = (16 * kb + fr < nk) ? S[kb][r] * scale : -INFINITY;
      mx = fmaxf(mx, S[kb][r]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float p = expf(S[kb][r] - mx);                 // masked keys: exp(-inf) = 0
      S[kb][r] = p;
      sum += p;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    m[r] = mx; l[r] = sum;
    const int h = 4 * fg + r;
    if (h < G) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) sp[w][h * DCH + 16 * kb + fr] = S[kb][r];
      if (fr == 0) {
        float* o = ws + ((size_t)(kvh * G + h) * nchunks + chunk) * 130;
        o[0] = m[r]; o[1] = l[r];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                      // this wave's LDS writes are done; only this wave reads them
  __builtin_amdgcn_wave_barrier();
  // ---- o[h][d] = sum_key p[h][key] V[key][d]: lane = keys 4 i + fg, dims 8 fr .. 8 fr + 7
  float acc[GMAX][8];
#pragma unroll
  for (int h = 0; h < GMAX; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = 4 * i + fg;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bits2f_lo(vv[i][e]); v[2 * e + 1] = bits2f_hi(vv[i][e]); }
#pragma unroll
    for (int h = 0; h < GMAX; ++h)
      if (h < G) {
        const float p = sp[w][h * DCH + key];              // 0 for key >= nk
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = fmaf(p, v[e], acc[h][e]);
      }
  }
#pragma unroll
  for (int h = 0; h < GMAX; ++h)
    if (h < G) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = acc[h][e];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        acc[h][e] = a;
      }
      if (fg == 0) {
        float* o = ws + ((size_t)(kvh * G + h) * nchunks + chunk) * 130 + 2 + 8 * fr;
        o[0] = acc[h][0]; o[1] = acc[h][1]; o[2] = acc[h][2]; o[3] = acc[h][3];
        o[4] = acc[h][4]; o[5] = acc[h][5]; o[6] = acc[h][6]; o[7] = acc[h][7];
      }
    }
}

// NS blocks per q head (one per 128/NS-dim slice), NG = 8 NS chunk-groups each.  Two passes so that no load depends on a
// running maximum: (1) the global maximum of the chunk maxima, (2) every group sums its chunks g, g+NG, ... with weights
// exp(m_c - M), four independent partial loads in flight per thread; (3) the groups are added through LDS.  The kernel is
// a chain of dependent L2 round trips: fewer chunks per group took it from 6.5 us (NS = 1) to ~4.5 us (NS = 2) at 172 chunks.
template <int NS>
__global__ __launch_bounds__(1024) void decode_combine_kernel(const float* ws, __bf16* out, int Lk_arg, const int* Lk_dev, long ws_scene) {
  constexpr int DW = 128 / NS, NG = 1024 / DW;
  __shared__ float sred[16], sl[NG], so[NG * DW];
  const int z = blockIdx.y, part = blockIdx.z;
  const int nchunks = ((Lk_dev ? Lk_dev[z] : Lk_arg) + DCH - 1) / DCH;
  const int h = blockIdx.x, tid = threadIdx.x, dl = tid % DW, d = DW * part + dl, g = tid / DW;
  ws += (size_t)z * ws_scene;
  out += (size_t)z * gridDim.x * 128;
  const float* p = ws + (size_t)h * nchunks * 130;
  float mx = -INFINITY;
  for (int c = tid; c < nchunks; c += 1024) mx = fmaxf(mx, p[c * 130]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) sred[tid >> 6] = mx;
  __syncthreads();
  float M = sred[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) M = fmaxf(M, sred[k]);
  float l = 0.f, o = 0.f;
  int c = g;
  for (; c + 3 * NG < nchunks; c += 4 * NG) {
    float m0 = p[c * 130], m1 = p[(c + NG) * 130], m2 = p[(c + 2 * NG) * 130], m3 = p[(c + 3 * NG) * 130];
    float l0 = p[c * 130 + 1], l1 = p[(c + NG) * 130 + 1], l2 = p[(c + 2 * NG) * 130 + 1], l3 = p[(c + 3 * NG) * 130 + 1];
    float o0 = p[c * 130 + 2 + d], o1 = p[(c + NG) * 130 + 2 + d], o2 = p[(c + 2 * NG) * 130 + 2 + d], o3 = p[(c + 3 * NG) * 130 + 2 + d];
    float f0 = expf(m0 - M), f1 = expf(m1 - M), f2 = expf(m2 - M), f3 = expf(m3 - M);
    l = fmaf(l0, f0, l); o = fmaf(o0, f0, o);
    l = fmaf(l1, f1, l); o = fmaf(o1, f1, o);
    l = fmaf(l2, f2, l); o = fmaf(o2, f2, o);
    l = fmaf(l3, f3, l); o = fmaf(o3, f3, o);
  }
  for (; c < nchunks; c += NG) {
    float f = expf(p[c * 130] - M);
    l = fmaf(p[c * 130 + 1], f, l);
    o = fmaf(p[c * 130 + 2 + d], f, o);
  }
  if (dl == 0) sl[g] = l;
  so[g * DW + dl] = o;
  __syncthreads();
  if (g == 0) {
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) { L += sl[k]; O += so[k * DW + dl]; }
    out[h * 128 + d] = f2bf(O / L);
  }
}

}  // namespace

template <int XMODE>
static int gemv_launch(const void* x, const float* nw, float eps, const void* W, const void* bias, void* out, void* res, int N, int K,
                       hipStream_t s) {
  // rows per block x chunks per trip = 16-byte weight loads in flight per lane (8): wide-N projections take more rows
  // per block (the activation chunk is reused), long-K ones more chunks per trip
  const int nch = K >> 3;
  if (N >= 8192) hipLaunchKernelGGL((gemv_bf16_kernel<XMODE, 8, 1>), dim3((N + 7) / 8), dim3(256), 0, s, x, nw, eps, (const __bf16*)W,
                                    (const __bf16*)bias, (__bf16*)out, (float*)res, N, K);
  else if (nch > 1024) hipLaunchKernelGGL((gemv_bf16_kernel<XMODE, 2, 5>), dim3((N + 1) / 2), dim3(256), 0, s, x, nw, eps, (const __bf16*)W,
                                          (const __bf16*)bias, (__bf16*)out, (float*)res, N, K);   // K 8960: one trip of 5 x 256 chunks
  else if (nch > 256) hipLaunchKernelGGL((gemv_bf16_kernel<XMODE, 2, 4>), dim3((N + 1) / 2), dim3(256), 0, s, x, nw, eps, (const __bf16*)W,
                                         (const __bf16*)bias, (__bf16*)out, (float*)res, N, K);
  else hipLaunchKernelGGL((gemv_bf16_kernel<XMODE, 2, 1>), dim3((N + 1) / 2), dim3(256), 0, s, x, nw, eps, (const __bf16*)W,
                          (const __bf16*)bias, (__bf16*)out, (float*)res, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_gemv_bf16(const void* x, const void* W, const void* bias, void* out, void* res, int N, int K, void* stream) {
  if (!x || !W || (!out && !res) || N <= 0 || K <= 0 || (K & 7)) return G2V_ERR_ARG;
  return gemv_launch<0>(x, nullptr, 0.f, W, bias, out, res, N, K, (hipStream_t)stream);
}

extern "C" int g2v_gemv_rmsnorm_bf16(const void* x_f32, const void* norm_w, float eps, const void* W, const void* bias, void* out,
                                     int N, int K, void* stream) {
  if (!x_f32 || !norm_w || !W || !out || N <= 0 || K <= 0 || (K & 7)) return G2V_ERR_ARG;
  return gemv_launch<1>(x_f32, (const float*)norm_w, eps, W, bias, out, nullptr, N, K, (hipStream_t)stream);
}

extern "C" int g2v_gemv_rmsnorm_swiglu_bf16(const void* x_f32, const void* norm_w, float eps, const void* W_gu, void* act_out,
                                            int N2, int K, void* stream) {
  if (!x_f32 || !norm_w || !W_gu || !act_out || N2 <= 0 || (N2 & 31) || K <= 0 || (K & 7)) return G2V_ERR_ARG;
  hipLaunchKernelGGL((gemv_bf16_kernel<1, 8, 1, true>), dim3(N2 / 8), dim3(256), 0, (hipStream_t)stream, x_f32, (const float*)norm_w, eps,
                     (const __bf16*)W_gu, (const __bf16*)nullptr, (__bf16*)act_out, (float*)nullptr, N2, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_gemv_swiglu_bf16(const void* gu, const void* W, void* res, int N, int K, void* stream) {
  if (!gu || !W || !res || N <= 0 || K <= 0 || (K & 15)) return G2V_ERR_ARG;
  return gemv_launch<2>(gu, nullptr, 0.f, W, nullptr, nullptr, res, N, K, (hipStream_t)stream);
}

extern "C" int g2v_swiglu_bf16(const void* gu, void* out, int n, void* stream) {
  if (!gu || !out || n <= 0) return G2V_ERR_ARG;
  hipLaunchKernelGGL(swiglu_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const __bf16*)gu, (__bf16*)out, n);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int64_t g2v_decode_attn_workspace(int Lk, int Hq) { return (int64_t)Hq * ((Lk + DCH - 1) / DCH) * 130 * 4; }

static int decode_attn_launch(const void* q, const void* k_cache, const void* v_cache, void* out, int Lk, const int* Lk_dev,
                              int grid_chunks, int Hq, int Hkv, float scale, void* workspace, void* stream, int batch = 1,
                              long scene_rows = 0, const FusedArgs* fused = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  const long ws_scene = (long)Hq * grid_chunks * 130;      // floats per scene
  if (fused)
    hipLaunchKernelGGL(decode_attn_kernel<true>, dim3((grid_chunks + 3) / 4, Hkv, batch), dim3(256), 0, s, (const __bf16*)q,
                       (__bf16*)k_cache, (__bf16*)v_cache, (float*)workspace, Lk, Lk_dev, Hq, Hkv, scale, scene_rows, ws_scene, *fused);
  else
    hipLaunchKernelGGL(decode_attn_kernel<false>, dim3((grid_chunks + 3) / 4, Hkv, batch), dim3(256), 0, s, (const __bf16*)q,
                       (__bf16*)k_cache, (__bf16*)v_cache, (float*)workspace, Lk, Lk_dev, Hq, Hkv, scale, scene_rows, ws_scene, FusedArgs{});
  G2V_CHECK_LAUNCH();
  // slices per head: 4 for long caches (chunks per group is what the kernel's latency chain scales with), 2 otherwise
  if (grid_chunks > 64)
    hipLaunchKernelGGL(decode_combine_kernel<4>, dim3(Hq, batch, 4), dim3(1024), 0, s, (const float*)workspace, (__bf16*)out, Lk, Lk_dev, ws_scene);
  else
    hipLaunchKernelGGL(decode_combine_kernel<2>, dim3(Hq, batch, 2), dim3(1024), 0, s, (const float*)workspace, (__bf16*)out, Lk, Lk_dev, ws_scene);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* out, int Lk, int Hq, int Hkv,
                               float scale, void* workspace, void* stream) {
  if (!q || !k_cache || !v_cache || !out || !workspace || Lk <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || Hq / Hkv > GMAX)
    return G2V_ERR_ARG;
  return decode_attn_launch(q, k_cache, v_cache, out, Lk, nullptr, (Lk + DCH - 1) / DCH, Hq, Hkv, scale, workspace, stream);
}

// graph-replayable form: the KV length is read from device memory; the grid covers max_len keys and surplus chunks exit
extern "C" int g2v_decode_attn_dyn(const void* q, const void* k_cache, const void* v_cache, void* out, const void* Lk_dev,
                                   int max_len, int Hq, int Hkv, float scale, void* workspace, void* stream) {
  if (!q || !k_cache || !v_cache || !out || !workspace || !Lk_dev || max_len <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv ||
      Hq / Hkv > GMAX) return G2V_ERR_ARG;
  return decode_attn_launch(q, k_cache, v_cache, out, 0, (const int*)Lk_dev, (max_len + DCH - 1) / DCH, Hq, Hkv, scale, workspace,
                            stream);
}

// batched decode: `batch` scenes, q / out [batch, Hq*128], caches [batch, scene_rows, Hkv, 128], Lk_dev int32[batch];
// workspace >= batch * g2v_decode_attn_workspace(max_len, Hq) bytes.  The grid covers max_len keys of every scene.
extern "C" int g2v_decode_attn_batch(const void* q, const void* k_cache, const void* v_cache, void* out, const void* Lk_dev, int batch,
                                     int64_t scene_rows, int max_len, int Hq, int Hkv, float scale, void* workspace, void* stream) {
  if (!q || !k_cache || !v_cache || !out || !workspace || !Lk_dev || batch <= 0 || batch > 65535 || max_len <= 0 || scene_rows < max_len ||
      Hq <= 0 || Hkv <= 0 || Hq % Hkv || Hq / Hkv > GMAX) return G2V_ERR_ARG;
  return decode_attn_launch(q, k_cache, v_cache, out, 0, (const int*)Lk_dev, (max_len + DCH - 1) / DCH, Hq, Hkv, scale, workspace,
                            stream, batch, scene_rows);
}

// the decode step's attention with the q/k-norm, mRoPE and KV-cache append folded in (see decode_attn_kernel<true>):
// qkv = the step's RAW fused projections bf16 [batch, (Hq + 2 Hkv) * 128]; q_norm_w / k_norm_w f32 [128] (und expert);
// cos / sin f32 [batch, 128] (g2v_mrope_table of the step's positions); Lk_dev[b] = cache length INCLUDING the new token,
// whose K / V this call writes to row Lk_dev[b] - 1 of scene b's block.  Other arguments as g2v_decode_attn_batch.
extern "C" int g2v_decode_attn_fused(const void* qkv, const void* q_norm_w, const void* k_norm_w, float eps, int und_rounding,
                                     const void* cos, const void* sin, void* k_cache, void* v_cache, void* out, const void* Lk_dev,
                                     int batch, int64_t scene_rows, int max_len, int Hq, int Hkv, float scale, void* workspace,
                                     void* stream) {
  if (!qkv || !q_norm_w || !k_norm_w || !cos || !sin || !k_cache || !v_cache || !out || !workspace || !Lk_dev || batch <= 0 ||
      batch > 65535 || max_len <= 0 || scene_rows < max_len || Hq <= 0 || Hkv <= 0 || Hq % Hkv || Hq / Hkv > GMAX) return G2V_ERR_ARG;
  FusedArgs fa{(const float*)q_norm_w, (const float*)k_norm_w, (const float*)cos, (const float*)sin, eps, und_rounding};
  return decode_attn_launch(qkv, k_cache, v_cache, out, 0, (const int*)Lk_dev, (max_len + DCH - 1) / DCH, Hq, Hkv, scale, workspace,
                            stream, batch, scene_rows, &fa);
}

// per-token bookkeeping kept on the device so a captured step replays without host writes:
// state = {rope position (x3), cache row, kv length}; all advance by one
__global__ void decode_advance_kernel(int* pos3, int* row, int* len, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // pos3 is [3, batch]
  if (i < 3 * batch) pos3[i] += 1;
  if (i < batch) { row[i] += 1; len[i] += 1; }
}

extern "C" int g2v_decode_advance(void* pos3, void* row, void* len, void* stream) {
  if (!pos3 || !row || !len) return G2V_ERR_ARG;
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)pos3, (int*)row, (int*)len, 1);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_decode_advance_batch(void* pos3, void* row, void* len, int batch, void* stream) {
  if (!pos3 || !row || !len || batch <= 0) return G2V_ERR_ARG;
  hipLaunchKernelGGL(decode_advance_kernel, dim3((3 * batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, (int*)pos3, (int*)row,
                     (int*)len, batch);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

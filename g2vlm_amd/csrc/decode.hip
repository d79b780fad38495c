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
template <int XMODE, int RB>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const void* xin, const float* norm_w, float eps, const __bf16* W,
                                                        const __bf16* bias, __bf16* out, float* res, int N, int K) {
  __shared__ float red[4][RB];
  __shared__ float s_rstd;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * RB;
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
  const int nch = K >> 3;
  for (int c = tid; c < nch; c += 256) {
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
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      int n = min(n0 + r, N - 1);
      u32x4 ww = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + (size_t)n * K) + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[r] = fmaf(bits2f_lo(ww[e]), xv[2 * e], acc[r]);
        acc[r] = fmaf(bits2f_hi(ww[e]), xv[2 * e + 1], acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    float s = wave_sum(acc[r]);
    if (lane == 0) red[w][r] = s;
  }
  __syncthreads();
  if (tid < RB && n0 + tid < N) {
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

__global__ __launch_bounds__(256) void decode_attn_kernel(const __bf16* q, const __bf16* kc, const __bf16* vc, float* ws,
                                                          int Lk_arg, const int* Lk_dev, int Hq, int Hkv, float scale) {
  // Lk comes from device memory when the step is replayed from a HIP graph (grid sized for the cache capacity)
  const int Lk = Lk_dev ? *Lk_dev : Lk_arg;
  if ((int)blockIdx.x * DCH >= Lk) return;
  __shared__ float sq[GMAX * 128];
  __shared__ float sp[GMAX * DCH];
  __shared__ float so[4 * GMAX * 128];
  const int G = Hq / Hkv, kvh = blockIdx.y, chunk = blockIdx.x, nchunks = (Lk + DCH - 1) / DCH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int k0 = chunk * DCH, nk = min(DCH, Lk - k0);
  for (int i = tid; i < G * 128; i += 256) sq[i] = bf2f(q[(size_t)(kvh * G) * 128 + i]);
  __syncthreads();
  // phase 1: scores.  4 lanes per key (32 dims each), 16 keys per wave, 64 per block
  {
    int key = w * 16 + (lane >> 2), sub = lane & 3;
    float acc[GMAX];
#pragma unroll
    for (int h = 0; h < GMAX; ++h) acc[h] = 0.f;
    if (key < nk) {
      const u32x4* kr = reinterpret_cast<const u32x4*>(kc + ((size_t)(k0 + key) * Hkv + kvh) * 128 + sub * 32);
      u32x4 kv[4] = {kr[0], kr[1], kr[2], kr[3]};
#pragma unroll
      for (int h = 0; h < GMAX; ++h) {
        if (h < G) {
          const float* qq = sq + h * 128 + sub * 32;
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a = fmaf(bits2f_lo(kv[c][e]), qq[c * 8 + e * 2], a);
              a = fmaf(bits2f_hi(kv[c][e]), qq[c * 8 + e * 2 + 1], a);
            }
          acc[h] = a;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < GMAX; ++h) {
      float a = acc[h];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      if (h < G && sub == 0) sp[h * DCH + key] = key < nk ? a * scale : -INFINITY;
    }
  }
  __syncthreads();
  // phase 2: per-head softmax statistics over the chunk (wave per head)
  for (int h = w; h < G; h += 4) {
    float s = sp[h * DCH + lane];
    float m = wave_max(s);
    float p = lane < nk ? expf(s - m) : 0.f;
    float l = wave_sum(p);
    sp[h * DCH + lane] = p;
    if (lane == 0) {
      float* o = ws + ((size_t)(kvh * G + h) * nchunks + chunk) * 130;
      o[0] = m; o[1] = l;
    }
  }
  __syncthreads();
  // phase 3: o[h][d] = sum_key p[h][key] V[key][d]; wave w takes keys w, w+4, ...; lane = 2 dims
  {
    float acc[GMAX][2];
#pragma unroll
    for (int h = 0; h < GMAX; ++h) acc[h][0] = acc[h][1] = 0.f;
    for (int key = w; key < nk; key += 4) {
      uint32_t vv = *reinterpret_cast<const uint32_t*>(vc + ((size_t)(k0 + key) * Hkv + kvh) * 128 + lane * 2);
      float v0 = bits2f_lo(vv), v1 = bits2f_hi(vv);
#pragma unroll
      for (int h = 0; h < GMAX; ++h)
        if (h < G) { float p = sp[h * DCH + key]; acc[h][0] = fmaf(p, v0, acc[h][0]); acc[h][1] = fmaf(p, v1, acc[h][1]); }
    }
#pragma unroll
    for (int h = 0; h < GMAX; ++h)
      if (h < G) { so[(w * GMAX + h) * 128 + lane * 2] = acc[h][0]; so[(w * GMAX + h) * 128 + lane * 2 + 1] = acc[h][1]; }
  }
  __syncthreads();
  for (int i = tid; i < G * 128; i += 256) {
    int h = i >> 7, d = i & 127;
    float v = (so[(0 * GMAX + h) * 128 + d] + so[(1 * GMAX + h) * 128 + d]) + (so[(2 * GMAX + h) * 128 + d] + so[(3 * GMAX + h) * 128 + d]);
    ws[((size_t)(kvh * G + h) * nchunks + chunk) * 130 + 2 + d] = v;
  }
}

// one block per q head: 8 chunk-groups x 128 dims; each group merges chunks g, g+8, ... online, then the 8 groups merge
__global__ __launch_bounds__(1024) void decode_combine_kernel(const float* ws, __bf16* out, int Lk_arg, const int* Lk_dev) {
  __shared__ float sm[8], sl[8], so[8 * 128];
  const int nchunks = ((Lk_dev ? *Lk_dev : Lk_arg) + DCH - 1) / DCH;
  const int h = blockIdx.x, d = threadIdx.x & 127, g = threadIdx.x >> 7;
  const float* p = ws + (size_t)h * nchunks * 130;
  float M = -INFINITY, l = 0.f, o = 0.f;
  for (int c = g; c < nchunks; c += 8) {
    float m = p[c * 130];
    float Mn = fmaxf(M, m);
    float fo = expf(M - Mn), fn = expf(m - Mn);
    l = l * fo + p[c * 130 + 1] * fn;
    o = o * fo + p[c * 130 + 2 + d] * fn;
    M = Mn;
  }
  if (d == 0) { sm[g] = M; sl[g] = l; }
  so[g * 128 + d] = o;
  __syncthreads();
  if (g == 0) {
    float Mt = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) Mt = fmaxf(Mt, sm[k]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float f = sm[k] == -INFINITY ? 0.f : expf(sm[k] - Mt);
      L = fmaf(sl[k], f, L);
      O = fmaf(so[k * 128 + d], f, O);
    }
    out[h * 128 + d] = f2bf(O / L);
  }
}

}  // namespace

template <int XMODE>
static int gemv_launch(const void* x, const float* nw, float eps, const void* W, const void* bias, void* out, void* res, int N, int K,
                       hipStream_t s) {
  // rows per block: keep >= ~4 blocks per CU in flight for the narrow projections, amortise x for the wide ones
  if (N >= 8192) hipLaunchKernelGGL((gemv_bf16_kernel<XMODE, 4>), dim3((N + 3) / 4), dim3(256), 0, s, x, nw, eps, (const __bf16*)W,
                                    (const __bf16*)bias, (__bf16*)out, (float*)res, N, K);
  else hipLaunchKernelGGL((gemv_bf16_kernel<XMODE, 1>), dim3(N), dim3(256), 0, s, x, nw, eps, (const __bf16*)W, (const __bf16*)bias,
                          (__bf16*)out, (float*)res, N, K);
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
                              int grid_chunks, int Hq, int Hkv, float scale, void* workspace, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(decode_attn_kernel, dim3(grid_chunks, Hkv), dim3(256), 0, s, (const __bf16*)q, (const __bf16*)k_cache,
                     (const __bf16*)v_cache, (float*)workspace, Lk, Lk_dev, Hq, Hkv, scale);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_combine_kernel, dim3(Hq), dim3(1024), 0, s, (const float*)workspace, (__bf16*)out, Lk, Lk_dev);
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

// per-token bookkeeping kept on the device so a captured step replays without host writes:
// state = {rope position (x3), cache row, kv length}; all advance by one
__global__ void decode_advance_kernel(int* pos3, int* row, int* len) {
  if (threadIdx.x < 3) pos3[threadIdx.x] += 1;
  if (threadIdx.x == 3) row[0] += 1;
  if (threadIdx.x == 4) len[0] += 1;
}

extern "C" int g2v_decode_advance(void* pos3, void* row, void* len, void* stream) {
  if (!pos3 || !row || !len) return G2V_ERR_ARG;
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)pos3, (int*)row, (int*)len);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

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

// one wave per output row; 4 waves per block
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const __bf16* x, const __bf16* W, const __bf16* bias, __bf16* out,
                                                        float* res, int N, int K) {
  int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  int lane = threadIdx.x & 63;
  const u32x4* wr = reinterpret_cast<const u32x4*>(W + (size_t)n * K);
  const u32x4* xr = reinterpret_cast<const u32x4*>(x);
  const int nch = K >> 3;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int c = lane;
  for (; c + 192 < nch; c += 256) {
    u32x4 w0 = wr[c], w1 = wr[c + 64], w2 = wr[c + 128], w3 = wr[c + 192];
    a0 = dot8(w0, xr[c], a0); a1 = dot8(w1, xr[c + 64], a1);
    a2 = dot8(w2, xr[c + 128], a2); a3 = dot8(w3, xr[c + 192], a3);
  }
  for (; c < nch; c += 64) a0 = dot8(wr[c], xr[c], a0);
  float s = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) {
    float v = bfround(s + (bias ? bf2f(bias[n]) : 0.f));
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

extern "C" int g2v_gemv_bf16(const void* x, const void* W, const void* bias, void* out, void* res, int N, int K, void* stream) {
  if (!x || !W || (!out && !res) || N <= 0 || K <= 0 || (K & 7)) return G2V_ERR_ARG;
  hipLaunchKernelGGL(gemv_bf16_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)W,
                     (const __bf16*)bias, (__bf16*)out, (float*)res, N, K);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
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

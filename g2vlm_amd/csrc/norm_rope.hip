// Row-wise fp32 norms, rotary embeddings and the fused qk-norm + mRoPE + KV-cache write.
// All of these are HBM-bound streaming kernels: one wave per row (or per row x head), 16-byte or
// 8-byte vector accesses, statistics in fp32, arithmetic order kept as the reference's eager ops
// (no fma contraction where the reference rounds between the multiply and the add).
#include "common.h"
#include "g2vlm_hip.h"

namespace {


template <bool IN_BF16, int MAXV>
__device__ __forceinline__ int load_row(const void* x, size_t row_off, int C, int lane, f32x4 (&v)[MAXV]) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
      if constexpr (IN_BF16) {
        u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(x) + row_off + c);
        v[i] = f32x4{bits2f_lo(w[0]), bits2f_hi(w[0]), bits2f_lo(w[1]), bits2f_hi(w[1])};
      } else {
        v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + row_off + c);
      }
      n = i + 1;
    }
  }
  return n;
}

template <bool OUT_BF16>
__device__ __forceinline__ void store4(void* out, size_t off, f32x4 y) {
  if constexpr (OUT_BF16) {
    u32x2 w = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(out) + off) = w;
  } else {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + off) = y;
  }
}

// MAXV = float4 groups per lane (C <= 256 * MAXV): sized to the row so the kernel keeps 8 waves/SIMD in flight
template <bool IN_BF16, bool OUT_BF16, int MAXV>
__global__ __launch_bounds__(256, 6) void layernorm_kernel(const void* x, int ldx, const float* w, const float* b, float eps,
                                                        void* out, int ldo, int M, int C) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int lane = threadIdx.x & 63;
  f32x4 v[MAXV];
  load_row<IN_BF16, MAXV>(x, (size_t)row * ldx, C, lane, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; q += d * d; }
    }
  }
  float var = wave_sum(q) / (float)C;
  float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < C) {
      f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = __fadd_rn(__fmul_rn(__fmul_rn(v[i][e] - mean, rstd), ww[e]), bb[e]);
      store4<OUT_BF16>(out, (size_t)row * ldo + c, y);
    }
  }
}

template <bool OUT_BF16, int MAXV>
__global__ __launch_bounds__(256, 6) void rmsnorm_kernel(const float* x, int ldx, const float* w_lo, const float* w_hi, int split,
                                                      float eps, void* out, int ldo, int M, int C) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int lane = threadIdx.x & 63;
  const float* w = row < split ? w_lo : w_hi;
  f32x4 v[MAXV];
  load_row<false, MAXV>(x, (size_t)row * ldx, C, lane, v);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < C) {
      f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = __fmul_rn(ww[e], __fmul_rn(v[i][e], rstd));
      store4<OUT_BF16>(out, (size_t)row * ldo + c, y);
    }
  }
}

// cos/sin [L,128]: dim d (and d+64) takes axis t for d<16, h for d<40, w otherwise
__global__ void mrope_table_kernel(const int* pos, int L, const float* inv_freq, float* cs, float* sn) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * 64) return;
  int r = i >> 6, d = i & 63;
  int axis = d < 16 ? 0 : (d < 40 ? 1 : 2);
  float f = __fmul_rn((float)pos[axis * L + r], inv_freq[d]);
  float c = cosf(f), s = sinf(f);
  cs[r * 128 + d] = c; cs[r * 128 + 64 + d] = c;
  sn[r * 128 + d] = s; sn[r * 128 + 64 + d] = s;
}

// 16 lanes per (row, head): lane j holds dims 4j..4j+3 and 64+4j..64+4j+3 (the rotate_half partners), so every access is
// 8 bytes and the RMS reduction is 4 xor-shuffles inside the 16-lane group; one wave = 4 (row, head) items.
// heads [0,Hq) = q, [Hq,Hq+Hkv) = k (norm + rope), [Hq+Hkv, Hq+2Hkv) = v (copy)
__global__ __launch_bounds__(256) void qknorm_mrope_cache_kernel(
    const __bf16* qkv, int L, int Hq, int Hkv, const float* qw_lo, const float* qw_hi, const float* kw_lo,
    const float* kw_hi, int split, float eps, int und_rounding, const float* cs, const float* sn, __bf16* q_out,
    __bf16* k_cache, __bf16* v_cache, const int* kv_rows) {
  const int nh = Hq + 2 * Hkv;
  const long item = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool live = item < (long)L * nh;
  const long it = live ? item : 0;
  const int row = (int)(it / nh), h = (int)(it - (long)row * nh);
  const int j = threadIdx.x & 15;
  const __bf16* src = qkv + (size_t)row * nh * 128 + h * 128 + 4 * j;
  u32x2 a = *reinterpret_cast<const u32x2*>(src), b = *reinterpret_cast<const u32x2*>(src + 64);
  if (h >= Hq + Hkv) {
    if (live) {
      __bf16* dst = v_cache + ((size_t)kv_rows[row] * Hkv + (h - Hq - Hkv)) * 128 + 4 * j;
      *reinterpret_cast<u32x2*>(dst) = a; *reinterpret_cast<u32x2*>(dst + 64) = b;
    }
    return;
  }
  float x0[4] = {bits2f_lo(a[0]), bits2f_hi(a[0]), bits2f_lo(a[1]), bits2f_hi(a[1])};
  float x1[4] = {bits2f_lo(b[0]), bits2f_hi(b[0]), bits2f_lo(b[1]), bits2f_hi(b[1])};
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) ss += x0[e] * x0[e] + x1[e] * x1[e];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rstd = 1.0f / sqrtf(ss / 128.f + eps);
  const bool isq = h < Hq;
  const float* w = (isq ? (row < split ? qw_lo : qw_hi) : (row < split ? kw_lo : kw_hi)) + 4 * j;
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(w), w1 = *reinterpret_cast<const f32x4*>(w + 64);
  const float* cp = cs + (size_t)row * 128 + 4 * j; const float* sp = sn + (size_t)row * 128 + 4 * j;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 64);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 64);
  float o0[4], o1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float n0 = __fmul_rn(x0[e], rstd), n1 = __fmul_rn(x1[e], rstd);
    if (und_rounding) { n0 = bfround(n0); n1 = bfround(n1); }
    n0 = __fmul_rn(w0[e], n0); n1 = __fmul_rn(w1[e], n1);
    // q*cos + rotate_half(q)*sin ; rotate_half = (-x2, x1)
    o0[e] = __fadd_rn(__fmul_rn(n0, c0[e]), __fmul_rn(-n1, s0[e]));
    o1[e] = __fadd_rn(__fmul_rn(n1, c1[e]), __fmul_rn(n0, s1[e]));
  }
  if (live) {
    __bf16* dst = (isq ? q_out + ((size_t)row * Hq + h) * 128 : k_cache + ((size_t)kv_rows[row] * Hkv + (h - Hq)) * 128) + 4 * j;
    *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3])};
    *reinterpret_cast<u32x2*>(dst + 64) = u32x2{pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
  }
}

// RoPE2D in bf16 arithmetic: out = bf16(bf16(t*cos) + bf16(rot(t)*sin)) per axis half
__global__ void rope2d_kernel(__bf16* x, int ld, int M, int col0, int n_heads, int D, const __bf16* cs, const __bf16* sn,
                              const int* pos, int P) {
  const int half = D / 2, quarter = D / 4;       // per-axis dim, pairs per axis
  int per_row = n_heads * half;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * per_row) return;
  int row = (int)(i / per_row), r = (int)(i - (long)row * per_row);
  int h = r / half, e = r - h * half;
  int axis = e / quarter, j = e - axis * quarter;
  int p = pos[(row % P) * 2 + axis];
  __bf16* t = x + (size_t)row * ld + col0 + h * D + axis * half;
  float t0 = bf2f(t[j]), t1 = bf2f(t[j + quarter]);
  float c0 = bf2f(cs[p * half + j]), c1 = bf2f(cs[p * half + j + quarter]);
  float s0 = bf2f(sn[p * half + j]), s1 = bf2f(sn[p * half + j + quarter]);
  float o0 = bfround(bfround(t0 * c0) + bfround(-t1 * s0));
  float o1 = bfround(bfround(t1 * c1) + bfround(t0 * s1));
  t[j] = f2bf(o0); t[j + quarter] = f2bf(o1);
}

// the same with four rotation pairs per lane (8-byte accesses instead of 2-byte ones); needs D % 16 == 0
__global__ void rope2d_vec4_kernel(__bf16* x, int ld, int M, int col0, int n_heads, int D, const __bf16* cs, const __bf16* sn,
                                   const int* pos, int P) {
  const int half = D / 2, quarter = D / 4, groups = quarter / 4;      // 4-pair groups per axis
  const int per_row = n_heads * 2 * groups;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * per_row) return;
  const int row = (int)(i / per_row), r = (int)(i - (long)row * per_row);
  const int h = r / (2 * groups), e = r - h * 2 * groups;
  const int axis = e / groups, j = (e - axis * groups) * 4;
  const int p = pos[(row % P) * 2 + axis];
  __bf16* t = x + (size_t)row * ld + col0 + h * D + axis * half + j;
  const u32x2 a = *reinterpret_cast<const u32x2*>(t), b = *reinterpret_cast<const u32x2*>(t + quarter);
  const u32x2 ca = *reinterpret_cast<const u32x2*>(cs + p * half + j), cb = *reinterpret_cast<const u32x2*>(cs + p * half + j + quarter);
  const u32x2 sa = *reinterpret_cast<const u32x2*>(sn + p * half + j), sb = *reinterpret_cast<const u32x2*>(sn + p * half + j + quarter);
  float t0[4] = {bits2f_lo(a[0]), bits2f_hi(a[0]), bits2f_lo(a[1]), bits2f_hi(a[1])};
  float t1[4] = {bits2f_lo(b[0]), bits2f_hi(b[0]), bits2f_lo(b[1]), bits2f_hi(b[1])};
  float c0[4] = {bits2f_lo(ca[0]), bits2f_hi(ca[0]), bits2f_lo(ca[1]), bits2f_hi(ca[1])};
  float c1[4] = {bits2f_lo(cb[0]), bits2f_hi(cb[0]), bits2f_lo(cb[1]), bits2f_hi(cb[1])};
  float s0[4] = {bits2f_lo(sa[0]), bits2f_hi(sa[0]), bits2f_lo(sa[1]), bits2f_hi(sa[1])};
  float s1[4] = {bits2f_lo(sb[0]), bits2f_hi(sb[0]), bits2f_lo(sb[1]), bits2f_hi(sb[1])};
  float o0[4], o1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o0[k] = bfround(bfround(t0[k] * c0[k]) + bfround(-t1[k] * s0[k]));
    o1[k] = bfround(bfround(t1[k] * c1[k]) + bfround(t0[k] * s1[k]));
  }
  *reinterpret_cast<u32x2*>(t) = u32x2{pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3])};
  *reinterpret_cast<u32x2*>(t + quarter) = u32x2{pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
}

// Qwen2-VL ViT rope (apply_rotary_pos_emb_vision, modeling_qwen2_vl.py:235-246): fp32 math on bf16 q,k
__global__ void rope_vision_kernel(__bf16* x, int ld, int M, int n_heads, int D, const float* cs, const float* sn) {
  const int half = D / 2;
  int per_row = n_heads * half;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * per_row) return;
  int row = (int)(i / per_row), r = (int)(i - (long)row * per_row);
  int h = r / half, j = r - h * half;
  __bf16* t = x + (size_t)row * ld + h * D;
  float t0 = bf2f(t[j]), t1 = bf2f(t[j + half]);
  const float* c = cs + (size_t)row * D; const float* s = sn + (size_t)row * D;
  float o0 = __fadd_rn(__fmul_rn(t0, c[j]), __fmul_rn(-t1, s[j]));
  float o1 = __fadd_rn(__fmul_rn(t1, c[j + half]), __fmul_rn(t0, s[j + half]));
  t[j] = f2bf(o0); t[j + half] = f2bf(o1);
}

}  // namespace

extern "C" int g2v_layernorm(const void* x, int x_dtype, int ldx, const void* w, const void* b, float eps, void* out,
                             int out_dtype, int ldo, int M, int C, void* stream) {
  if (!x || !w || !b || !out || C <= 0 || C > 2048 || (C & 3) || (ldx & 3) || (ldo & 3) || M < 0) return G2V_ERR_ARG;
  if (M == 0) return G2V_OK;
  dim3 grid((M + 3) / 4), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const float* wf = (const float*)w; const float* bf = (const float*)b;
#define LN_LAUNCH(IB, OB, MV) hipLaunchKernelGGL((layernorm_kernel<IB, OB, MV>), grid, blk, 0, s, x, ldx, wf, bf, eps, out, ldo, M, C)
#define LN_BY_C(IB, OB) do { if (C <= 512) LN_LAUNCH(IB, OB, 2); else if (C <= 1024) LN_LAUNCH(IB, OB, 4); else if (C <= 1536) LN_LAUNCH(IB, OB, 6); else LN_LAUNCH(IB, OB, 8); } while (0)
  if (x_dtype == G2V_F32 && out_dtype == G2V_BF16) LN_BY_C(false, true);
  else if (x_dtype == G2V_F32 && out_dtype == G2V_F32) LN_BY_C(false, false);
  else if (x_dtype == G2V_BF16 && out_dtype == G2V_BF16) LN_BY_C(true, true);
  else if (x_dtype == G2V_BF16 && out_dtype == G2V_F32) LN_BY_C(true, false);
  else return G2V_ERR_ARG;
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_rmsnorm(const void* x, int ldx, const void* w_lo, const void* w_hi, int split, float eps, void* out,
                           int out_dtype, int ldo, int M, int C, void* stream) {
  if (!x || !w_lo || !w_hi || !out || C <= 0 || C > 2048 || (C & 3) || (ldx & 3) || (ldo & 3) || M < 0) return G2V_ERR_ARG;
  if (M == 0) return G2V_OK;
  dim3 grid((M + 3) / 4), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define RMS_LAUNCH(OB, MV) hipLaunchKernelGGL((rmsnorm_kernel<OB, MV>), grid, blk, 0, s, (const float*)x, ldx, (const float*)w_lo, (const float*)w_hi, split, eps, out, ldo, M, C)
#define RMS_BY_C(OB) do { if (C <= 512) RMS_LAUNCH(OB, 2); else if (C <= 1024) RMS_LAUNCH(OB, 4); else if (C <= 1536) RMS_LAUNCH(OB, 6); else RMS_LAUNCH(OB, 8); } while (0)
  if (out_dtype == G2V_BF16) RMS_BY_C(true);
  else if (out_dtype == G2V_F32) RMS_BY_C(false);
  else return G2V_ERR_ARG;
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_mrope_table(const void* pos, int L, const void* inv_freq, void* cos, void* sin, void* stream) {
  if (!pos || !inv_freq || !cos || !sin || L < 0) return G2V_ERR_ARG;
  if (L == 0) return G2V_OK;
  hipLaunchKernelGGL(mrope_table_kernel, dim3((L * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const int*)pos, L,
                     (const float*)inv_freq, (float*)cos, (float*)sin);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_qknorm_mrope_cache(const void* qkv, int L, int Hq, int Hkv, const void* qw_lo, const void* qw_hi,
                                      const void* kw_lo, const void* kw_hi, int split, float eps, int und_rounding,
                                      const void* cos, const void* sin, void* q_out, void* k_cache, void* v_cache,
                                      const void* kv_rows, void* stream) {
  if (!qkv || !qw_lo || !qw_hi || !kw_lo || !kw_hi || !cos || !sin || !q_out || !k_cache || !v_cache || !kv_rows ||
      L < 0 || Hq <= 0 || Hkv <= 0) return G2V_ERR_ARG;
  if (L == 0) return G2V_OK;
  long items = (long)L * (Hq + 2 * Hkv);
  hipLaunchKernelGGL(qknorm_mrope_cache_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)qkv, L, Hq, Hkv, (const float*)qw_lo, (const float*)qw_hi, (const float*)kw_lo,
                     (const float*)kw_hi, split, eps, und_rounding, (const float*)cos, (const float*)sin, (__bf16*)q_out,
                     (__bf16*)k_cache, (__bf16*)v_cache, (const int*)kv_rows);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_rope2d(void* x, int ld, int M, int col0, int n_heads, int D, const void* cos, const void* sin,
                          const void* pos, int P, void* stream) {
  if (!x || !cos || !sin || !pos || M < 0 || n_heads <= 0 || D <= 0 || (D & 3) || P <= 0) return G2V_ERR_ARG;
  if (M == 0) return G2V_OK;
  long n = (long)M * n_heads * (D / 2);
  if ((D & 15) == 0 && (ld & 3) == 0 && (col0 & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 &&
      ((reinterpret_cast<uintptr_t>(cos) | reinterpret_cast<uintptr_t>(sin)) & 7) == 0) {
    n /= 4;
    hipLaunchKernelGGL(rope2d_vec4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (__bf16*)x, ld, M,
                       col0, n_heads, D, (const __bf16*)cos, (const __bf16*)sin, (const int*)pos, P);
    G2V_CHECK_LAUNCH();
    return G2V_OK;
  }
  hipLaunchKernelGGL(rope2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (__bf16*)x, ld, M, col0,
                     n_heads, D, (const __bf16*)cos, (const __bf16*)sin, (const int*)pos, P);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_rope_vision(void* x, int ld, int M, int n_heads, int D, const void* cos, const void* sin, void* stream) {
  if (!x || !cos || !sin || M < 0 || n_heads <= 0 || D <= 0 || (D & 1)) return G2V_ERR_ARG;
  if (M == 0) return G2V_OK;
  long n = (long)M * n_heads * (D / 2);
  hipLaunchKernelGGL(rope_vision_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (__bf16*)x, ld, M,
                     n_heads, D, (const float*)cos, (const float*)sin);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

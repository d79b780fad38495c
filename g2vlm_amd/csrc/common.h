// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libg2vlm_hip.so.
// Wave = 64 lanes; bf16 MFMA fragments follow MI355X guide §3 (16x16x32 and 32x32x16 maps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define G2V_OK 0
#define G2V_ERR_ARG (-22)      // EINVAL
#define G2V_ERR_LAUNCH (-5)    // EIO: hipGetLastError() != success after a launch

#define G2V_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return G2V_ERR_LAUNCH;  \
  } while (0)

__device__ __forceinline__ float bf2f(__bf16 x) { return (float)x; }
__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }   // v_cvt_pk_bf16_f32: RNE, NaN-safe
// Round-trip through bf16 that the optimiser cannot see through.  hipcc (ROCm 7.2) folds
// (float)(__bf16)(a*b) + c into fma(a,b,c) -- the intermediate bf16 rounding the reference performs
// between eager ops silently disappears.  The conversion instruction inside an asm is opaque to it.
__device__ __forceinline__ float bfround(float x) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
  return __uint_as_float(r << 16);
}

__device__ __forceinline__ float bits2f_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bits2f_lo(uint32_t w) { return __uint_as_float(w << 16); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return *reinterpret_cast<uint32_t*>(&v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU / SiLU / sigmoid in fp32 (callers round to bf16 where the reference does)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU without the libm erff call (~3x fewer instructions in a GEMM epilogue): erfc(a), a = |x|/sqrt(2), from the
// Chebyshev fit t*exp(-a^2 + P(t)), t = 1/(1 + a/2) (fractional error < 1.2e-7 everywhere, Numerical Recipes 6.2),
// evaluated in the exp2 domain; erf = +-(1 - erfc) is then rounded to fp32 exactly where torch's erff result is, so
// 0.5*x*(1 + erf) reproduces the reference's cancellation for negative x (modeling/pi3/.../mlp.py, dinov2 MLP: nn.GELU()).
__device__ __forceinline__ float gelu_fast(float x) {
  const float a = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, a, 1.0f));
  constexpr float L2E = 1.4426950408889634f;
  float p = 0.17087277f * L2E;
  p = fmaf(p, t, -0.82215223f * L2E);
  p = fmaf(p, t, 1.48851587f * L2E);
  p = fmaf(p, t, -1.13520398f * L2E);
  p = fmaf(p, t, 0.27886807f * L2E);
  p = fmaf(p, t, -0.18628806f * L2E);
  p = fmaf(p, t, 0.09678418f * L2E);
  p = fmaf(p, t, 0.37409196f * L2E);
  p = fmaf(p, t, 1.00002368f * L2E);
  p = fmaf(p, t, -1.26551223f * L2E);
  p = fmaf(-a * L2E, a, p);
  const float e = t * __builtin_amdgcn_exp2f(p);           // erfc(a)
  const float er = copysignf(1.0f - e, x);                 // erf(x / sqrt 2)
  return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // callers round to bf16

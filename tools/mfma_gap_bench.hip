// How many vector instructions hide in the gap of a 32x32x16 bf16 MFMA with ONE wave per SIMD, for the operand forms
// flash_fwd64_kernel uses (asm MFMAs; accumulator in AGPRs or VGPRs; B operand from AGPRs).  tools/mfma_gap_bench.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: D/C in AGPR (a[128..]), A/B VGPR.  MODE 1: D/C VGPR, B from AGPR a[64:67].  MODE 2: builtin MFMA (compiler-managed).
// NF fillers per MFMA of kind FK: 0 v_fma_f32, 1 v_exp_f32, 2 fma feeding exp (dependent pair = 2 instructions), 3 ds_read_b128
template <int MODE, int NF, int FK>
__global__ __launch_bounds__(256, 1) void gap_kernel(float* out, unsigned long long* cyc, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + lane * 0.001f + i); b[i] = (__bf16)(seed - i * 0.5f); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = seed + i;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  if (MODE != 2) {
    asm volatile("" ::: "a64", "a65", "a66", "a67");
    for (int j = 0; j < 1; ++j) asm volatile("v_accvgpr_write_b32 a64, %0\n v_accvgpr_write_b32 a65, %0\n v_accvgpr_write_b32 a66, %0\n v_accvgpr_write_b32 a67, %0\n s_nop 7" :: "v"(seed));
  }
  bf16x8 ld = a;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) {
        if (j == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" :: "v"(a), "v"(b) : "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143");
        if (j == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" :: "v"(a), "v"(b) : "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159");
        if (j == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" :: "v"(a), "v"(b) : "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175");
        if (j == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" :: "v"(a), "v"(b) : "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191");
      } else if (MODE == 1) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[64:67], %0" : "+v"(acc[j]) : "v"(a));
      } else {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if (FK == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k & 7]) : "v"(seed));
        if (FK == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 7]));
        if (FK == 2) asm volatile("v_fma_f32 %0, %0, %1, %1\n v_exp_f32 %0, %0" : "+v"(f[k & 7]) : "v"(seed));
        if (FK == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lane * 16));
      }
    }
    if (FK == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (MODE == 0) { float t; asm volatile("s_nop 15\n s_nop 15\n v_accvgpr_read_b32 %0, a128" : "=v"(t)); s += t; }
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  for (int i = 0; i < 8; ++i) s += f[i];
  s += (float)ld[0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int FK>
static void run_nf(int nf, float* out, unsigned long long* cyc, int iters, hipStream_t s) {
#define CASE(N) case N: hipLaunchKernelGGL((gap_kernel<MODE, N, FK>), dim3(256), dim3(256), 0, s, out, cyc, iters, 1.0f); break;
  switch (nf) { CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) default: break; }
#undef CASE
}
extern "C" int mfma_gap_bench(int mode, int nf, int fk, void* out, void* cyc, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  float* o = (float*)out; unsigned long long* c = (unsigned long long*)cyc;
#define MF(M, F) if (mode == M && fk == F) run_nf<M, F>(nf, o, c, iters, s);
  MF(0, 0) MF(0, 1) MF(0, 2) MF(0, 3) MF(1, 0) MF(1, 1) MF(1, 2) MF(1, 3) MF(2, 0) MF(2, 1) MF(2, 2) MF(2, 3)
#undef MF
  return (int)hipGetLastError();
}

// Energy per FLOP by MFMA shape: the same 64x64 output tile per wave, the same FLOPs per launch, random bf16 operands in registers
// (four distinct fragments per operand, rotated, so consecutive MFMAs see different data), one wave per SIMD, every CU busy.
// SHAPE 0: v_mfma_f32_32x32x16_bf16 (2 x 2 accumulators of 16), SHAPE 1: v_mfma_f32_16x16x32_bf16 (4 x 4 accumulators of 4).
// tools/mfma_shape_bench.py times them alone and alternating with a GEMM launch (pair time = energy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// the accumulators are a0 .. a63, named in asm MFMAs: hipcc's own allocation of 16 four-register accumulators put 28
// v_accvgpr_mov and 14 s_nop into the 16-MFMA loop body
template <int R> __device__ __forceinline__ void mfma32(const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(R), "i"(R + 15));
}
template <int R> __device__ __forceinline__ void mfma16(const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(R), "i"(R + 3));
}
template <int R> __device__ __forceinline__ void azero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "i"(R)); }
template <int R> __device__ __forceinline__ float aread() { float v; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R)); return v; }
template <int... I> __device__ __forceinline__ void azero_all(std::integer_sequence<int, I...>) { (azero<I>(), ...); }

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void shape_kernel(const bf16x8* __restrict__ rnd, float* out, int iters) {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",
               "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39",
               "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59",
               "a60", "a61", "a62", "a63");
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = rnd[(size_t)tid * 8 + i]; b[i] = rnd[(size_t)tid * 8 + 4 + i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));      // loaded and waited for before the loop
  azero_all(std::make_integer_sequence<int, 64>{});
  asm volatile("s_nop 7");
  for (int it = 0; it < iters; ++it) {                       // K = 32 per iteration
    if constexpr (SHAPE == 0) {                              // two k-steps of 16, 2 x 2 accumulators of 16 registers
      mfma32<0>(a[0], b[0]); mfma32<16>(a[0], b[1]); mfma32<32>(a[1], b[0]); mfma32<48>(a[1], b[1]);
      mfma32<0>(a[2], b[2]); mfma32<16>(a[2], b[3]); mfma32<32>(a[3], b[2]); mfma32<48>(a[3], b[3]);
    } else {                                                 // one k-step of 32, 4 x 4 accumulators of 4 registers
      mfma16<0>(a[0], b[0]); mfma16<4>(a[0], b[1]); mfma16<8>(a[0], b[2]); mfma16<12>(a[0], b[3]);
      mfma16<16>(a[1], b[0]); mfma16<20>(a[1], b[1]); mfma16<24>(a[1], b[2]); mfma16<28>(a[1], b[3]);
      mfma16<32>(a[2], b[0]); mfma16<36>(a[2], b[1]); mfma16<40>(a[2], b[2]); mfma16<44>(a[2], b[3]);
      mfma16<48>(a[3], b[0]); mfma16<52>(a[3], b[1]); mfma16<56>(a[3], b[2]); mfma16<60>(a[3], b[3]);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  out[tid] = aread<0>() + aread<17>() + aread<34>() + aread<51>();
}

// FLOPs per launch: 256 blocks x 4 waves x iters x 2 x 64 x 64 x 32
extern "C" int mfma_shape_bench(int shape, const void* rnd, void* out, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (shape == 0) hipLaunchKernelGGL(shape_kernel<0>, dim3(256), dim3(256), 0, s, (const bf16x8*)rnd, (float*)out, iters);
  else hipLaunchKernelGGL(shape_kernel<1>, dim3(256), dim3(256), 0, s, (const bf16x8*)rnd, (float*)out, iters);
  return (int)hipGetLastError();
}

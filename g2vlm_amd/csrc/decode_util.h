// Lane-level helpers shared by the decode-step kernels (decode_layer.hip, decode_batch.hip).
#pragma once
#include "common.h"

// In-kernel stamps (diagnostic build only: -DG2V_STAMPS, tools/decode_stamps.py; the shipped library executes none).  Lane 0
// of every wave stores s_memtime at up to 8 points of the kernel plus s_memrealtime at its start and end.
#ifdef G2V_STAMPS
#define G2V_STAMP_ARG , unsigned long long* dbg
#define G2V_STAMP_PASS , g_stamp_buf
#define G2V_STAMP_PASS_DEV , dbg
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
#define G2V_STAMP_PASS_DEV
#define G2V_STAMP(i)
#define G2V_STAMP_RT(i)
#endif


namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;

__device__ __forceinline__ float dot2(uint32_t w, uint32_t x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf2_t*>(&w), *reinterpret_cast<bf2_t*>(&x), acc, false);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// sum over each 16-lane row, every lane gets it; same pairing and order as `for (o = 8; o; o >>= 1) x += shfl_xor(x, o)`
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_f<0x128>(x);                                     // row_ror:8 == lane ^ 8
  x += dpp_f<0x124>(x);                                     // row_ror:4: lane ^ 4 up to the ^8 the first step made equal
  x += dpp_f<0x122>(x);
  x += dpp_f<0x121>(x);
  return x;
}

__device__ __forceinline__ float readlane_f(float x, int lane) {   // the builtin is integer-typed: a bare float argument is CONVERTED
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
// 64-lane sum without LDS round trips: DPP inside the 16-lane rows, then the four row sums through scalar registers.
// (`wave_sum` in common.h goes through ds_bpermute: ~150 cycles per step for a wave that runs alone on its SIMD.)
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x = row16_sum(x);
  const float r0 = readlane_f(x, 0), r1 = readlane_f(x, 16), r2 = readlane_f(x, 32), r3 = readlane_f(x, 48);
  return (r0 + r1) + (r2 + r3);
}

}  // namespace

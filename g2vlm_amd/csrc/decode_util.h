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

// ---- many wave sums at once ("recursive halving") ---------------------------------------------------------------------------
// V values per lane (V a power of two, 2 <= V <= 32), wanted: the 64-lane sum of each.  One butterfly per value costs ~11
// instructions; here every step EXCHANGES half of the values with the partner lane and adds, so the number of live values halves
// each step: V - 1 exchange-adds + (6 - log2 V) plain butterfly steps in all.  Partners: lane ^ 32 and ^ 16 by
// v_permlane32_swap / v_permlane16_swap (one swap serves both directions), ^ 8 by DPP row_ror:8, then i <-> 7 - i inside each
// group of 8 (row_half_mirror; the two always differ in bit 2), ^ 2 and ^ 1 by quad_perm.  Afterwards lane l holds, in the
// returned value, the total of value number l >> (6 - log2 V).  fp32 summation order differs from wave_sum_dpp's.
template <int M>
__device__ __forceinline__ float lane_partner(float x) {   // the partner's x for the step with mask M (8, 4, 2, 1)
  if constexpr (M == 8) return dpp_f<0x128>(x);
  else if constexpr (M == 4) return dpp_f<0x141>(x);
  else if constexpr (M == 2) return dpp_f<0x4E>(x);
  else return dpp_f<0xB1>(x);
}
template <int M>
__device__ __forceinline__ float halve_pair(float lo, float hi, int lane) {   // lanes with bit M clear end with sum(lo), the others sum(hi)
  if constexpr (M == 32) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
  } else if constexpr (M == 16) {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
  } else {
    const bool up = lane & M;
    const float keep = up ? hi : lo, send = up ? lo : hi;
    return keep + lane_partner<M>(send);
  }
}
template <int M>
__device__ __forceinline__ float all_pair(float x) {        // x + partner's x
  if constexpr (M == 32) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
  } else if constexpr (M == 16) {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
  } else {
    return x + lane_partner<M>(x);
  }
}
template <int V, int M>
__device__ __forceinline__ float reduce_transpose_step(float (&v)[V], int lane) {
  if constexpr (V > 1) {
    float h[V / 2];
#pragma unroll
    for (int j = 0; j < V / 2; ++j) h[j] = halve_pair<M>(v[j], v[j + V / 2], lane);
    if constexpr (M > 1) return reduce_transpose_step<V / 2, M / 2>(h, lane);
    else return h[0];
  } else if constexpr (M >= 1) {
    float x = all_pair<M>(v[0]);
    if constexpr (M > 1) {
      float one[1] = {x};
      return reduce_transpose_step<1, M / 2>(one, lane);
    } else {
      return x;
    }
  }
}
template <int V>
__device__ __forceinline__ float reduce_transpose(float (&v)[V], int lane) {
  static_assert(V >= 2 && V <= 32 && (V & (V - 1)) == 0, "2, 4, 8, 16 or 32 values");
  return reduce_transpose_step<V, 32>(v, lane);
}
constexpr int g2v_pow2_ge(int n) { return n <= 2 ? 2 : (n <= 4 ? 4 : (n <= 8 ? 8 : (n <= 16 ? 16 : 32))); }
constexpr int g2v_log2(int n) { return n <= 1 ? 0 : 1 + g2v_log2(n / 2); }

}  // namespace

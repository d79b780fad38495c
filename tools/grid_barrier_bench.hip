// Micro-benchmark behind the decision on a persistent whole-step decode kernel: what does a grid-wide
// barrier between 256 co-resident workgroups cost on MI355X, with the memory-model fences a real phase boundary needs?
//
//   mode 0: arrive (agent-scope atomic add) + spin on an agent-scope load; release / acquire fences around it
//   mode 1: the same without the fences (lower bound: the bare atomic + poll round trips)
//   mode 2: mode 0 plus a phase-like exchange: every wave publishes 64 floats before the barrier and reads 16 KB of
//           what the other blocks published after it (what a GEMV phase does with the activation vector)
//   mode 3: two-level: one counter per group of 32 blocks (an XCD's share when blocks are dealt round-robin), the last
//           arriver of a group bumps the global counter, everyone polls the global one
//   mode 4: mode 3 without the fences
//   mode 5: mode 4 plus the exchange of mode 2 done with agent-scope (sc1) stores and loads instead of fences
//   mode 7: 8 group counters and no second level: an arrival is ONE fire-and-forget atomic on the block's group word, a waiter
//           polls all 8 words with one 8-lane load; sc1 exchange as in mode 5
//   mode 6: flags: every block stores its epoch into its own word (no read-modify-write), a waiter's 256 threads read
//           the 256 words in one coalesced load and vote; exchange as in mode 5
// Every spin is bounded (SPIN_LIMIT polls, then the block gives up and raises `err`): the grid always drains.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ bool spin_until(const unsigned* ctr, unsigned target, int* err) {
  for (int it = 0; it < SPIN_LIMIT; ++it) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  *err = 1;
  return false;
}

template <int MODE>
__global__ __launch_bounds__(512) void barrier_bench_kernel(unsigned* ctr, unsigned* grp, float* xch, int iters, long long* cycles, int* err) {
  __shared__ int giveup;
  const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) giveup = 0;
  __syncthreads();
  float sink = 0.f;
  long long t0 = 0;
  for (int it = 0; it < iters; ++it) {
    if (it == 1) t0 = __builtin_readcyclecounter();     // iteration 0 absorbs the launch skew
    if (MODE == 2) {
      xch[(size_t)(it & 1) * nb * 512 + b * 512 + tid] = (float)(it + tid);
    }
    if (MODE >= 5) {
      __hip_atomic_store(&xch[(size_t)(it & 1) * nb * 512 + b * 512 + tid], (float)(it + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);                                   // the stores have been acknowledged before the arrival is published
    }
    __syncthreads();
    if (MODE == 7) {
      if (tid == 0) __hip_atomic_fetch_add(&grp[(b & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid < 64) {
        bool ok = false;
        const unsigned target = (unsigned)(it + 1) * (nb / 8);
        for (int sp = 0; sp < SPIN_LIMIT; ++sp) {
          const unsigned v = tid < 8 ? __hip_atomic_load(&grp[tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
          if (__all((int)(v >= target))) { ok = true; break; }
        }
        if (!ok) { giveup = 1; *err = 1; }
      }
    } else if (MODE == 6) {
      if (tid == 0) __hip_atomic_store(&grp[b], (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid < 256) {
        bool ok = false;
        for (int sp = 0; sp < SPIN_LIMIT; ++sp) {
          const unsigned v = tid < nb ? __hip_atomic_load(&grp[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
          if (__all((int)(v >= (unsigned)(it + 1)))) { ok = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { giveup = 1; *err = 1; }
      }
    } else if (tid == 0) {
      if (MODE != 1 && MODE != 4 && MODE != 5) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // buffer_wbl2 sc1
      const unsigned target = (unsigned)(it + 1) * nb;
      if (MODE == 3 || MODE == 4 || MODE == 5) {
        const int g = b & 7;                                         // blocks are dealt to the XCDs round-robin
        const unsigned old = __hip_atomic_fetch_add(&grp[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (unsigned)(it + 1) * (nb / 8))
          __hip_atomic_fetch_add(ctr, (unsigned)(nb / 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!spin_until(ctr, target, err)) giveup = 1;
      if (MODE != 1 && MODE != 4 && MODE != 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1
    }
    __syncthreads();
    if (giveup) break;
    if (MODE >= 5) {
      const float* src = xch + (size_t)(it & 1) * nb * 512;
#pragma unroll
      for (int j = 0; j < 8; ++j) sink += __hip_atomic_load(&src[((b + 1 + 31 * j) % nb) * 512 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 2) {
      // 16 KB per block: 8 floats per thread from other blocks' rows
      const float* src = xch + (size_t)(it & 1) * nb * 512;
#pragma unroll
      for (int j = 0; j < 8; ++j) sink += src[((b + 1 + 31 * j) % nb) * 512 + tid];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cycles[b] = t1 - t0;
  if (sink == 123.456f) cycles[0] = 0;
}
}  // namespace

extern "C" int grid_barrier_bench(int mode, int nblocks, int iters, void* ctr, void* grp, void* xch, void* cycles, void* err, void* stream) {
  auto k = mode == 0 ? barrier_bench_kernel<0> : mode == 1 ? barrier_bench_kernel<1> : mode == 2 ? barrier_bench_kernel<2>
         : mode == 3 ? barrier_bench_kernel<3> : mode == 4 ? barrier_bench_kernel<4> : mode == 5 ? barrier_bench_kernel<5> : mode == 6 ? barrier_bench_kernel<6> : barrier_bench_kernel<7>;
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(512), 0, (hipStream_t)stream, (unsigned*)ctr, (unsigned*)grp, (float*)xch, iters,
                     (long long*)cycles, (int*)err);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

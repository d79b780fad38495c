// LDS-DMA (global_load_lds_dwordx4) as asm statements + the small helpers around them; shared by attn.hip and gemm_4w.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// LDS-DMA as asm statements.  hipcc models the builtin (__builtin_amdgcn_global_load_lds) as a store to LDS that any later
// ds_read may alias, and puts an s_waitcnt vmcnt(0) in front of the next LDS read - every piece's full memory latency inside
// the tile loop.  An asm statement is not counted: the landing of the pieces is tracked by the kernel's own counted vmcnt and
// the tile barrier (guide 5.7 item 1, 'No VGPR destination').  M0 (LDS base of the piece) is written in the same statement;
// the s_nop covers SALU-write -> M0 use and a freshly computed SGPR base.
__device__ __forceinline__ void dma16_saddr(const char* base_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");
}
// the same with a base that was formed long before (steady state of flash_fwd64_kernel: at the tile's entry, 64 MFMAs earlier):
// only the M0 write needs its one wait state
__device__ __forceinline__ void dma16_saddr_settled(const char* base_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");
}
__device__ __forceinline__ void dma16_vaddr(const char* lane_ptr, uint32_t lds_addr_uniform) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(lane_ptr), "s"(lds_addr_uniform) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {       // a pointer every lane holds the same value of -> an SGPR pair
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)p; }   // a flat pointer into LDS: low 32 bits = LDS byte address

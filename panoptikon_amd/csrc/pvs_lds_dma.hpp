// pvs_lds_dma.hpp — LDS-DMA and wait primitives shared by the streaming kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// LDS-DMA issued from inline asm: hipcc models the builtin form as an LDS store that may
// alias every later ds_read and drains vmcnt(0) in front of them (two full pipeline drains
// per tile in the first build of this kernel).  An asm statement is invisible to its waitcnt
// insertion, so the counted s_waitcnt vmcnt(N) below are the only waits.  M0 carries the
// wave-uniform LDS destination; each lane lands at M0 + lane*size.  Source address =
// SGPR base (uniform: tile + k-slab) + 32-bit VGPR offset (lane's row/chunk inside the slab).
// `nt`: every corpus byte is read once per launch by exactly one CU (streaming policy; default / sc0 / sc1 measured 1.80 ms
// against 1.77 for the 256-query pass).
__device__ static inline void dma16(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    // (M0 declared clobbered instead of saved and restored: 1.270 vs 1.282 ms at 128 queries, noise)
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
// N contiguous 1-KiB pieces behind one M0 write: the immediate offset moves the global source and the LDS destination alike
// (tools/probe/dma_offset_test.hip)
template <int N>
__device__ static inline void dma16_group(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    static_assert(N >= 1 && N <= 4, "13-bit signed immediate");
    uint32_t keep;
    if constexpr (N == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ static inline void dma4(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
__device__ static inline uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)p;
}
template <int N>
__device__ static inline void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ static inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


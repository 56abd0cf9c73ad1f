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

// Global loads and a returning atomic that the compiler's waitcnt insertion does not see (like the LDS-DMA), so that they can
// travel IN FRONT of a stage's 16 DMA instructions and be read behind a counted wait: a compiler-visible load there makes hipcc
// wait for vmcnt(0) — the whole prefetched stage — at the value's first use, and hipcc turns atomicAdd(p, 1) of one lane into a
// wave-aggregated atomic followed by s_waitcnt vmcnt(0) + v_readfirstlane on the spot (the dequeue's round trip, once per unit).
// Their destinations are ACCUMULATION registers a250..a255 named in the instructions: an asm output operand in an ordinary VGPR is
// "defined" for the compiler the moment the asm statement ends, and it is free to copy it (a loop-carried variable, a live-range
// split) before the data has arrived — the first build of this kernel read stale norms and unit numbers that way.  Nothing else
// in these kernels comes near a250 (they use at most 60 accumulation registers as spill space); the values are read back with
// v_accvgpr_read behind the wait, inside the same asm statement.
__device__ static inline void row_scalars_async(const float *norm2, const uint8_t *mask, const uint32_t *trank) {
    if (norm2) asm volatile("global_load_dword a250, %0, off" ::"v"(norm2) : "memory", "a250");
    if (mask) asm volatile("global_load_ubyte a251, %0, off" ::"v"(mask) : "memory", "a251");
    if (trank) asm volatile("global_load_dword a252, %0, off" ::"v"(trank) : "memory", "a252");
}
// loads return in order: with N younger memory instructions allowed in flight the older ones have landed
template <int N>
__device__ static inline void row_scalars_wait(uint32_t &aa, uint32_t &mk, uint32_t &rk) {
    asm volatile("s_waitcnt vmcnt(%3)\n\tv_accvgpr_read_b32 %0, a250\n\tv_accvgpr_read_b32 %1, a251\n\tv_accvgpr_read_b32 %2, a252"
                 : "=v"(aa), "=v"(mk), "=v"(rk)
                 : "n"(N)
                 : "memory");
}
__device__ static inline void dequeue_async(uint32_t *counter) {  // ONE lane executes this
    const uint32_t one = 1;
    asm volatile("v_accvgpr_write_b32 a254, %1\n\tglobal_atomic_add a255, %0, a254, off sc0" ::"v"(counter), "v"(one) : "memory", "a254", "a255");
}
// everything this wave has in flight has landed; returns what the last dequeue_async fetched (lane 0's)
__device__ static inline uint32_t wait_all_and_dequeued() {
    uint32_t v;
    asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a255" : "=v"(v) : : "memory");
    return v;
}

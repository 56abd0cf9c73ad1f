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
#ifdef PVS_DMA_M0_CLOBBER  // tuning: tell the compiler M0 is gone instead of saving and restoring it around every piece
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#else
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
#endif
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


// ---- elastic hand-off between the waves of a workgroup (instead of s_barrier): monotonic arrival counters in LDS.
// `lds_signal` adds one arrival from this wave (call it from every lane: only lane 0 issues the atomic);
// `lds_await2` polls until both counters have reached their targets, with a bound on the number of polls so that a
// protocol bug can never hang the device: it returns false when it gave up.
__device__ static inline void lds_signal(uint32_t lds_counter, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(lds_counter), "v"(1u) : "memory");
}
__device__ static inline bool lds_await2(uint32_t lds_a, uint32_t need_a, uint32_t lds_b, uint32_t need_b) {
    for (int it = 0; it < (1 << 16); it++) {  // ~3 ms at the very least; a legitimate wait is a few microseconds
        uint32_t va, vb;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(va), "=&v"(vb) : "v"(lds_a), "v"(lds_b) : "memory");
        const uint32_t sa = __builtin_amdgcn_readfirstlane(va), sb = __builtin_amdgcn_readfirstlane(vb);
        if ((int)(sa - need_a) >= 0 && (int)(sb - need_b) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

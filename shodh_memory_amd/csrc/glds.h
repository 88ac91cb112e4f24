// glds.h -- LDS-DMA (global_load_lds_dwordx4) helpers shared by the streaming MFMA kernels (scan_mfma.hip, encoder.hip)
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace shodh {

// LDS-DMA: 16 bytes per lane from each lane's own global address (uniform base + per-lane byte offset) to
// LDS [m0 + lane*16]. hipcc neither counts nor waits for it (inline asm): completion is tracked by hand with
// s_waitcnt vmcnt(N) below. M0 is compiler-reserved, so it is saved and restored inside the statement.
__device__ __forceinline__ const unsigned char *uniform_ptr(const unsigned char *p) {   // provably wave-uniform for the "s" constraint
    const uint64_t v = (uint64_t)p;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);   // (the builtin returns int: no sign extension)
    return (const unsigned char *)(((uint64_t)hi << 32) | (uint64_t)lo);
}
__device__ __forceinline__ void glds16(const void *gbase, uint32_t voff_bytes, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

// the same, device-coherent (sc1: the line is fetched from beyond this XCD's L2): for words other workgroups update while the kernel runs
__device__ __forceinline__ void glds16_coherent(const void *gbase, uint32_t voff_bytes, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
// ... and 4 bytes per lane to LDS [m0 + lane*4]
__device__ __forceinline__ void glds4_coherent(const void *gbase, uint32_t voff_bytes, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

}  // namespace shodh

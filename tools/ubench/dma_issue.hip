// What an LDS-DMA instruction (global_load_lds_dwordx4, 1 KB per wave) costs the wave that issues it, by the number of waves of the workgroup that issue
// at the same time, with and without the matrix pipe / LDS reads of the emit scan beside it. One 512-thread workgroup per CU streams 48-KB tiles of a
// 768-MB buffer into a three-tile LDS ring, two tiles in flight (the shape of mfma_scan_kernel at 384 dimensions).
// hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue && ./dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int TILE = 48 * 1024, NB = 3;

__device__ __forceinline__ void glds16(const void *gbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}

// W issuing waves; MATH: every wave runs 48 MFMAs per tile, one ds_read_b128 each (the emit scan's chains); ROT: one wave issues the whole tile, in rotation
template <int W, bool MATH, bool ROT, bool CONTIG = false>
__global__ __launch_bounds__(512) void k(const unsigned char *rows, uint32_t n_tiles, unsigned long long *stall, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PER = 48 / W;
    uint32_t sel = blockIdx.x; const uint32_t step = gridDim.x;
    long long st = 0, cnt = 0;
    auto issue_tile = [&](uint32_t t, uint32_t b, int i) {      // piece i of this wave's share
        const unsigned char *src = rows + (size_t)t * TILE;
        const uint64_t v = (uint64_t)src;
        const unsigned char *us = (const unsigned char *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
        const int n = ROT ? i : (CONTIG ? wave * PER + i : wave + W * i);
        const long long t0 = clock64();
        glds16(us, (uint32_t)(n * 1024 + lane * 16), (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_lds + b * TILE + n * 1024)));
        st += clock64() - t0; cnt += 1;
    };
    constexpr int PERI = ROT ? 48 : PER;
    // prologue: two tiles
    for (int b = 0; b < 2; ++b) {
        const uint32_t t = sel + b * step < n_tiles ? sel + b * step : sel;
        const bool mine = ROT ? wave == b : wave < W;
        if (mine) for (int i = 0; i < PERI; ++i) issue_tile(t, b, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    half8 bq[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) bq[j][e] = (_Float16)(lane + j + e);
    floatx16 acc = {0};
    uint32_t cur = 0, tno = 0;
    for (; sel < n_tiles; sel += step, ++tno) {
        const uint32_t pfb = (cur + 2) % NB;
        const uint32_t psel = sel + 2 * step < n_tiles ? sel + 2 * step : sel;
        const bool mine = ROT ? (uint32_t)wave == ((tno + 2) & 7u) : wave < W;
        const unsigned char *buf = smem + cur * TILE;
        if (MATH) {
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                const half8 a = *reinterpret_cast<const half8 *>(buf + (lane & 31) * 768 + ((s % 24) * 2 + (lane >> 5)) * 16 + (s / 24) * 32 * 768);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[s & 3], acc, 0, 0, 0);
                if (mine && (ROT || (s % (48 / PER) == 0 && s / (48 / PER) < PER))) issue_tile(psel, pfb, ROT ? s : s / (48 / PER));
            }
        } else {
            if (mine) for (int i = 0; i < PERI; ++i) issue_tile(psel, pfb, i);
        }
        // the tile after this one has landed: at most this tile's pieces are still in flight
        if (ROT) {
            if ((uint32_t)wave == ((tno + 1) & 7u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (issued a tile ago)
        } else {
            if (wave < W) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER) : "memory");
        }
        __syncthreads();
        cur = cur + 1 == NB ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && cnt) { atomicAdd(stall, (unsigned long long)st); atomicAdd(stall + 1, (unsigned long long)cnt); }
    if (MATH && acc[0] == 12345.678f) sink[0] = acc[1];
}

template <int W, bool MATH, bool ROT, bool CONTIG = false>
void run(const unsigned char *rows, uint32_t n_tiles, unsigned long long *stall, float *sink, const char *name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k<W, MATH, ROT, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, NB * TILE));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<W, MATH, ROT, CONTIG>), dim3(256), dim3(512), NB * TILE, 0, rows, n_tiles, stall, sink);
    CK(hipDeviceSynchronize());
    CK(hipMemset(stall, 0, 16));
    const int N = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k<W, MATH, ROT, CONTIG>), dim3(256), dim3(512), NB * TILE, 0, rows, n_tiles, stall, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, stall, 16, hipMemcpyDeviceToHost));
    printf("%-34s %7.1f us per pass, %5.2f TB/s, %6.1f cycles per LDS-DMA instruction at issue\n", name, ms * 1000.0f / N, (double)n_tiles * TILE / (ms * 1e-3 / N) / 1e12, (double)h[0] / (double)h[1]);
}

int main() {
    const uint32_t n_tiles = 15625;
    unsigned char *rows; CK(hipMalloc(&rows, (size_t)n_tiles * TILE)); CK(hipMemset(rows, 0x11, (size_t)n_tiles * TILE));
    unsigned long long *stall; CK(hipMalloc(&stall, 16)); float *sink; CK(hipMalloc(&sink, 64));
    run<1, false, false>(rows, n_tiles, stall, sink, "stream alone, 1 issuing wave");
    run<2, false, false>(rows, n_tiles, stall, sink, "stream alone, 2 issuing waves");
    run<4, false, false>(rows, n_tiles, stall, sink, "stream alone, 4 issuing waves");
    run<8, false, false>(rows, n_tiles, stall, sink, "stream alone, 8 issuing waves");
    run<2, false, false, true>(rows, n_tiles, stall, sink, "stream alone, 2 waves, contiguous halves");
    run<4, false, false, true>(rows, n_tiles, stall, sink, "stream alone, 4 waves, contiguous quarters");
    run<8, false, false, true>(rows, n_tiles, stall, sink, "stream alone, 8 waves, contiguous eighths");
    return 0;
    run<1, true, false>(rows, n_tiles, stall, sink, "with the chains, 1 issuing wave");
    run<2, true, false>(rows, n_tiles, stall, sink, "with the chains, 2 issuing waves");
    run<4, true, false>(rows, n_tiles, stall, sink, "with the chains, 4 issuing waves");
    run<8, true, false>(rows, n_tiles, stall, sink, "with the chains, 8 issuing waves");
    run<1, true, true>(rows, n_tiles, stall, sink, "with the chains, one wave per tile in rotation");
    return 0;
}

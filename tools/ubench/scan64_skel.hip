// Skeleton of a one-wave-per-SIMD emit scan at 384 dimensions: 256-thread workgroups, wave w keeps 64 queries (two 32-query B-fragment sets, 192 registers) and
// every A fragment read from LDS feeds TWO MFMAs; the corpus streams through a three-tile LDS ring by LDS-DMA as in mfma_scan_kernel. No survivors, no thresholds
// that are ever reached: what the structure costs per tile, against the 8-wave kernel's ~5 300 cycles.
// hipcc --offload-arch=gfx950 -O3 scan64_skel.hip -o scan64_skel && ./scan64_skel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int KSTEPS = 24, PITCH = 768, TILE = 64 * PITCH, NBUF = 3, NPC = 12, D = 6, RING = 8, NS = 48;

__device__ __forceinline__ void glds16(const void *gbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ const unsigned char *uniform_ptr(const unsigned char *p) {
    const uint64_t v = (uint64_t)p;
    return (const unsigned char *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}

// SYNC 0: s_barrier per tile (first fragments of a tile requested after it); 1: progress words, first fragments of tile t + 1 requested during the last steps of tile t
template <int SYNC, int DMA_EVERY, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void skel(const unsigned char *rows, uint32_t n_tiles, const half8 *qfrag, float thr, float *sink, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    volatile uint32_t *sync_l = reinterpret_cast<volatile uint32_t *>(smem + NBUF * TILE);      // [0..3] landed, [4..7] finished
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l31 = lane & 31;
    uint32_t srcoff[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int p = i * 256 + tid, row = p / 48, slot = p % 48, c = (slot & ~15) | ((slot & 15) ^ (row & 15));
        srcoff[i] = (uint32_t)(row * PITCH + c * 16);
    }
    const uint32_t wave_lds = smem_lds + (uint32_t)wave * 1024u;
    uint32_t sel = blockIdx.x; const uint32_t step = gridDim.x;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const uint32_t t = sel + b * step < n_tiles ? sel + b * step : sel;
        const unsigned char *src = uniform_ptr(rows + (size_t)t * TILE);
#pragma unroll
        for (int i = 0; i < NPC; ++i) glds16(src, srcoff[i], wave_lds + b * TILE + i * 4096);
    }
    half8 bq0[KSTEPS], bq1[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) { bq0[ks] = qfrag[((wave * 2 + 0) * KSTEPS + ks) * 64 + lane]; bq1[ks] = qfrag[((wave * 2 + 1) * KSTEPS + ks) * 64 + lane]; }
    const int sw = l31 & 15;
    int aoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aoff[j] = l31 * PITCH + (((2 * j + hi) ^ sw) << 4);
    if (tid < 8) sync_l[tid] = tid < 4 ? 2u : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    floatx16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0};
    const floatx16 zero16 = {0};
    half8 ring[RING];
    auto rd = [&](const unsigned char *b_, int st) {
        const int rb = st / KSTEPS, ks = st % KSTEPS;
        ring[st % RING] = *reinterpret_cast<const half8 *>(b_ + rb * 32 * PITCH + aoff[ks & 7] + (ks >> 3) * 256);
    };
    auto poll = [&](int base, uint32_t need) {
        for (;;) { const uint32_t v = sync_l[base + (lane & 3)]; if (__builtin_amdgcn_ballot_w64(v >= need) == ~0ull) break; __builtin_amdgcn_s_sleep(1); }
    };
    uint32_t cur = 0, tno = 0;
    float hit = 0.0f;
    if (SYNC == 1) {
#pragma unroll
        for (int st = 0; st < D; ++st) rd(smem, st);
    }
    const long long t0 = clock64();
    for (; sel < n_tiles; sel += step, ++tno) {
        const unsigned char *buf = smem + cur * TILE;
        const uint32_t nxt = cur + 1 == NBUF ? 0 : cur + 1;
        const unsigned char *nbuf = smem + nxt * TILE;
        const uint32_t pfb = cur + 2 >= NBUF ? cur + 2 - NBUF : cur + 2;
        const uint32_t psel = sel + 2 * step < n_tiles ? sel + 2 * step : sel;
        const unsigned char *psrc = uniform_ptr(rows + (size_t)psel * TILE);
        const uint32_t pdst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + pfb * TILE));
        if (SYNC == 0) {
#pragma unroll
            for (int st = 0; st < D; ++st) rd(buf, st);
        }
        float m0 = -1e30f, m1 = -1e30f, n0 = -1e30f, n1 = -1e30f;
        uint32_t ff = 0;
#pragma unroll
        for (int st = 0; st < KSTEPS; ++st) {
            rd(buf, st + D);
            a00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq0[st], st == 0 ? zero16 : a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq1[st], st == 0 ? zero16 : a01, 0, 0, 0);
            if (!(ABL & 2) && st < 16) { m1 = fmaxf(m1, a10[st]); n1 = fmaxf(n1, a11[st]); }
            if (SYNC == 1 && st == 12) ff = sync_l[4 + (lane & 3)];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__builtin_amdgcn_ballot_w64(m1 >= thr || n1 >= thr) != 0) hit += m1 + n1;
        if (SYNC == 1) {
            // the buffer this tile's DMA refills: everybody through with the tile before this one; then what this wave has in flight is a tile old
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(ff >= tno) != ~0ull, 0)) poll(4, tno);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) sync_l[wave] = tno + 2u;
        }
        uint32_t fl = 0;
#pragma unroll
        for (int st = KSTEPS; st < NS; ++st) {
            if (SYNC == 1 && st == NS - D - 4) fl = sync_l[lane & 3];
            if (SYNC == 1 && st == NS - D - 1) { if (__builtin_expect(__builtin_amdgcn_ballot_w64(fl >= tno + 2u) != ~0ull, 0)) poll(0, tno + 2u); }
            if (st + D < NS) rd(buf, st + D);
            else if (SYNC == 1) rd(nbuf, st + D - NS);
            a10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq0[st - KSTEPS], st == KSTEPS ? zero16 : a10, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq1[st - KSTEPS], st == KSTEPS ? zero16 : a11, 0, 0, 0);
            if (!(ABL & 2) && st - KSTEPS >= 2 && st - KSTEPS < 18) { m0 = fmaxf(m0, a00[st - KSTEPS - 2]); n0 = fmaxf(n0, a01[st - KSTEPS - 2]); }
            if (!(ABL & 1) && ((st - KSTEPS) % DMA_EVERY) == 0 && (st - KSTEPS) / DMA_EVERY < NPC) { const int pi = (st - KSTEPS) / DMA_EVERY; glds16(psrc, srcoff[pi], pdst + pi * 4096); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__builtin_amdgcn_ballot_w64(m0 >= thr || n0 >= thr) != 0) hit += m0 + n0;
        if (SYNC == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPC) : "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F | (D << 8));      // lgkmcnt(D): LDS operations return in order -- everything but the D fragments of the next tile is in
            if (lane == 0) sync_l[4 + wave] = tno + 1u;
        }
        cur = nxt;
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (hit != 0.0f || a10[3] == 1234.5f) sink[blockIdx.x * 256 + tid] = hit + a10[0] + a11[1] + a00[2] + a01[3] + (float)ring[0][0];
    if (tid == 0 && blockIdx.x == 100) { cyc[0] = (unsigned long long)(t1 - t0); cyc[1] = tno; }
}

template <int SYNC, int DMA_EVERY, int ABL>
void run(const unsigned char *rows, uint32_t n_tiles, const half8 *q, float *sink, unsigned long long *cyc, const char *name) {
    const size_t lds = NBUF * TILE + 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)skel<SYNC, DMA_EVERY, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((skel<SYNC, DMA_EVERY, ABL>), dim3(256), dim3(256), lds, 0, rows, n_tiles, q, 3.0e38f, sink, cyc);
    CK(hipDeviceSynchronize());
    const int N = 50;
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((skel<SYNC, DMA_EVERY, ABL>), dim3(256), dim3(256), lds, 0, rows, n_tiles, q, 3.0e38f, sink, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
    printf("%-56s %7.1f us per pass, %6.0f cycles per tile (matrix pipe 3072)\n", name, ms * 1000.0f / N, (double)h[0] / (double)h[1]);
}

int main() {
    const uint32_t n_tiles = 15625;
    unsigned char *rows; CK(hipMalloc(&rows, (size_t)n_tiles * TILE));
    {   // fp16 noise of unit-vector size (x 256): the clocks follow the data
        const size_t n = (size_t)n_tiles * TILE / 2; _Float16 *h = (_Float16 *)malloc(n * 2); uint32_t s = 12345;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.013f); }
        CK(hipMemcpy(rows, h, n * 2, hipMemcpyHostToDevice)); free(h);
    }
    half8 *q; CK(hipMalloc(&q, 8 * KSTEPS * 64 * 16));
    { const size_t n = 8 * KSTEPS * 64 * 8; _Float16 *h = (_Float16 *)malloc(n * 2); uint32_t s = 777; for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.013f); } CK(hipMemcpy(q, h, n * 2, hipMemcpyHostToDevice)); free(h); }
    float *sink; CK(hipMalloc(&sink, 256 * 256 * 4)); unsigned long long *cyc; CK(hipMalloc(&cyc, 16));
    run<0, 2, 0>(rows, n_tiles, q, sink, cyc, "4 waves x 64 queries, s_barrier per tile, DMA every 2nd step");
    run<1, 2, 0>(rows, n_tiles, q, sink, cyc, "4 waves x 64 queries, progress words, DMA every 2nd step");
    run<1, 1, 0>(rows, n_tiles, q, sink, cyc, "4 waves x 64 queries, progress words, DMA every step");
    run<1, 2, 1>(rows, n_tiles, q, sink, cyc, "... progress words, NO DMA in the loop (stale LDS)");
    run<1, 2, 2>(rows, n_tiles, q, sink, cyc, "... progress words, no maximum folding");
    run<1, 2, 3>(rows, n_tiles, q, sink, cyc, "... progress words, neither (MFMA + ds_read stream alone)");
    return 0;
}

// The single-query scan as a stream: ONE wave of a 1024-thread workgroup issues the LDS-DMA of 32-row tiles (24 KB at 384-d fp16) into a five-tile ring, one
// s_barrier per tile, the sixteen waves read their rows out of LDS (16 lanes per row, 3 ds_read_b128 + 48 FMAs per lane) -- against solo_scan_kernel's direct
// global loads from all sixteen waves (129-131 us with the arithmetic off, 5.9 TB/s). hipcc --offload-arch=gfx950 -O3 solo_stream.hip -o solo_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void *gbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ const unsigned char *uniform_ptr(const unsigned char *p) {
    const uint64_t v = (uint64_t)p;
    return (const unsigned char *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
template <int TR, int NBUF, int NISS, bool ILV = false>      // rows per tile, tiles in the ring, issuing waves; ILV: workgroup b takes tiles b, b + grid, ... instead of a contiguous slice
__global__ __launch_bounds__(1024) void k(const unsigned char *rows, uint32_t n_rows, const float *q, uint32_t *keys) {
    constexpr int ROWB = 768, TILE = TR * ROWB, NP = TILE / 1024, PER = NP / NISS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), sr = lane >> 4, seg = lane & 15;
    const uint32_t slice = (n_rows + gridDim.x - 1) / gridDim.x, row0 = blockIdx.x * slice;
    const uint32_t n_loc = row0 >= n_rows ? 0 : (n_rows - row0 < slice ? n_rows - row0 : slice), n_tiles = (n_loc + TR - 1) / TR;
    float qv[24];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[c * 8 + e] = q[(c * 16 + seg) * 8 + e];
    const unsigned char *base = rows + (size_t)row0 * ROWB;
    auto issue = [&](uint32_t t) {      // tile t of the slice into buffer t % NBUF (this wave's share)
        const unsigned char *src = uniform_ptr(ILV ? rows + ((size_t)t * gridDim.x + blockIdx.x) * TILE : base + (size_t)t * TILE);
        const uint32_t dst = smem_lds + (t % NBUF) * TILE;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int n = (int)wave * PER + i; glds16(src, (uint32_t)(n * 1024 + lane * 16), (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + n * 1024))); }
    };
    if (wave < NISS) for (uint32_t t = 0; t < NBUF - 1 && t < n_tiles; ++t) issue(t);
    uint32_t best = 0xFFFFFFFFu;
    constexpr int WPT = TR / 4;      // waves that compute per tile
    for (uint32_t t = 0; t < n_tiles; ++t) {
        if (wave < NISS) {      // tile t has landed: at most the NBUF - 2 tiles after it are in flight
            const uint32_t after = n_tiles - 1 - t < (uint32_t)(NBUF - 2) ? n_tiles - 1 - t : (uint32_t)(NBUF - 2);
            if (after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * PER > 63 ? 63 : 3 * PER) : "memory");
            else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PER > 63 ? 63 : 2 * PER) : "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(1 * PER > 63 ? 63 : 1 * PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (wave < NISS && t + NBUF - 1 < n_tiles) issue(t + NBUF - 1);      // into the buffer of tile t - 1: everybody is through with it
        if ((wave / WPT) == (t % (16 / WPT))) {
            const uint32_t rl = (wave % WPT) * 4 + sr;
            const unsigned char *rp = smem + (t % NBUF) * TILE + rl * ROWB + seg * 16;
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const u32x4 h = *reinterpret_cast<const u32x4 *>(rp + c * 256);
                const _Float16 *hv = reinterpret_cast<const _Float16 *>(&h);
#pragma unroll
                for (int e = 0; e < 8; e += 2) { p0 = __builtin_fmaf(qv[c * 8 + e], (float)hv[e], p0); p1 = __builtin_fmaf(qv[c * 8 + e + 1], (float)hv[e + 1], p1); }
            }
            float pr = p0 + p1;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) pr += __shfl_xor(pr, off);
            const uint32_t key = __float_as_uint(pr);
            best = key < best ? key : best;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) keys[blockIdx.x * 16 + wave] = best;
}
template <int TR, int NBUF, int NISS, bool ILV = false>
void run(const unsigned char *rows, uint32_t n_rows, const float *q, uint32_t *keys, const char *name) {
    const size_t lds = (size_t)NBUF * TR * 768;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k<TR, NBUF, NISS, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<TR, NBUF, NISS, ILV>), dim3(256), dim3(1024), lds, 0, rows, n_rows, q, keys);
    CK(hipDeviceSynchronize());
    const int N = 50;
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k<TR, NBUF, NISS, ILV>), dim3(256), dim3(1024), lds, 0, rows, n_rows, q, keys);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %7.1f us per pass, %5.2f TB/s\n", name, ms * 1000.0f / N, (double)n_rows * 768 / (ms * 1e-3 / N) / 1e12);
}
int main() {
    const uint32_t n_rows = 1000000;
    unsigned char *rows; CK(hipMalloc(&rows, (size_t)(n_rows + 40000) * 768)); CK(hipMemset(rows, 0x31, (size_t)(n_rows + 40000) * 768));
    float *q; CK(hipMalloc(&q, 384 * 4)); CK(hipMemset(q, 0, 384 * 4)); uint32_t *keys; CK(hipMalloc(&keys, 256 * 16 * 4));
    run<32, 5, 1>(rows, n_rows, q, keys, "32-row tiles, ring of 5, one issuing wave");
    run<32, 5, 2>(rows, n_rows, q, keys, "32-row tiles, ring of 5, two issuing waves");
    run<64, 3, 1>(rows, n_rows, q, keys, "64-row tiles, ring of 3, one issuing wave");
    run<64, 3, 2>(rows, n_rows, q, keys, "64-row tiles, ring of 3, two issuing waves");
    run<32, 5, 1, true>(rows, n_rows, q, keys, "32-row tiles, ring of 5, one issuer, tiles interleaved");
    run<64, 3, 1, true>(rows, n_rows, q, keys, "64-row tiles, ring of 3, one issuer, tiles interleaved");
    run<64, 3, 2, true>(rows, n_rows, q, keys, "64-row tiles, ring of 3, two issuers, tiles interleaved");
    run<16, 8, 1>(rows, n_rows, q, keys, "16-row tiles, ring of 8, one issuing wave");
    return 0;
}

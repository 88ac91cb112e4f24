// LDS gather micro-benchmark: 8-bit PQ codes indexing a 48 KiB distance table, one query per 4-byte entry (ds_read_b32) against
// two / four queries interleaved per entry (ds_read_b64 / ds_read_b128): lookups per second over the whole chip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_gather.hip -o /tmp/lds_gather && /tmp/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int W>   // W queries per entry
__global__ __launch_bounds__(1024) void gather_kernel(const uint32_t *codes, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tab = reinterpret_cast<float *>(smem);
    for (int i = threadIdx.x; i < 12288; i += 1024) tab[i] = (float)(i & 1023) * 0.001f;
    __syncthreads();
    uint32_t cw[12];
    for (int i = 0; i < 12; ++i) cw[i] = codes[(blockIdx.x * 1024 + threadIdx.x) * 12 + i];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int MS = 48 / W;                      // sub-quantisers per 48-KiB tile
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            const uint32_t c = (cw[m >> 2] >> ((m & 3) * 8)) & 0xFFu;
            if (W == 1) acc[0] += tab[m * 256 + c];
            else if (W == 2) { const f32x2 v = *reinterpret_cast<const f32x2 *>(smem + (m * 256 + c) * 8); acc[0] += v.x; acc[1] += v.y; }
            else { const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + (m * 256 + c) * 16); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) cw[i] = cw[i] * 1664525u + 1013904223u;      // new codes every pass
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int W> void run(const uint32_t *codes, float *out, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *)gather_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    gather_kernel<W><<<blocks, 1024, 49152>>>(codes, out, 2);
    hipEventRecord(a);
    gather_kernel<W><<<blocks, 1024, 49152>>>(codes, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lookups = (double)blocks * 1024 * iters * 48;      // (48 / W) gathers of W values
    printf("W=%d: %.3f ms, %.1f G lookups/s, %.1f G gather-lanes/s\n", W, ms, lookups / ms * 1e-6, lookups / W / ms * 1e-6);
}
int main() {
    const int blocks = 2048, iters = 400;
    uint32_t *codes; float *out;
    hipMalloc(&codes, (size_t)blocks * 1024 * 12 * 4); hipMalloc(&out, (size_t)blocks * 1024 * 4);
    uint32_t *h = (uint32_t *)malloc((size_t)blocks * 1024 * 12 * 4);
    uint32_t s = 12345; for (size_t i = 0; i < (size_t)blocks * 1024 * 12; ++i) { s = s * 1103515245u + 12345u; h[i] = s ^ (s >> 13) ^ (s << 7); }
    hipMemcpy(codes, h, (size_t)blocks * 1024 * 12 * 4, hipMemcpyHostToDevice);
    run<1>(codes, out, blocks, iters); run<2>(codes, out, blocks, iters); run<4>(codes, out, blocks, iters);
    return 0;
}

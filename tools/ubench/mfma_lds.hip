// micro-benchmark: MFMA 32x32x16 f16 issue rate with / without concurrent LDS fragment reads.
// hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only, NACC independent accumulators
// MODE 1: MFMA + one ds_read_b128 per MFMA, prefetch distance D (register ring)
// MODE 2: MFMA + one ds_read_b128 per 2 MFMAs
template <int MODE, int NACC, int D>
__global__ __launch_bounds__(512) void k(const _Float16 *in, float *out, int iters, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 48 * 1024 / 16; i += blockDim.x) reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(in)[i];
    __syncthreads();
    half8 b = *reinterpret_cast<const half8 *>(in + lane * 8);
    floatx16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5, sw = l31 & 15;
    int aoff[8];
    for (int j = 0; j < 8; ++j) aoff[j] = l31 * 768 + (((2 * j + hi) ^ sw) << 4);
    half8 ring[D > 0 ? D : 1];
    long long t0 = clock64();
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 24; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, acc[u % NACC], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = *reinterpret_cast<const half8 *>(smem + aoff[d & 7] + (d >> 3) * 256);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                half8 a = ring[u % D];
                if (MODE == 1 || (u & 1) == 0) {
                    const int ks = (u + D) % 24;
                    ring[u % D] = *reinterpret_cast<const half8 *>(smem + ((it & 1) * 24576) + aoff[ks & 7] + (ks >> 3) * 256);
                }
                acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int NACC, int D>
void run(const char *name, int threads, const _Float16 *in, float *out, long long *cyc) {
    const int iters = 2000;
    hipFuncSetAttribute((const void *)k<MODE, NACC, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NACC, D><<<256, threads, 96 * 1024>>>(in, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NACC, D><<<256, threads, 96 * 1024>>>(in, out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_wave = 24.0 * iters;
    const int waves_per_simd = threads / 256;
    printf("%-44s threads=%d  %.1f clk64-ticks/MFMA/wave  wall %.3f ms  => %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", name, threads,
           (double)c / mfma_per_wave, ms, ms * 1e6 / (mfma_per_wave * waves_per_simd),
           mfma_per_wave * (threads / 64) * 256 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    _Float16 *in; float *out; long long *cyc;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipMemset(in, 0x3c, 1 << 20);
    for (int threads : {256, 512}) {
        run<0, 1, 0>("MFMA only, 1 accumulator (dependent)", threads, in, out, cyc);
        run<0, 2, 0>("MFMA only, 2 accumulators", threads, in, out, cyc);
        run<0, 4, 0>("MFMA only, 4 accumulators", threads, in, out, cyc);
        run<1, 1, 4>("MFMA + 1 ds_read_b128/MFMA, dist 4, 1 acc", threads, in, out, cyc);
        run<1, 1, 8>("MFMA + 1 ds_read_b128/MFMA, dist 8, 1 acc", threads, in, out, cyc);
        run<1, 2, 8>("MFMA + 1 ds_read_b128/MFMA, dist 8, 2 acc", threads, in, out, cyc);
        run<1, 4, 12>("MFMA + 1 ds_read_b128/MFMA, dist 12, 4 acc", threads, in, out, cyc);
        run<2, 4, 8>("MFMA + 1 ds_read_b128/2 MFMA, dist 8, 4 acc", threads, in, out, cyc);
    }
    return 0;
}

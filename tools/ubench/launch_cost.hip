// What a launch of G workgroups x T threads with L bytes of dynamic LDS costs when every workgroup does `work` microseconds of nothing (s_sleep) --
// the fixed cost around one-workgroup-per-query kernels (probe_select_kernel, adc_bound_kernel, lm_merge_kernel, final_stage_kernel at 1024 queries).
// hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k(unsigned *out, int spins) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);      // ~127 * 64 cycles
    if (threadIdx.x == 0 && smem[0] == 3) out[blockIdx.x] = 1;
}
int main() {
    unsigned *d; CK(hipMalloc(&d, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grids[] = {256, 1024, 4096}, threads[] = {64, 256, 512}, ldss[] = {0, 16 << 10, 38 << 10, 75 << 10}, spins[] = {0, 4};
    for (int sp : spins) for (int g : grids) for (int t : threads) for (int l : ldss) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(t), l, 0, d, sp);
        CK(hipDeviceSynchronize());
        const int N = 200;
        CK(hipEventRecord(e0));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(g), dim3(t), l, 0, d, sp);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("spins %d grid %5d x %3d threads, LDS %6d B: %.2f us per launch (back to back)\n", sp, g, t, l, ms * 1000.0f / N);
    }
    return 0;
}

// what does v_cvt_pk_u8_f32 do with fractions, negatives and values above 255? (round 3: the INT8 quantising epilogue wants f32 -> u8 in one instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        unsigned r = 0;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(r) : "v"(in[i]), "v"(0u));
        out[i] = r;
    }
}
int main() {
    const float h[] = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 255.4f, 255.5f, 300.0f, -0.4f, -0.6f, -5.0f, 127.49999f, 127.5f, 128.5f, 1e9f, -1e9f, 0.49999997f};
    const int n = sizeof(h) / 4;
    float *d; unsigned *o; unsigned ho[64];
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%.8g -> %u\n", h[i], ho[i] & 0xFF);
    return 0;
}

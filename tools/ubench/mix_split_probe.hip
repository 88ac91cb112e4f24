// mix_split_probe.hip -- the f16 (hi, lo) split of encoder_int8_fast.h's f16_split8 in its two forms, compared bit for bit on the device:
//   reference: hi = (f16)x, lo = (f16)(x - (float)hi)           (conversions and an f32 subtraction)
//   product:   v_cvt_pk_f16_f32, v_fma_mixlo_f16 / v_fma_mixhi_f16 (x * 1.0 - hi with hi read as f16, result written as f16)
// over 2^24 values: normal magnitudes of both signs, softmax weights down to the f16 subnormals, zeros, values past the f16 range.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/ubench/mix_split_probe.hip -o /tmp/mix_split_probe && /tmp/mix_split_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void probe(const float *x, uint32_t *ref, uint32_t *got, size_t n2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    uint16_t u[4]; memcpy(&u[0], &ha, 2); memcpy(&u[1], &hb, 2); memcpy(&u[2], &la, 2); memcpy(&u[3], &lb, 2);
    ref[2 * i] = u[0] | ((uint32_t)u[1] << 16); ref[2 * i + 1] = u[2] | ((uint32_t)u[3] << 16);
    uint32_t hi2, lo2;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi2) : "v"(a), "v"(b));
    asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo2) : "v"(a), "v"(hi2));
    asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo2) : "v"(b), "v"(hi2));
    got[2 * i] = hi2; got[2 * i + 1] = lo2;
}
int main() {
    const size_t n = 1u << 24;
    std::vector<float> x(n);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t r = (uint32_t)(st >> 32);
        const int kind = (int)(i & 7);
        float v;
        if (kind < 4) { uint32_t bits = (r & 0x807FFFFFu) | ((uint32_t)(127 - 20 + (int)((r >> 23) & 31)) << 23); memcpy(&v, &bits, 4); }      // |v| in [2^-20, 2^12)
        else if (kind < 6) { uint32_t bits = (r & 0x007FFFFFu) | ((uint32_t)(127 - 30 + (int)((r >> 23) % 31)) << 23); memcpy(&v, &bits, 4); }   // (0, 1]: softmax weights down to 2^-30
        else if (kind == 6) v = (r & 1) ? 0.0f : -0.0f;
        else { uint32_t bits = (r & 0x807FFFFFu) | ((uint32_t)(127 + 14 + (int)((r >> 23) & 3)) << 23); memcpy(&v, &bits, 4); }                  // around / past the f16 maximum
        x[i] = v;
    }
    float *dx; uint32_t *dr, *dg;
    hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dg, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<(unsigned)((n / 2 + 255) / 256), 256>>>(dx, dr, dg, n / 2);
    std::vector<uint32_t> r(n), g(n);
    hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(g.data(), dg, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) if (r[i] != g[i]) { if (bad < 8) printf("mismatch at %zu: x = %a %a ref %08x got %08x\n", i, x[(i & ~(size_t)1)], x[(i & ~(size_t)1) + 1], r[i], g[i]); ++bad; }
    printf("mix_split_probe: %zu values, %zu mismatching words\n", n, bad);
    return bad ? 1 : 0;
}

// common.h -- shared host/device helpers of libshodh_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/shodh_hip.h"
#include "guard.h"

namespace shodh {

void set_error(const char *fmt, ...);

#define SHODH_HIP_TRY(expr)                                                                    \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::shodh::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return SHODH_ERR_DEVICE;                                                           \
        }                                                                                      \
    } while (0)

#define SHODH_TRY(expr)              \
    do {                             \
        int s__ = (expr);            \
        if (s__ != SHODH_OK) return s__; \
    } while (0)

constexpr uint64_t KEY_NONE = 0xFFFFFFFFFFFFFFFFull;

// f32::total_cmp as an unsigned ascending key (vamana.rs:1185 sorts dist with total_cmp)
__host__ __device__ __forceinline__ uint32_t order_key(float x) {
    uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = __float_as_uint(x);
#else
    memcpy(&b, &x, 4);
#endif
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float order_key_inv(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}
// (dist, id) -> one ascending 64-bit key: (dist total_cmp asc, id asc) == vamana.rs:1185
__host__ __device__ __forceinline__ uint64_t make_key(float dist, uint32_t id) {
    return ((uint64_t)order_key(dist) << 32) | (uint64_t)id;
}

static inline uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
static inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// raises a kernel's dynamic-LDS limit once (hipFuncSetAttribute costs host time on every call)
int ensure_dynamic_lds(const void *fn, size_t bytes);

}  // namespace shodh

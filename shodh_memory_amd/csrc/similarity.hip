// similarity.hip -- src/similarity.rs:10-48: cosine_similarity for many pairs and top_k_similar, on the device, in the
// reference's accumulation order (dot_product_inline, distance_inline.rs:67-173). The reference calls these for tens of
// pairs per request (pairwise checks), so one pair per thread is all the parallelism that exists; the sort of
// top_k_similar is the host's (a stable sort by OrderedFloat, descending).
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)

namespace shodh {

// ---- pairwise cosine (similarity.rs:10-24), one pair per thread ------------------------------------------------
template <int ORDER>
__device__ __forceinline__ float ref_dot(const float *a, const float *b, uint32_t n) {
    if (ORDER == SHODH_ORDER_SCALAR4) {
        const uint32_t un = n & ~3u;
        float sum = 0.0f;
        for (uint32_t i = 0; i < un; i += 4) {
            float t = a[i] * b[i];
            t = t + a[i + 1] * b[i + 1];
            t = t + a[i + 2] * b[i + 2];
            t = t + a[i + 3] * b[i + 3];
            sum = sum + t;
        }
        for (uint32_t j = un; j < n; ++j) sum = sum + a[j] * b[j];
        return sum;
    } else {
        const uint32_t sn = n & ~7u;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = 0; i < sn; i += 8)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[l] = __builtin_fmaf(a[i + l], b[i + l], acc[l]);
        float r = acc[0] + acc[1];
        r = r + acc[2]; r = r + acc[3]; r = r + acc[4]; r = r + acc[5]; r = r + acc[6]; r = r + acc[7];
        for (uint32_t j = sn; j < n; ++j) r = r + a[j] * b[j];
        return r;
    }
}
template <int ORDER>
__global__ void cosine_batch_kernel(const float *a, const float *b, uint64_t n, uint32_t dim, float *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *x = a + i * dim, *y = b + i * dim;
    const float dot = ref_dot<ORDER>(x, y, dim);
    const float na = __builtin_sqrtf(ref_dot<ORDER>(x, x, dim));
    const float nb = __builtin_sqrtf(ref_dot<ORDER>(y, y, dim));
    float r;
    if (na == 0.0f || nb == 0.0f) r = 0.0f;
    else {
        r = dot / (na * nb);
        if (r < -1.0f) r = -1.0f;
        if (r > 1.0f) r = 1.0f;
    }
    out[i] = r;
}


// one query against n candidates (top_k_similar's map, similarity.rs:31-37)
template <int ORDER>
__global__ void cosine_one_to_many_kernel(const float *q, const float *cands, uint64_t n, uint32_t dim, float *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *y = cands + i * dim;
    const float dot = ref_dot<ORDER>(q, y, dim);
    const float na = __builtin_sqrtf(ref_dot<ORDER>(q, q, dim));
    const float nb = __builtin_sqrtf(ref_dot<ORDER>(y, y, dim));
    float r;
    if (na == 0.0f || nb == 0.0f) r = 0.0f;
    else {
        r = dot / (na * nb);
        if (r < -1.0f) r = -1.0f;       // f32::clamp: NaN stays NaN
        if (r > 1.0f) r = 1.0f;
    }
    out[i] = r;
}

// OrderedFloat<f32>::cmp (ordered-float 5.x): NaN is equal to NaN and greater than everything else; -0.0 == +0.0
static inline int ordered_float_cmp(float a, float b) {
    const bool an = std::isnan(a), bn = std::isnan(b);
    if (an || bn) return an && bn ? 0 : (an ? 1 : -1);
    return a < b ? -1 : (a > b ? 1 : 0);
}

}  // namespace shodh

using namespace shodh;

extern "C" {

int shodh_cosine_similarity_batch(int device, const float *a, const float *b, uint64_t n, uint32_t dim, uint32_t order, float *out) {
    if (n && (!a || !b || !out)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (n == 0) return SHODH_OK;
    SHODH_HIP_TRY(hipSetDevice(device));
    float *d = nullptr;
    SHODH_HIP_TRY(dev_alloc((void **)&d, (2 * n * dim + n) * 4));
    hipError_t e = hipMemcpy(d, a, n * dim * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n * dim, b, n * dim * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (order == SHODH_ORDER_AVX2) hipLaunchKernelGGL((cosine_batch_kernel<SHODH_ORDER_AVX2>), dim3((uint32_t)ceil_div(n, 64)), dim3(64), 0, nullptr, d, d + n * dim, n, dim, d + 2 * n * dim);
        else hipLaunchKernelGGL((cosine_batch_kernel<SHODH_ORDER_SCALAR4>), dim3((uint32_t)ceil_div(n, 64)), dim3(64), 0, nullptr, d, d + n * dim, n, dim, d + 2 * n * dim);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n * dim, n * 4, hipMemcpyDeviceToHost);
    dev_free(d);
    if (e != hipSuccess) { set_error("cosine batch failed: %s", hipGetErrorString(e)); return SHODH_ERR_DEVICE; }
    return SHODH_OK;
}


int shodh_top_k_similar(int device, const float *query, uint32_t query_dim, const float *cands, uint64_t n, uint32_t dim,
                        uint64_t k, uint32_t order, float *out_scores, uint32_t *out_index, uint64_t *count_out) {
    if (!count_out || (n && (!cands || !query)) || (n && k && (!out_scores || !out_index))) { set_error("null argument"); return SHODH_ERR_INVALID; }
    *count_out = 0;
    if (n == 0) return SHODH_OK;                                   // empty candidates -> empty result (similarity.rs:117-122)
    if (n > 0xFFFFFFFFull) { set_error("too many candidates"); return SHODH_ERR_INVALID; }
    std::vector<float> score(n, 0.0f);
    if (query_dim == dim && dim != 0) {                            // a.len() != b.len() -> 0.0 for every candidate (similarity.rs:11-13)
        SHODH_HIP_TRY(hipSetDevice(device));
        float *d = nullptr;
        SHODH_HIP_TRY(dev_alloc((void **)&d, ((n + 1) * dim + n) * 4));
        hipError_t e = hipMemcpy(d, query, (size_t)dim * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d + dim, cands, n * dim * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            float *d_out = d + (n + 1) * dim;
            if (order == SHODH_ORDER_AVX2) hipLaunchKernelGGL((cosine_one_to_many_kernel<SHODH_ORDER_AVX2>), dim3((uint32_t)ceil_div(n, 64)), dim3(64), 0, nullptr, d, d + dim, n, dim, d_out);
            else hipLaunchKernelGGL((cosine_one_to_many_kernel<SHODH_ORDER_SCALAR4>), dim3((uint32_t)ceil_div(n, 64)), dim3(64), 0, nullptr, d, d + dim, n, dim, d_out);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpy(score.data(), d_out, n * 4, hipMemcpyDeviceToHost);
        }
        dev_free(d);
        if (e != hipSuccess) { set_error("top_k_similar failed: %s", hipGetErrorString(e)); return SHODH_ERR_DEVICE; }
    }
    std::vector<uint32_t> idx(n);
    for (uint64_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    // scored.sort_by(|a, b| b.0.cmp(&a.0)) -- slice::sort_by is stable: equal scores keep their input order (no id tie-break)
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ordered_float_cmp(score[b], score[a]) < 0; });
    const uint64_t m = k < n ? k : n;
    for (uint64_t i = 0; i < m; ++i) { out_scores[i] = score[idx[i]]; out_index[i] = idx[i]; }
    *count_out = m;
    return SHODH_OK;
}

}  // extern "C"

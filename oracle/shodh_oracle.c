/*
 * shodh_oracle.c -- CPU restatement of the shodh-memory embed-and-recall hot path.
 * TEST INFRASTRUCTURE ONLY (see shodh_oracle.h). Compile with -ffp-contract=off.
 * All file:line citations are into /root/reference/src/.
 */
#define _GNU_SOURCE
#include "shodh_oracle.h"

#include <math.h>
#include <float.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* distance_inline.rs                                                                    */
/* ------------------------------------------------------------------------------------ */

/* vector_db/distance_inline.rs:157-173  dot_product_scalar_inline
 *   sum += a[i]*b[i] + a[i+1]*b[i+1] + a[i+2]*b[i+2] + a[i+3]*b[i+3];  (left-assoc, no FMA) */
float so_dot_scalar4(const float *a, const float *b, size_t n) {
    size_t un = n & ~(size_t)3;
    float sum = 0.0f;
    for (size_t i = 0; i < un; i += 4) {
        float g = a[i] * b[i];
        g = g + a[i + 1] * b[i + 1];
        g = g + a[i + 2] * b[i + 2];
        g = g + a[i + 3] * b[i + 3];
        sum = sum + g;
    }
    for (size_t j = un; j < n; ++j) sum = sum + a[j] * b[j];
    return sum;
}

/* vector_db/distance_inline.rs:67-111  dot_product_avx2_inline, emulated lane by lane with
 * fmaf (single rounding == _mm256_fmadd_ps).  One 8-lane accumulator; the 2x unrolled loop
 * feeds the SAME accumulator, so it is equivalent to a plain 8-wide loop. */
float so_dot_avx2(const float *a, const float *b, size_t n) {
    size_t sn = n & ~(size_t)7;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < sn; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(a[i + l], b[i + l], acc[l]);
    float r = acc[0] + acc[1];
    r = r + acc[2]; r = r + acc[3]; r = r + acc[4]; r = r + acc[5]; r = r + acc[6]; r = r + acc[7];
    for (size_t j = sn; j < n; ++j) r = r + a[j] * b[j];
    return r;
}

#if defined(__x86_64__)
__attribute__((target("avx2,fma")))
static float dot_avx2_intrin(const float *a, const float *b, size_t n) {
    size_t sn = n & ~(size_t)7;
    __m256 sum = _mm256_setzero_ps();
    size_t i = 0;
    while (i + 16 <= sn) {
        __m256 va1 = _mm256_loadu_ps(a + i), vb1 = _mm256_loadu_ps(b + i);
        __m256 va2 = _mm256_loadu_ps(a + i + 8), vb2 = _mm256_loadu_ps(b + i + 8);
        sum = _mm256_fmadd_ps(va1, vb1, sum);
        sum = _mm256_fmadd_ps(va2, vb2, sum);
        i += 16;
    }
    while (i < sn) {
        sum = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), sum);
        i += 8;
    }
    float s[8];
    _mm256_storeu_ps(s, sum);
    volatile float r = s[0] + s[1];
    r = r + s[2]; r = r + s[3]; r = r + s[4]; r = r + s[5]; r = r + s[6]; r = r + s[7];
    float rr = r;
    for (size_t j = sn; j < n; ++j) rr = rr + a[j] * b[j];
    return rr;
}
#endif

float so_dot_avx2_native(const float *a, const float *b, size_t n) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return dot_avx2_intrin(a, b, n);
#endif
    return so_dot_avx2(a, b, n);
}

float so_dot(const float *a, const float *b, size_t n, int order) {
    return order == SO_ORDER_AVX2 ? so_dot_avx2_native(a, b, n) : so_dot_scalar4(a, b, n);
}

/* :283-305 euclidean_squared_scalar_inline */
float so_l2sq_scalar4(const float *a, const float *b, size_t n) {
    size_t un = n & ~(size_t)3;
    float sum = 0.0f;
    for (size_t i = 0; i < un; i += 4) {
        float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
        float g = d0 * d0;
        g = g + d1 * d1;
        g = g + d2 * d2;
        g = g + d3 * d3;
        sum = sum + g;
    }
    for (size_t j = un; j < n; ++j) { float d = a[j] - b[j]; sum = sum + d * d; }
    return sum;
}

/* :219-253 euclidean_squared_avx2_inline (NOT 2x unrolled) */
float so_l2sq_avx2(const float *a, const float *b, size_t n) {
    size_t sn = n & ~(size_t)7;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < sn; i += 8)
        for (int l = 0; l < 8; ++l) { float d = a[i + l] - b[i + l]; acc[l] = fmaf(d, d, acc[l]); }
    float r = acc[0] + acc[1];
    r = r + acc[2]; r = r + acc[3]; r = r + acc[4]; r = r + acc[5]; r = r + acc[6]; r = r + acc[7];
    for (size_t j = sn; j < n; ++j) { float d = a[j] - b[j]; r = r + d * d; }
    return r;
}

float so_l2sq(const float *a, const float *b, size_t n, int order) {
    return order == SO_ORDER_AVX2 ? so_l2sq_avx2(a, b, n) : so_l2sq_scalar4(a, b, n);
}

/* :413-431 l2_norm_squared_scalar_inline */
float so_normsq_scalar4(const float *a, size_t n) { return so_dot_scalar4(a, a, n); }

/* :355-385 l2_norm_squared_avx2_inline */
float so_normsq_avx2(const float *a, size_t n) {
    size_t sn = n & ~(size_t)7;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < sn; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(a[i + l], a[i + l], acc[l]);
    float r = acc[0] + acc[1];
    r = r + acc[2]; r = r + acc[3]; r = r + acc[4]; r = r + acc[5]; r = r + acc[6]; r = r + acc[7];
    for (size_t j = sn; j < n; ++j) r = r + a[j] * a[j];
    return r;
}

float so_normsq(const float *a, size_t n, int order) {
    return order == SO_ORDER_AVX2 ? so_normsq_avx2(a, n) : so_normsq_scalar4(a, n);
}

/* :315-317 */
float so_l2_norm(const float *a, size_t n, int order) { return sqrtf(so_normsq(a, n, order)); }

/* :444-458 cosine_similarity_inline: dot / sqrt(nsq(a)*nsq(b)); denom < 1e-10 -> 0; no clamp */
float so_cosine_similarity_inline(const float *a, const float *b, size_t n, int order) {
    float dot = so_dot(a, b, n, order);
    float na = so_normsq(a, n, order);
    float nb = so_normsq(b, n, order);
    float den = sqrtf(na * nb);
    if (den < 1e-10f) return 0.0f;
    return dot / den;
}

/* :464-466 */
float so_cosine_distance_inline(const float *a, const float *b, size_t n, int order) {
    return 1.0f - so_cosine_similarity_inline(a, b, n, order);
}

/* :479-481 */
float so_normalized_distance(const float *a, const float *b, size_t n, int order) {
    return -so_dot(a, b, n, order);
}

/* :487-490 */
int so_is_normalized(const float *a, size_t n, float eps, int order) {
    return fabsf(so_normsq(a, n, order) - 1.0f) < eps;
}

/* :494-502 (multiplies by the reciprocal) */
void so_normalize_inplace(float *a, size_t n, int order) {
    float norm = so_l2_norm(a, n, order);
    if (norm > 1e-10f) {
        float inv = 1.0f / norm;
        for (size_t i = 0; i < n; ++i) a[i] = a[i] * inv;
    }
}

/* vector_db/vamana.rs:755-761 VamanaIndex::distance */
float so_metric_distance(const float *a, const float *b, size_t n, int metric, int order) {
    switch (metric) {
        case SO_METRIC_EUCLIDEAN: return so_l2sq(a, b, n, order);
        case SO_METRIC_COSINE: return 1.0f - so_cosine_similarity_inline(a, b, n, order);
        default: return so_normalized_distance(a, b, n, order);
    }
}

/* ------------------------------------------------------------------------------------ */
/* f32::total_cmp                                                                        */
/* ------------------------------------------------------------------------------------ */

static inline int32_t total_key_i32(float x) {
    int32_t b;
    memcpy(&b, &x, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);  /* core::f32::total_cmp */
    return b;
}

int so_total_cmp(float a, float b) {
    int32_t x = total_key_i32(a), y = total_key_i32(b);
    return (x > y) - (x < y);
}

uint32_t so_total_order_key(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* ------------------------------------------------------------------------------------ */
/* similarity.rs                                                                         */
/* ------------------------------------------------------------------------------------ */

static inline float clampf(float x, float lo, float hi) {
    /* f32::clamp: NaN stays NaN */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* similarity.rs:10-24 */
float so_cosine_similarity(const float *a, size_t na, const float *b, size_t nb, int order) {
    if (na != nb) return 0.0f;
    float dot = so_dot(a, b, na, order);
    float norm_a = sqrtf(so_dot(a, a, na, order));
    float norm_b = sqrtf(so_dot(b, b, na, order));
    if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;
    return clampf(dot / (norm_a * norm_b), -1.0f, 1.0f);
}

/* OrderedFloat<f32>::cmp : NaN is greatest, NaN == NaN, -0 == +0 */
static int ordered_float_cmp(float a, float b) {
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an - bn;  /* (1,1)->0, (1,0)->1, (0,1)->-1 */
    return (a > b) - (a < b);
}

typedef struct { float s; uint32_t i; } scored_t;

static void merge_sort_scored_desc(scored_t *v, scored_t *tmp, size_t n) {
    /* stable merge sort, descending on OrderedFloat (slice::sort_by is stable) */
    if (n < 2) return;
    size_t h = n / 2;
    merge_sort_scored_desc(v, tmp, h);
    merge_sort_scored_desc(v + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* comparator: b.0.cmp(&a.0); take left unless right is strictly "less" under it */
        if (ordered_float_cmp(v[j].s, v[i].s) > 0) tmp[k++] = v[j++]; else tmp[k++] = v[i++];
    }
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, n * sizeof(scored_t));
}

/* similarity.rs:27-48 */
size_t so_top_k_similar(const float *query, const float *cands, size_t n, size_t dim, size_t k,
                        int order, float *out_scores, uint32_t *out_index) {
    if (n == 0) return 0;
    scored_t *v = (scored_t *)malloc(n * sizeof(scored_t));
    scored_t *tmp = (scored_t *)malloc(n * sizeof(scored_t));
    for (size_t i = 0; i < n; ++i) {
        v[i].s = so_cosine_similarity(query, dim, cands + i * dim, dim, order);
        v[i].i = (uint32_t)i;
    }
    merge_sort_scored_desc(v, tmp, n);
    size_t m = k < n ? k : n;
    for (size_t i = 0; i < m; ++i) { out_scores[i] = v[i].s; out_index[i] = v[i].i; }
    free(v); free(tmp);
    return m;
}

/* ------------------------------------------------------------------------------------ */
/* vamana.rs brute_force_search                                                          */
/* ------------------------------------------------------------------------------------ */

typedef struct { uint32_t id; float d; } idd_t;

static int idd_cmp(const void *pa, const void *pb) {
    const idd_t *a = (const idd_t *)pa, *b = (const idd_t *)pb;
    int c = so_total_cmp(a->d, b->d);           /* a.1.total_cmp(&b.1) */
    if (c) return c;
    return (a->id > b->id) - (a->id < b->id);   /* .then_with(|| a.0.cmp(&b.0)) */
}

/* vamana.rs:1167-1188 */
size_t so_brute_force_search(const float *rows, size_t n, size_t dim, const uint8_t *deleted,
                             const float *q, size_t k, int metric, int order,
                             uint32_t *out_ids, float *out_dist) {
    idd_t *v = (idd_t *)malloc((n ? n : 1) * sizeof(idd_t));
    float *clone = (float *)malloc((dim ? dim : 1) * sizeof(float));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (deleted && deleted[i]) continue;
        memcpy(clone, rows + i * dim, dim * sizeof(float));      /* get_vector() clones (:444-451) */
        v[m].id = (uint32_t)i;
        v[m].d = so_metric_distance(q, clone, dim, metric, order); /* distance(query, &vec) */
        ++m;
    }
    qsort(v, m, sizeof(idd_t), idd_cmp);   /* keys are unique (id), so stability is moot */
    size_t out = m < k ? m : k;
    for (size_t i = 0; i < out; ++i) { out_ids[i] = v[i].id; out_dist[i] = v[i].d; }
    free(v); free(clone);
    return out;
}

/* identical output; keeps the k best in a sorted buffer instead of sorting all n */
size_t so_brute_force_search_select(const float *rows, size_t n, size_t dim, const uint8_t *deleted,
                                    const float *q, size_t k, int metric, int order,
                                    uint32_t *out_ids, float *out_dist) {
    if (k == 0) return 0;
    idd_t *best = (idd_t *)malloc(k * sizeof(idd_t));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (deleted && deleted[i]) continue;
        idd_t c;
        c.id = (uint32_t)i;
        c.d = so_metric_distance(q, rows + i * dim, dim, metric, order);
        if (m == k && idd_cmp(&c, &best[k - 1]) >= 0) continue;
        size_t p = (m < k) ? m : k - 1;
        while (p > 0 && idd_cmp(&c, &best[p - 1]) < 0) { best[p] = best[p - 1]; --p; }
        best[p] = c;
        if (m < k) ++m;
    }
    for (size_t i = 0; i < m; ++i) { out_ids[i] = best[i].id; out_dist[i] = best[i].d; }
    free(best);
    return m;
}

/* ------------------------------------------------------------------------------------ */
/* retrieval.rs search_ids post-processing                                               */
/* ------------------------------------------------------------------------------------ */

typedef struct { uint8_t u[16]; float s; } uuid_score_t;

static int uuid_score_cmp_desc(const void *pa, const void *pb) {
    const uuid_score_t *a = (const uuid_score_t *)pa, *b = (const uuid_score_t *)pb;
    int c = so_total_cmp(b->s, a->s);   /* b.1.total_cmp(&a.1) */
    if (c) return c;
    return memcmp(a->u, b->u, 16);      /* a.0.cmp(&b.0): Uuid orders by its 16 bytes */
}

/* memory/retrieval.rs:920-963 */
size_t so_search_ids_postprocess(const uint32_t *vec_ids, const float *dists, size_t n_res,
                                 const uint8_t *vector_to_memory, size_t n_vectors, size_t limit,
                                 uint8_t *out_uuid, float *out_sim) {
    static const uint8_t none[16] = {255, 255, 255, 255, 255, 255, 255, 255,
                                     255, 255, 255, 255, 255, 255, 255, 255};
    uuid_score_t *best = (uuid_score_t *)malloc((n_res ? n_res : 1) * sizeof(uuid_score_t));
    size_t nb = 0;
    for (size_t r = 0; r < n_res; ++r) {
        float similarity = -dists[r];
        if (vec_ids[r] >= n_vectors) continue;
        const uint8_t *u = vector_to_memory + (size_t)vec_ids[r] * 16;
        if (memcmp(u, none, 16) == 0) continue;
        size_t j = 0;
        for (; j < nb; ++j) if (memcmp(best[j].u, u, 16) == 0) break;
        if (j == nb) { memcpy(best[nb].u, u, 16); best[nb].s = similarity; ++nb; }
        else if (similarity > best[j].s) best[j].s = similarity;   /* strict '>' */
    }
    qsort(best, nb, sizeof(uuid_score_t), uuid_score_cmp_desc);
    size_t out = nb < limit ? nb : limit;
    for (size_t i = 0; i < out; ++i) { memcpy(out_uuid + i * 16, best[i].u, 16); out_sim[i] = best[i].s; }
    free(best);
    return out;
}

/* ------------------------------------------------------------------------------------ */
/* pq.rs                                                                                 */
/* ------------------------------------------------------------------------------------ */

/* vector_db/pq.rs:393-395: a.iter().zip(b).map(|(x,y)| (x-y).powi(2)).sum()  (sequential) */
float so_squared_l2(const float *a, const float *b, size_t n) {
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) { float d = a[i] - b[i]; s = s + d * d; }
    return s;
}

/* pq.rs:220-257 */
void so_pq_encode(const float *cb, size_t M, size_t ncent, size_t sub, const float *v, uint8_t *codes) {
    for (size_t m = 0; m < M; ++m) {
        uint8_t best = 0;
        float best_dist = FLT_MAX;
        for (size_t c = 0; c < ncent; ++c) {
            float d = so_squared_l2(v + m * sub, cb + (m * ncent + c) * sub, sub);
            if (d < best_dist) { best_dist = d; best = (uint8_t)c; }
        }
        codes[m] = best;
    }
}

/* pq.rs:259-291 */
void so_pq_decode(const float *cb, size_t M, size_t ncent, size_t sub, const uint8_t *codes, float *out) {
    for (size_t m = 0; m < M; ++m) memcpy(out + m * sub, cb + (m * ncent + codes[m]) * sub, sub * sizeof(float));
}

/* pq.rs:329-351 */
void so_pq_build_distance_table(const float *cb, size_t M, size_t ncent, size_t sub, const float *q, float *table) {
    for (size_t m = 0; m < M; ++m)
        for (size_t c = 0; c < ncent; ++c)
            table[m * ncent + c] = so_squared_l2(q + m * sub, cb + (m * ncent + c) * sub, sub);
}

/* pq.rs:358-368 */
float so_pq_distance_with_table(const float *table, size_t M, size_t ncent, const uint8_t *codes, size_t ncodes) {
    float total = 0.0f;
    for (size_t m = 0; m < ncodes; ++m) {
        if (m >= M || (size_t)codes[m] >= ncent) return FLT_MAX;
        total = total + table[m * ncent + codes[m]];
    }
    return total;
}

/* pq.rs:297-324 */
float so_pq_asymmetric_distance(const float *cb, size_t M, size_t ncent, size_t sub, const float *q, const uint8_t *codes) {
    float total = 0.0f;
    for (size_t m = 0; m < M; ++m) total = total + so_squared_l2(q + m * sub, cb + (m * ncent + codes[m]) * sub, sub);
    return total;
}

/* pq.rs:152-217 (init from a supplied shuffle; no early exit) */
void so_pq_kmeans(const float *vectors, size_t n, size_t dim, size_t k, size_t iterations,
                  const uint32_t *init_perm, float *cent) {
    for (size_t c = 0; c < k; ++c) memcpy(cent + c * dim, vectors + (size_t)init_perm[c % n] * dim, dim * sizeof(float));
    uint32_t *assign = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    float *newc = (float *)malloc(k * dim * sizeof(float));
    size_t *counts = (size_t *)malloc(k * sizeof(size_t));
    for (size_t it = 0; it < iterations; ++it) {
        for (size_t i = 0; i < n; ++i) {
            size_t best = 0; float bd = FLT_MAX;
            for (size_t c = 0; c < k; ++c) {
                float d = so_squared_l2(vectors + i * dim, cent + c * dim, dim);
                if (d < bd) { bd = d; best = c; }
            }
            assign[i] = (uint32_t)best;
        }
        memset(newc, 0, k * dim * sizeof(float));
        memset(counts, 0, k * sizeof(size_t));
        for (size_t i = 0; i < n; ++i) {
            size_t c = assign[i];
            counts[c] += 1;
            for (size_t j = 0; j < dim; ++j) newc[c * dim + j] = newc[c * dim + j] + vectors[i * dim + j];
        }
        for (size_t c = 0; c < k; ++c) if (counts[c] > 0) {
            for (size_t j = 0; j < dim; ++j) cent[c * dim + j] = newc[c * dim + j] / (float)counts[c];
        }
    }
    free(assign); free(newc); free(counts);
}

/* ------------------------------------------------------------------------------------ */
/* spann.rs                                                                              */
/* ------------------------------------------------------------------------------------ */

/* vector_db/spann.rs:562-571 (strictly sequential iterator sums; NOT distance_inline) */
float so_spann_compute_distance(const float *a, const float *b, size_t n, int metric) {
    if (metric == SO_METRIC_EUCLIDEAN) return so_squared_l2(a, b, n);
    float dot = 0.0f;
    for (size_t i = 0; i < n; ++i) dot = dot + a[i] * b[i];
    return 1.0f - dot;
}

/* spann.rs:545-558 */
size_t so_spann_find_nearest_centroid(const float *v, const float *centroids, size_t P, size_t dim, int metric) {
    size_t best = 0; float bd = FLT_MAX;
    for (size_t i = 0; i < P; ++i) {
        float d = so_spann_compute_distance(v, centroids + i * dim, dim, metric);
        if (d < bd) { bd = d; best = i; }
    }
    return best;
}

/* spann.rs:136-139 */
size_t so_spann_compute_partitions(size_t num_vectors) {
    size_t p = (size_t)ceil(sqrt((double)num_vectors));
    return p < 1 ? 1 : p;
}

/* spann.rs:466-541 */
size_t so_spann_kmeans(const float *vectors, size_t n, size_t dim, size_t k, size_t iterations,
                       int metric, const uint32_t *init_perm, float *cent) {
    for (size_t c = 0; c < k; ++c) memcpy(cent + c * dim, vectors + (size_t)init_perm[c % n] * dim, dim * sizeof(float));
    uint32_t *assign = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    float *newc = (float *)malloc(k * dim * sizeof(float));
    size_t *counts = (size_t *)malloc(k * sizeof(size_t));
    size_t it = 0;
    for (; it < iterations; ++it) {
        size_t changed = 0;
        for (size_t i = 0; i < n; ++i) {
            uint32_t na = (uint32_t)so_spann_find_nearest_centroid(vectors + i * dim, cent, k, dim, metric);
            if (na != assign[i]) ++changed;
            assign[i] = na;
        }
        memset(newc, 0, k * dim * sizeof(float));
        memset(counts, 0, k * sizeof(size_t));
        for (size_t i = 0; i < n; ++i) {
            size_t c = assign[i];
            counts[c] += 1;
            for (size_t j = 0; j < dim; ++j) newc[c * dim + j] = newc[c * dim + j] + vectors[i * dim + j];
        }
        for (size_t c = 0; c < k; ++c) if (counts[c] > 0)
            for (size_t j = 0; j < dim; ++j) cent[c * dim + j] = newc[c * dim + j] / (float)counts[c];
        if (changed == 0) { ++it; break; }
    }
    free(assign); free(newc); free(counts);
    return it;
}

/* max-heap on (OrderedFloat(dist), id) == std BinaryHeap<(OrderedFloat<f32>, u32)> */
static int heap_less(const idd_t *a, const idd_t *b) {
    int c = ordered_float_cmp(a->d, b->d);
    if (c) return c < 0;
    return a->id < b->id;
}
static void heap_push(idd_t *h, size_t *n, idd_t x) {
    size_t i = (*n)++;
    h[i] = x;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (heap_less(&h[p], &h[i])) { idd_t t = h[p]; h[p] = h[i]; h[i] = t; i = p; } else break;
    }
}
static void heap_pop(idd_t *h, size_t *n) {
    if (*n == 0) return;
    h[0] = h[--(*n)];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && heap_less(&h[m], &h[l])) m = l;
        if (r < *n && heap_less(&h[m], &h[r])) m = r;
        if (m == i) break;
        idd_t t = h[m]; h[m] = h[i]; h[i] = t; i = m;
    }
}

typedef struct { size_t i; float d; } pd_t;
static int pd_cmp(const void *pa, const void *pb) {
    const pd_t *a = (const pd_t *)pa, *b = (const pd_t *)pb;
    int c = so_total_cmp(a->d, b->d);
    if (c) return c;
    return (a->i > b->i) - (a->i < b->i);
}

/* spann.rs:574-693 */
size_t so_spann_search(const float *centroids, size_t P, size_t dim, int metric,
                       const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes,
                       const float *codebook, size_t M, size_t ncent, size_t sub,
                       size_t num_probes, const float *q, size_t k,
                       uint32_t *out_ids, float *out_dist) {
    if (P == 0) return 0;
    pd_t *pd = (pd_t *)malloc(P * sizeof(pd_t));
    for (size_t i = 0; i < P; ++i) { pd[i].i = i; pd[i].d = so_spann_compute_distance(q, centroids + i * dim, dim, metric); }
    qsort(pd, P, sizeof(pd_t), pd_cmp);
    size_t np = num_probes < P ? num_probes : P;
    float *table = (float *)malloc(M * ncent * sizeof(float));
    so_pq_build_distance_table(codebook, M, ncent, sub, q, table);
    idd_t *heap = (idd_t *)malloc((k + 2) * sizeof(idd_t));
    size_t hn = 0;
    for (size_t p = 0; p < np; ++p) {
        size_t part = pd[p].i;
        for (uint64_t e = list_off[part]; e < list_off[part + 1]; ++e) {
            idd_t x;
            x.d = so_pq_distance_with_table(table, M, ncent, codes + e * M, M);
            x.id = ids[e];
            heap_push(heap, &hn, x);
            if (hn > k) heap_pop(heap, &hn);
        }
    }
    qsort(heap, hn, sizeof(idd_t), idd_cmp);   /* (dist total_cmp asc, id asc) */
    for (size_t i = 0; i < hn; ++i) { out_ids[i] = heap[i].id; out_dist[i] = heap[i].d; }
    free(pd); free(table); free(heap);
    return hn;
}

/* ------------------------------------------------------------------------------------ */
/* minilm.rs: mean-pool + finalize_pooled, hash embedder                                 */
/* ------------------------------------------------------------------------------------ */

/* embeddings/minilm.rs:959-978 + :846-878 (native-384 branch: no prenorm, no truncation) */
void so_mean_pool_finalize(const float *hidden, const int64_t *mask, size_t S, size_t H, float *out) {
    float mask_sum = 0.0f;
    for (size_t d = 0; d < H; ++d) out[d] = 0.0f;
    for (size_t s = 0; s < S; ++s) {
        if (mask[s] == 1) {
            for (size_t d = 0; d < H; ++d) out[d] = out[d] + hidden[s * H + d];
            mask_sum = mask_sum + 1.0f;
        }
    }
    if (mask_sum > 0.0f) for (size_t d = 0; d < H; ++d) out[d] = out[d] / mask_sum;
    for (size_t d = 0; d < H; ++d) if (isnan(out[d]) || isinf(out[d])) out[d] = 0.0f;
    float nsq = 0.0f;
    for (size_t d = 0; d < H; ++d) nsq = nsq + out[d] * out[d];
    float norm = sqrtf(nsq);
    if (norm > FLT_EPSILON && !isnan(norm)) for (size_t d = 0; d < H; ++d) out[d] = out[d] / norm;
}

/* MiniLMEmbedder::finalize_pooled (minilm.rs:846-878) with the nomic branch: NaN/Inf scrub; if apply_prenorm: parameter-free
 * LayerNorm over the FULL width (mean = sum/n, var = sum((x-mean)^2)/n, denom = sqrt(var + 1e-5), applied when denom > EPSILON);
 * truncate to out_dim (Matryoshka); L2-normalise the kept prefix when norm > EPSILON. Returns the output length. */
size_t so_finalize_pooled(const float *pooled, size_t n, int apply_prenorm, size_t out_dim, float *out) {
    float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; ++i) tmp[i] = (isnan(pooled[i]) || isinf(pooled[i])) ? 0.0f : pooled[i];
    if (apply_prenorm) {
        const float nf = (float)n;
        float sum = 0.0f;
        for (size_t i = 0; i < n; ++i) sum = sum + tmp[i];
        const float mean = sum / nf;
        float vs = 0.0f;
        for (size_t i = 0; i < n; ++i) vs = vs + (tmp[i] - mean) * (tmp[i] - mean);
        const float var = vs / nf;
        const float denom = sqrtf(var + 1e-5f);
        if (denom > FLT_EPSILON) for (size_t i = 0; i < n; ++i) tmp[i] = (tmp[i] - mean) / denom;
    }
    size_t m = n > out_dim ? out_dim : n;
    float nsq = 0.0f;
    for (size_t i = 0; i < m; ++i) nsq = nsq + tmp[i] * tmp[i];
    const float norm = sqrtf(nsq);
    const int ok = norm > FLT_EPSILON && !isnan(norm);
    for (size_t i = 0; i < m; ++i) out[i] = ok ? tmp[i] / norm : tmp[i];
    free(tmp);
    return m;
}

/* SipHash-1-3, keys (0,0): std::collections::hash_map::DefaultHasher::new() */
#define ROTL64(x, b) (((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND do { v0 += v1; v1 = ROTL64(v1, 13); v1 ^= v0; v0 = ROTL64(v0, 32); \
                      v2 += v3; v3 = ROTL64(v3, 16); v3 ^= v2; \
                      v0 += v3; v3 = ROTL64(v3, 21); v3 ^= v0; \
                      v2 += v1; v1 = ROTL64(v1, 17); v1 ^= v2; v2 = ROTL64(v2, 32); } while (0)

static uint64_t siphash13(const uint8_t *in, size_t len) {
    uint64_t k0 = 0, k1 = 0;
    uint64_t v0 = 0x736f6d6570736575ULL ^ k0, v1 = 0x646f72616e646f6dULL ^ k1;
    uint64_t v2 = 0x6c7967656e657261ULL ^ k0, v3 = 0x7465646279746573ULL ^ k1;
    size_t end = len - (len % 8);
    for (size_t i = 0; i < end; i += 8) {
        uint64_t m = 0;
        for (int j = 0; j < 8; ++j) m |= (uint64_t)in[i + j] << (8 * j);
        v3 ^= m; SIPROUND; v0 ^= m;
    }
    uint64_t b = (uint64_t)len << 56;
    for (size_t j = 0; j < len % 8; ++j) b |= (uint64_t)in[end + j] << (8 * j);
    v3 ^= b; SIPROUND; v0 ^= b;
    v2 ^= 0xff; SIPROUND; SIPROUND; SIPROUND;
    return v0 ^ v1 ^ v2 ^ v3;
}

/* impl Hash for str: state.write(bytes); state.write_u8(0xff) */
uint64_t so_siphash13_str(const uint8_t *bytes, size_t len) {
    uint8_t *buf = (uint8_t *)malloc(len + 1);
    memcpy(buf, bytes, len);
    buf[len] = 0xff;
    uint64_t h = siphash13(buf, len + 1);
    free(buf);
    return h;
}

static size_t utf8_len(uint8_t c) {
    if (c < 0x80) return 1;
    if ((c >> 5) == 0x6) return 2;
    if ((c >> 4) == 0xe) return 3;
    if ((c >> 3) == 0x1e) return 4;
    return 1;
}
static uint32_t utf8_decode(const uint8_t *p, size_t l) {
    if (l == 1) return p[0];
    if (l == 2) return ((p[0] & 0x1f) << 6) | (p[1] & 0x3f);
    if (l == 3) return ((p[0] & 0x0f) << 12) | ((p[1] & 0x3f) << 6) | (p[2] & 0x3f);
    return ((p[0] & 0x07) << 18) | ((p[1] & 0x3f) << 12) | ((p[2] & 0x3f) << 6) | (p[3] & 0x3f);
}
/* char::is_whitespace (Unicode White_Space) */
static int is_ws(uint32_t c) {
    return (c >= 9 && c <= 13) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F ||
           c == 0x205F || c == 0x3000;
}

/* embeddings/minilm.rs:777-831 generate_embedding_simplified + :749-771 normalize */
void so_hash_embed(const char *utf8, size_t len, size_t dim, float *e) {
    const uint8_t *t = (const uint8_t *)utf8;
    for (size_t j = 0; j < dim; ++j) e[j] = 0.0f;
    /* words */
    size_t pos = 0, wi = 0;
    while (pos < len) {
        while (pos < len) { size_t l = utf8_len(t[pos]); if (pos + l > len) l = len - pos; if (!is_ws(utf8_decode(t + pos, l))) break; pos += l; }
        if (pos >= len) break;
        size_t start = pos;
        while (pos < len) { size_t l = utf8_len(t[pos]); if (pos + l > len) l = len - pos; if (is_ws(utf8_decode(t + pos, l))) break; pos += l; }
        uint64_t h = so_siphash13_str(t + start, pos - start);
        for (size_t j = 0; j < dim; ++j) {
            size_t index = (wi * 7 + j) % dim;
            size_t bit = j < 64 ? j : (wi * 7 + j) % 64;
            e[index] = e[index] + (float)((h >> bit) & 1) * 0.1f;
        }
        ++wi;
    }
    /* char bigrams */
    size_t nchars = 0;
    for (size_t p = 0; p < len;) { size_t l = utf8_len(t[p]); if (p + l > len) l = len - p; p += l; ++nchars; }
    if (nchars >= 2) {
        size_t p = 0;
        for (size_t i = 0; i + 1 < nchars; ++i) {
            size_t l0 = utf8_len(t[p]); if (p + l0 > len) l0 = len - p;
            size_t p1 = p + l0;
            size_t l1 = utf8_len(t[p1]); if (p1 + l1 > len) l1 = len - p1;
            uint64_t h = so_siphash13_str(t + p, l0 + l1);
            for (size_t j = 0; j < 32; ++j) {
                size_t index = (size_t)((h + (uint64_t)j) % (uint64_t)dim);
                e[index] = e[index] + (float)((h >> (j % 64)) & 1) * 0.05f;
            }
            p = p1;
        }
    }
    /* normalize(): scrub, norm = sqrt(sum x*x) sequential, fail if NaN or < EPSILON */
    for (size_t j = 0; j < dim; ++j) if (isnan(e[j]) || isinf(e[j])) e[j] = 0.0f;
    float nsq = 0.0f;
    for (size_t j = 0; j < dim; ++j) nsq = nsq + e[j] * e[j];
    float norm = sqrtf(nsq);
    if (isnan(norm) || norm < FLT_EPSILON) { for (size_t j = 0; j < dim; ++j) e[j] = 0.0f; return; }
    for (size_t j = 0; j < dim; ++j) e[j] = e[j] / norm;
}

/* ------------------------------------------------------------------------------------ */
/* relevance.rs                                                                          */
/* ------------------------------------------------------------------------------------ */

/* relevance.rs:64-100, :383-397 */
void so_weights_default(so_weights *w) {
    w->semantic = 0.18f; w->entity = 0.17f; w->tag = 0.05f; w->importance = 0.05f;
    w->momentum = 0.28f; w->access_count = 0.14f; w->graph_strength = 0.13f; w->update_count = 0;
}

/* relevance.rs:401-418 */
void so_weights_normalize(so_weights *w) {
    float sum = w->semantic + w->entity;
    sum = sum + w->tag; sum = sum + w->importance; sum = sum + w->momentum;
    sum = sum + w->access_count; sum = sum + w->graph_strength;
    if (sum > 0.0f) {
        w->semantic /= sum; w->entity /= sum; w->tag /= sum; w->importance /= sum;
        w->momentum /= sum; w->access_count /= sum; w->graph_strength /= sum;
    }
}

static inline float f32_max(float a, float b) { return fmaxf(a, b); } /* f32::max ignores NaN like fmaxf */
static inline float f32_min(float a, float b) { return fminf(a, b); }

/* relevance.rs:427-465 */
void so_weights_apply_feedback(so_weights *w, int sem, int ent, int tag, int helpful) {
    const float LR = 0.05f, MINW = 0.05f;
    float direction = helpful ? 1.0f : -1.0f;
    float delta = LR * direction;
    if (sem) w->semantic = f32_max(w->semantic + delta, MINW);
    if (ent) w->entity = f32_max(w->entity + delta, MINW);
    if (tag) w->tag = f32_max(w->tag + delta, MINW);
    if (helpful && !sem && !ent && !tag) w->importance = f32_max(w->importance + delta, MINW);
    float aux = LR * direction * 0.5f;
    w->momentum = f32_max(w->momentum + aux, MINW);
    w->access_count = f32_max(w->access_count + aux, MINW);
    w->graph_strength = f32_max(w->graph_strength + aux, MINW);
    so_weights_normalize(w);
    w->update_count += 1;
}

/* relevance.rs:601-606 */
float so_calibrate_score(float s) {
    if (!isfinite(s)) return 0.0f;
    return 1.0f / (1.0f + expf(-10.0f * (s - 0.5f)));
}

/* relevance.rs:529-594 */
float so_fuse_scores_full(const so_weights *w, float sem, float ent, float tag, float imp,
                          float momentum_ema, uint32_t access_count, float graph_strength) {
    float c_sem = so_calibrate_score(sem);
    float c_ent = so_calibrate_score(ent);
    float c_tag = so_calibrate_score(tag);
    float c_imp = so_calibrate_score(imp);
    float nm = (momentum_ema + 1.0f) / 2.0f;
    float am;
    if (nm > 0.65f) am = f32_min(nm * 1.5f, 1.0f);
    else if (nm < 0.40f) am = f32_max(nm * 0.3f, 0.0f);
    else am = nm;
    float c_mom = so_calibrate_score(am);
    float as;
    if (access_count == 0) as = 0.0f;
    else { float la = log2f((float)access_count + 1.0f); as = f32_min(la / 4.0f, 1.0f); }
    float c_acc = so_calibrate_score(as);
    float c_gr = so_calibrate_score(graph_strength);
    float r = w->semantic * c_sem + w->entity * c_ent;
    r = r + w->tag * c_tag;
    r = r + w->importance * c_imp;
    r = r + w->momentum * c_mom;
    r = r + w->access_count * c_acc;
    r = r + w->graph_strength * c_gr;
    return isfinite(r) ? r : 0.0f;
}

/* relevance.rs:471-487 */
float so_fuse_scores(const so_weights *w, float sem, float ent, float tag, float imp) {
    return so_fuse_scores_full(w, sem, ent, tag, imp, 0.0f, 0, 0.5f);
}
/* relevance.rs:499-517 */
float so_fuse_scores_with_momentum(const so_weights *w, float sem, float ent, float tag, float imp, float mom) {
    return so_fuse_scores_full(w, sem, ent, tag, imp, mom, 0, 0.5f);
}

/* str::to_lowercase (library/alloc/src/str.rs): every char through its full Unicode lower-case mapping, U+03A3 through the
 * Final_Sigma rule (map_uppercase_sigma / case_ignorable_then_cased). Data: oracle/unicode_lower.inc, generated from Python's
 * Unicode 13 database by tools/gen_unicode_lower.py. Works on code points; returns a malloc'ed code-point array. */
#include "unicode_lower.inc"
static int cp_in(const unsigned int (*tab)[2], unsigned int n, uint32_t c) {
    for (unsigned int lo = 0, hi = n; lo < hi;) {
        unsigned int mid = lo + (hi - lo) / 2;
        if (c < tab[mid][0]) hi = mid; else if (c > tab[mid][1]) lo = mid + 1; else return 1;
    }
    return 0;
}
static uint32_t *decode_cps(const char *s, size_t *n_out) {
    size_t len = strlen(s), n = 0;
    uint32_t *cps = (uint32_t *)malloc((len + 1) * sizeof(uint32_t));
    for (size_t pos = 0; pos < len;) {
        size_t l = utf8_len((uint8_t)s[pos]);
        if (pos + l > len) l = len - pos;
        cps[n++] = utf8_decode((const uint8_t *)s + pos, l);
        pos += l;
    }
    *n_out = n;
    return cps;
}
static uint32_t *lowercase_cps(const uint32_t *in, size_t n, size_t *n_out) {
    uint32_t *out = (uint32_t *)malloc((3 * n + 1) * sizeof(uint32_t));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = in[i];
        if (c == 0x03A3) {
            /* is_word_final = case_ignorable_then_cased(from[..i].chars().rev()) && !case_ignorable_then_cased(from[i+2..].chars()) */
            int before = 0, after = 0;
            for (size_t j = i; j-- > 0;) { if (cp_in(UCASE_IGNORABLE, UCASE_IGNORABLE_N, in[j])) continue; before = cp_in(UCASED, UCASED_N, in[j]); break; }
            for (size_t j = i + 1; j < n; ++j) { if (cp_in(UCASE_IGNORABLE, UCASE_IGNORABLE_N, in[j])) continue; after = cp_in(UCASED, UCASED_N, in[j]); break; }
            out[m++] = (before && !after) ? 0x03C2 : 0x03C3;
            continue;
        }
        int mapped = 0;
        for (unsigned int lo = 0, hi = ULOWER_N; lo < hi;) {
            unsigned int mid = lo + (hi - lo) / 2;
            if (ULOWER_FROM[mid] == c) { for (int j = 0; j < 3; ++j) if (ULOWER_TO[mid][j]) out[m++] = ULOWER_TO[mid][j]; mapped = 1; break; }
            if (ULOWER_FROM[mid] < c) lo = mid + 1; else hi = mid;
        }
        if (!mapped) out[m++] = c;
    }
    *n_out = m;
    return out;
}
static size_t find_cps(const uint32_t *hay, size_t nh, const uint32_t *needle, size_t nn) {   /* str::contains on chars == on bytes */
    if (nn == 0) return 0;
    for (size_t i = 0; i + nn <= nh; ++i) if (memcmp(hay + i, needle, nn * sizeof(uint32_t)) == 0) return i;
    return (size_t)-1;
}

/* relevance.rs:680-705 */
float so_calculate_tag_score(const char *content, const char *const *tags, size_t n_tags) {
    if (n_tags == 0) return 0.0f;
    size_t nc0, nc;
    uint32_t *c0 = decode_cps(content, &nc0);
    uint32_t *ctx = lowercase_cps(c0, nc0, &nc);
    free(c0);
    int matches = 0;
    for (size_t t = 0; t < n_tags; ++t) {
        size_t nt0, nt;
        uint32_t *t0 = decode_cps(tags[t], &nt0);
        uint32_t *tag = lowercase_cps(t0, nt0, &nt);
        free(t0);
        if (find_cps(ctx, nc, tag, nt) != (size_t)-1) { matches += 1; free(tag); continue; }
        size_t pos = 0;
        while (pos < nc) {                                        /* split_whitespace */
            while (pos < nc && is_ws(ctx[pos])) ++pos;
            if (pos >= nc) break;
            size_t st = pos;
            while (pos < nc && !is_ws(ctx[pos])) ++pos;
            size_t wl = pos - st;
            int word_starts_with_tag = wl >= nt && memcmp(ctx + st, tag, nt * sizeof(uint32_t)) == 0;
            int tag_starts_with_word = nt >= wl && memcmp(tag, ctx + st, wl * sizeof(uint32_t)) == 0;
            if (word_starts_with_tag || tag_starts_with_word) { matches += 1; break; }
        }
        free(tag);
    }
    free(ctx);
    return (float)matches / (float)n_tags;
}

/* relevance.rs:1524-1547 (age_hours = Duration::num_hours() as u64: negative wraps to huge) */
float so_apply_recency_boost(float base, int64_t age_hours_signed, uint64_t boost_hours, float mult) {
    if (boost_hours == 0) return base;
    uint64_t age_hours = (uint64_t)age_hours_signed;
    if (age_hours <= boost_hours) {
        float decay = 1.0f - ((float)age_hours / (float)boost_hours);
        float boost = 1.0f + (mult - 1.0f) * decay;
        return f32_min(base * boost, 1.0f);
    }
    return base;
}

/* relevance.rs:801-918 */
typedef struct { uint32_t index; float relevance_score; int64_t created_at; uint8_t id[16]; uint8_t reason; } surfaced_t;
static int surfaced_cmp(const void *pa, const void *pb) {
    const surfaced_t *a = (const surfaced_t *)pa, *b = (const surfaced_t *)pb;
    int c = so_total_cmp(b->relevance_score, a->relevance_score);          /* b.relevance_score.total_cmp(&a.relevance_score) */
    if (c) return c;
    if (b->created_at != a->created_at) return b->created_at < a->created_at ? -1 : 1;   /* b.created_at.cmp(&a.created_at) */
    return memcmp(a->id, b->id, 16);                                       /* a.id.cmp(&b.id): hyphenated lower-case hex orders like the bytes */
}
size_t so_rank_surfaced(const so_weights *w, float min_importance, uint64_t recency_boost_hours, float recency_boost_multiplier,
                        float graph_boost_multiplier, size_t max_results, size_t n, const float *semantic, const float *entity,
                        const float *tag, const float *importance, const float *momentum_ema, const uint32_t *access_count,
                        const float *graph_strength, const int64_t *age_hours, const int64_t *created_at_ns, const uint8_t *uuid,
                        uint32_t *out_index, float *out_score, uint8_t *out_reason) {
    surfaced_t *results = (surfaced_t *)malloc((n ? n : 1) * sizeof(surfaced_t));
    size_t nr = 0;
    for (size_t i = 0; i < n; ++i) {
        float imp = importance[i];
        if (imp < min_importance) continue;
        float semantic_score = semantic[i], entity_score = entity[i];
        float fused_score = so_fuse_scores_full(w, semantic_score, entity_score, tag[i], imp, momentum_ema[i], access_count[i], graph_strength[i]);
        uint8_t reason;
        if (semantic_score > 0.0f && entity_score > 0.0f) reason = 0;
        else if (entity_score > 0.0f) reason = 1;
        else if (semantic_score > 0.0f) reason = 2;
        else reason = 3;
        float recency_boosted = so_apply_recency_boost(fused_score, age_hours[i], recency_boost_hours, recency_boost_multiplier);
        float final_score = recency_boosted;
        if (entity_score > 0.0f) final_score = f32_min(recency_boosted * graph_boost_multiplier, 1.0f);
        results[nr].index = (uint32_t)i; results[nr].relevance_score = final_score; results[nr].created_at = created_at_ns[i];
        memcpy(results[nr].id, uuid + i * 16, 16); results[nr].reason = reason;
        ++nr;
    }
    qsort(results, nr, sizeof(surfaced_t), surfaced_cmp);      /* keys are unique down to the id, so stability does not matter */
    size_t kept = 0;
    for (size_t i = 0; i < nr; ++i) if (results[i].relevance_score >= 0.25f) results[kept++] = results[i];
    if (kept > max_results) kept = max_results;
    for (size_t i = 0; i < kept; ++i) { out_index[i] = results[i].index; out_score[i] = results[i].relevance_score; out_reason[i] = results[i].reason; }
    free(results);
    return kept;
}

/* ------------------------------------------------------------------------------------ */
/* hybrid_search.rs RRFusion                                                             */
/* ------------------------------------------------------------------------------------ */

/* memory/hybrid_search.rs:536-594 */
size_t so_rrf_fuse(float k, const float *weights, size_t n_lists, const uint8_t *uuids,
                   const size_t *list_len, uint8_t *out_uuid, float *out_score, size_t out_cap) {
    float *wn = (float *)malloc((n_lists ? n_lists : 1) * sizeof(float));
    float sum = 0.0f;
    for (size_t l = 0; l < n_lists; ++l) sum = sum + weights[l];
    for (size_t l = 0; l < n_lists; ++l) wn[l] = sum > 0.0f ? weights[l] / sum : 1.0f / (float)n_lists;
    size_t total = 0;
    for (size_t l = 0; l < n_lists; ++l) total += list_len[l];
    uuid_score_t *acc = (uuid_score_t *)malloc((total ? total : 1) * sizeof(uuid_score_t));
    size_t na = 0, base = 0;
    for (size_t l = 0; l < n_lists; ++l) {
        for (size_t rank = 0; rank < list_len[l]; ++rank) {
            const uint8_t *u = uuids + (base + rank) * 16;
            float contrib = wn[l] / (k + (float)(rank + 1));
            size_t j = 0;
            for (; j < na; ++j) if (memcmp(acc[j].u, u, 16) == 0) break;
            if (j == na) { memcpy(acc[na].u, u, 16); acc[na].s = 0.0f; ++na; }
            acc[j].s = acc[j].s + contrib;
        }
        base += list_len[l];
    }
    qsort(acc, na, sizeof(uuid_score_t), uuid_score_cmp_desc);
    size_t out = na < out_cap ? na : out_cap;
    for (size_t i = 0; i < out; ++i) { memcpy(out_uuid + i * 16, acc[i].u, 16); out_score[i] = acc[i].s; }
    free(acc); free(wn);
    return out;
}

/* ------------------------------------------------------------------------------------ */
/* memory/mod.rs recall Layer 4: hybrid + graph leg fusion                               */
/* ------------------------------------------------------------------------------------ */

static inline float f32_clamp(float x, float lo, float hi) { if (x < lo) return lo; if (x > hi) return hi; return x; }  /* NaN stays */

/* memory/graph_retrieval.rs:81-101 with constants.rs:478-510 */
void so_density_weights(float graph_density, float out3[3]) {
    const float GW_MIN = 0.1f, GW_MAX = 0.5f, LING = 0.15f, TH_MIN = 0.5f, TH_MAX = 2.0f;
    float graph_weight;
    if (graph_density <= TH_MIN) graph_weight = GW_MAX;
    else if (graph_density >= TH_MAX) graph_weight = GW_MIN;
    else {
        float ratio = (graph_density - TH_MIN) / (TH_MAX - TH_MIN);
        graph_weight = GW_MAX - ratio * (GW_MAX - GW_MIN);
    }
    out3[2] = LING;
    out3[0] = 1.0f - graph_weight - LING;
    out3[1] = graph_weight;
}

/* memory/mod.rs:3878-3921 */
void so_leg_fusion_weights(int has_density, float graph_density, float graph_weight_override, float graph_w_floor,
                           float *graph_w_out, float *hybrid_w_out) {
    float semantic_w = 0.6f, graph_w = 0.3f, linguistic_w = 0.1f;
    if (has_density) { float w[3]; so_density_weights(graph_density, w); semantic_w = w[0]; graph_w = w[1]; linguistic_w = w[2]; }
    if (!isnan(graph_weight_override)) graph_w = f32_clamp(graph_weight_override, 0.0f, 1.0f);
    if (!isnan(graph_w_floor)) {
        float f = graph_w_floor;
        if (f > graph_w) {
            float max_floor = f32_max(1.0f - linguistic_w - 0.05f, 0.0f);
            float requested = f32_clamp(f, 0.0f, 0.95f);
            graph_w = f32_min(requested, max_floor);
            semantic_w = f32_max(1.0f - graph_w - linguistic_w, 0.0f);
        }
    }
    *graph_w_out = graph_w;
    *hybrid_w_out = semantic_w + linguistic_w;
}

typedef struct { int h; float s; } leg_item_t;                 /* (handle of the id, score) */
static const uint8_t *g_leg_ids;                               /* id table the comparator looks handles up in */
static int leg_item_cmp(const void *pa, const void *pb) {      /* b.1.total_cmp(&a.1).then_with(|| a.0.cmp(b.0)) */
    const leg_item_t *a = (const leg_item_t *)pa, *b = (const leg_item_t *)pb;
    int c = so_total_cmp(b->s, a->s);
    if (c) return c;
    return memcmp(g_leg_ids + (size_t)a->h * 16, g_leg_ids + (size_t)b->h * 16, 16);
}
static float leg_peak(const leg_item_t *xs, size_t n) {        /* mod.rs:4123-4134 */
    if (n == 0) return 1.0f;
    float max = xs[0].s, sum = 0.0f;
    for (size_t i = 0; i < n; ++i) sum = sum + xs[i].s;
    float mean = sum / (float)n;
    if (mean > 1e-6f) return max / mean;
    return 1.0f;
}
static float leg_overlap(const leg_item_t *by_vec, size_t nv, const leg_item_t *by_bm, size_t nb, size_t k) {
    if (nv < k) k = nv;
    if (nb < k) k = nb;
    if (k < 1) k = 1;
    size_t count = 0;
    for (size_t i = 0; i < k && i < nb; ++i) {
        int in_top_v = 0;
        for (size_t j = 0; j < k && j < nv; ++j) if (by_vec[j].h == by_bm[i].h) in_top_v = 1;
        count += (size_t)in_top_v;
    }
    return (float)count / (float)k;
}

size_t so_fuse_legs(const int sw[8], const float fp[13], const uint8_t *hybrid_uuid, const float *hybrid_bm25,
                    const float *hybrid_vec, size_t n_hybrid, const uint8_t *graph_uuid, const float *graph_activation,
                    size_t n_graph, size_t query_len, uint8_t *out_uuid, float *out_score, size_t out_cap, float *vec_trust_out) {
    const int v2_fusion = sw[0], explicit_flat = sw[1], sum_fusion = sw[2], rrf_escape = sw[3], leg = sw[4];
    const int flat_adaptive = sw[5], adapt_feature = sw[6], symmetric = sw[7];
    const float graph_w = fp[0], hybrid_w = fp[1], k = fp[2];
    const float flat_consensus = f32_clamp(fp[3], 0.0f, 1.0f), adapt_trust_max = fp[4];
    const float fw_graph = fp[5], fw_vec = fp[6], fw_bm25 = fp[7];
    /* one table of distinct ids (hybrid first, then graph); everything below works on handles into it */
    size_t cap = n_hybrid + n_graph + 1, nid = 0;
    uint8_t *ids = (uint8_t *)malloc(cap * 16);
    int *hh = (int *)malloc((n_hybrid + 1) * sizeof(int)), *gh = (int *)malloc((n_graph + 1) * sizeof(int));
    float *c_bm = (float *)calloc(cap, sizeof(float)), *c_vec = (float *)calloc(cap, sizeof(float));
    char *in_comp = (char *)calloc(cap, 1), *in_fused = (char *)calloc(cap, 1);
    float *fused = (float *)calloc(cap, sizeof(float));
    size_t *first_seen = (size_t *)calloc(cap, sizeof(size_t));
    for (size_t i = 0; i < n_hybrid + n_graph; ++i) {
        const uint8_t *u = i < n_hybrid ? hybrid_uuid + i * 16 : graph_uuid + (i - n_hybrid) * 16;
        size_t j = 0;
        while (j < nid && memcmp(ids + j * 16, u, 16) != 0) ++j;
        if (j == nid) { memcpy(ids + nid * 16, u, 16); ++nid; }
        if (i < n_hybrid) hh[i] = (int)j; else gh[i - n_hybrid] = (int)j;
    }
    /* hybrid_components: HashMap insert, the last (bm25, vector) of an id wins (:3832-3838) */
    size_t order_n = 0;
    for (size_t i = 0; i < n_hybrid; ++i) {
        int h = hh[i];
        if (!in_comp[h]) { in_comp[h] = 1; first_seen[order_n++] = (size_t)h; }
        c_bm[h] = hybrid_bm25[i]; c_vec[h] = hybrid_vec[i];
    }
    float max_activation = 0.0f;
    for (size_t i = 0; i < n_graph; ++i) max_activation = f32_max(max_activation, graph_activation[i]);
    float graph_max_act = max_activation;
    max_activation = f32_max(max_activation, 1e-6f);
    float n_graph_f = (float)(n_graph > 0 ? n_graph : 1), n_hybrid_f = (float)(n_hybrid > 0 ? n_hybrid : 1);
    /* SHODH_LEG (:3975-3988) */
    for (size_t o = 0; o < order_n; ++o) {
        size_t h = first_seen[o];
        if (leg == 1) { if (!(c_vec[h] > 0.0f)) in_comp[h] = 0; c_bm[h] = 0.0f; }
        else if (leg == 2) { if (!(c_bm[h] > 0.0f)) in_comp[h] = 0; c_vec[h] = 0.0f; }
        else if (leg == 3) in_comp[h] = 0;
    }
    int graph_leg_on = !(leg == 1 || leg == 2);
    float max_vec = 0.0f, max_bm = 0.0f;
    size_t n_comp = 0;
    for (size_t o = 0; o < order_n; ++o) {
        size_t h = first_seen[o];
        if (!in_comp[h]) continue;
        ++n_comp;
        max_vec = f32_max(max_vec, c_vec[h]); max_bm = f32_max(max_bm, c_bm[h]);
    }
    max_vec = f32_max(max_vec, 1e-6f); max_bm = f32_max(max_bm, 1e-6f);
    int flat_fusion = explicit_flat || (!rrf_escape && !v2_fusion && !sum_fusion);

    float effective_vec_trust = 1.0f;
    if (flat_adaptive) {
        leg_item_t *by_vec = (leg_item_t *)malloc((order_n + 1) * sizeof(leg_item_t));
        leg_item_t *by_bm = (leg_item_t *)malloc((order_n + 1) * sizeof(leg_item_t));
        size_t nv = 0, nb = 0;
        for (size_t o = 0; o < order_n; ++o) {
            size_t h = first_seen[o];
            if (!in_comp[h]) continue;
            if (c_vec[h] > 0.0f) { by_vec[nv].h = (int)h; by_vec[nv].s = c_vec[h]; ++nv; }
            if (c_bm[h] > 0.0f) { by_bm[nb].h = (int)h; by_bm[nb].s = c_bm[h]; ++nb; }
        }
        g_leg_ids = ids;
        qsort(by_vec, nv, sizeof(leg_item_t), leg_item_cmp);
        qsort(by_bm, nb, sizeof(leg_item_t), leg_item_cmp);
        float t;
        if (adapt_feature == 0) {                              /* fitted gate (:4093-4168) */
            static const float FIT[11][3] = {
                {2.77242f, 1.87083f, -0.301375f}, {1.3841f, 0.180661f, -0.0212517f}, {0.307389f, 0.170755f, -0.243719f},
                {93.4129f, 43.8602f, -0.556471f}, {0.597371f, 0.0823258f, 0.236463f}, {106.897f, 36.4931f, 0.597537f},
                {31.4138f, 7.54534f, -0.881755f}, {116.222f, 38.3149f, 0.582615f}, {9.90148f, 0.987682f, 0.304049f},
                {0.358194f, 0.0710801f, 0.0264384f}, {54.1034f, 16.0475f, -0.571797f}};
            float agreement = (nv == 0 || nb == 0) ? 0.0f : leg_overlap(by_vec, nv, by_bm, nb, 10);
            float feats[11];
            feats[0] = leg_peak(by_bm, nb); feats[1] = leg_peak(by_vec, nv); feats[2] = agreement; feats[3] = max_bm;
            feats[4] = max_vec; feats[5] = (float)nb; feats[6] = (float)nv; feats[7] = (float)n_comp; feats[8] = (float)n_graph;
            feats[9] = graph_max_act; feats[10] = (float)query_len;
            float s = -0.985886f;
            for (int i = 0; i < 11; ++i) { float z = FIT[i][2] * (feats[i] - FIT[i][0]); s = s + z / FIT[i][1]; }
            t = 1.0f / (1.0f + expf(-f32_clamp(s, -30.0f, 30.0f)));
        } else if (adapt_feature == 1) {                       /* agreement (:4169-4205) */
            float kf = f32_max(fp[8], 1.0f);
            size_t agree_k = kf >= 1.8446744e19f ? (size_t)-1 : (size_t)kf;
            if (nb == 0) t = 1.0f;
            else if (nv == 0) t = 0.0f;
            else {
                float overlap = leg_overlap(by_vec, nv, by_bm, nb, agree_k);
                float span = f32_max(fp[10] - fp[9], 1e-6f);
                t = f32_clamp((fp[10] - overlap) / span, 0.0f, 1.0f);
            }
        } else {                                               /* BM25 peakedness (:4206-4228); positive scores summed in first-insertion order */
            float sum = 0.0f; size_t cnt = 0;
            for (size_t o = 0; o < order_n; ++o) { size_t h = first_seen[o]; if (in_comp[h] && c_bm[h] > 0.0f) { sum = sum + c_bm[h]; ++cnt; } }
            float mean_bm = cnt == 0 ? 0.0f : sum / (float)cnt;
            float bm_peak = mean_bm > 1e-6f ? max_bm / mean_bm : 1.0f;
            float span = f32_max(fp[12] - fp[11], 1e-6f);
            t = f32_clamp((fp[12] - bm_peak) / span, 0.0f, 1.0f);
        }
        if (symmetric) effective_vec_trust = f32_max(1.0f + (adapt_trust_max - 1.0f) * (2.0f * t - 1.0f), 0.2f);
        else effective_vec_trust = 1.0f + (adapt_trust_max - 1.0f) * t;
        free(by_vec); free(by_bm);
    }
    if (vec_trust_out) *vec_trust_out = effective_vec_trust;

    /* graph leg (:4348-4413) */
    for (size_t r = 0; r < n_graph; ++r) {
        if (!graph_leg_on) break;
        float activation = graph_activation[r], rrf_score;
        if (sum_fusion) rrf_score = fw_graph * f32_clamp(activation / max_activation, 0.0f, 1.0f);
        else if (v2_fusion) {
            float borda = graph_w * ((n_graph_f - (float)r) / n_graph_f);
            float rescue = graph_w * f32_clamp(activation / max_activation, 0.0f, 1.0f);
            rrf_score = borda + rescue;
        } else if (flat_fusion) rrf_score = graph_w * f32_clamp(activation / max_activation, 0.0f, 1.0f);
        else rrf_score = graph_w / (k + (float)(r + 1));
        int h = gh[r];
        in_fused[h] = 1;
        fused[h] = fused[h] + rrf_score;
        float activation_factor = 1.0f;
        if (!(v2_fusion || sum_fusion)) activation_factor = 1.0f + graph_w * 0.3f * f32_clamp(activation, 0.0f, 1.0f);
        fused[h] = fused[h] * activation_factor;
    }
    /* hybrid leg (:4414-4468) */
    for (size_t r = 0; r < n_hybrid; ++r) {
        int h = hh[r];
        float bm25 = in_comp[h] ? c_bm[h] : 0.0f, vec = in_comp[h] ? c_vec[h] : 0.0f;
        float hybrid_rrf;
        if (sum_fusion) hybrid_rrf = fw_vec * f32_clamp(vec / max_vec, 0.0f, 1.0f) + fw_bm25 * f32_clamp(bm25 / max_bm, 0.0f, 1.0f);
        else if (flat_fusion) {
            float vn = f32_clamp(vec / max_vec, 0.0f, 1.0f) * effective_vec_trust;
            float bn = f32_clamp(bm25 / max_bm, 0.0f, 1.0f);
            float hi, lo;
            if (vn >= bn) { hi = vn; lo = bn; } else { hi = bn; lo = vn; }
            hybrid_rrf = hybrid_w * (hi + flat_consensus * lo);
        } else if (v2_fusion) hybrid_rrf = hybrid_w * ((n_hybrid_f - (float)r) / n_hybrid_f);
        else hybrid_rrf = hybrid_w / (k + (float)(r + 1));
        in_fused[h] = 1;
        fused[h] = fused[h] + hybrid_rrf;
    }
    uuid_score_t *res = (uuid_score_t *)malloc((nid + 1) * sizeof(uuid_score_t));
    size_t nres = 0;
    for (size_t h = 0; h < nid; ++h) if (in_fused[h]) { memcpy(res[nres].u, ids + h * 16, 16); res[nres].s = fused[h]; ++nres; }
    qsort(res, nres, sizeof(uuid_score_t), uuid_score_cmp_desc);
    for (size_t i = 0; i < nres && i < out_cap; ++i) { memcpy(out_uuid + i * 16, res[i].u, 16); out_score[i] = res[i].s; }
    free(res); free(first_seen); free(fused); free(in_fused); free(in_comp); free(c_vec); free(c_bm); free(gh); free(hh); free(ids);
    return nres;
}

/* vector_db/vamana_persist.rs:155-163 */
uint64_t so_fnv1a64(const uint8_t *data, size_t len) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < len; ++i) { h ^= data[i]; h *= 0x100000001b3ULL; }
    return h;
}

/* ------------------------------------------------------------------------------------ */
/* threaded CPU-baseline driver                                                          */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    const float *rows; size_t n, dim; const float *queries; size_t nq, k; int order, full_sort;
    uint32_t *out_ids; float *out_dist; size_t *next; pthread_mutex_t *mu; const uint8_t *deleted;
} bf_job_t;

static void *bf_worker(void *arg) {
    bf_job_t *j = (bf_job_t *)arg;
    for (;;) {
        pthread_mutex_lock(j->mu);
        size_t qi = (*j->next)++;
        pthread_mutex_unlock(j->mu);
        if (qi >= j->nq) break;
        const float *q = j->queries + qi * j->dim;
        if (j->full_sort)
            so_brute_force_search(j->rows, j->n, j->dim, j->deleted, q, j->k, SO_METRIC_NDP, j->order,
                                  j->out_ids + qi * j->k, j->out_dist + qi * j->k);
        else
            so_brute_force_search_select(j->rows, j->n, j->dim, j->deleted, q, j->k, SO_METRIC_NDP, j->order,
                                         j->out_ids + qi * j->k, j->out_dist + qi * j->k);
    }
    return NULL;
}

/* `deleted` (may be NULL): one byte per row, non-zero = tombstoned (vamana.rs:1175-1177 skips them) */
double so_bench_brute_force_del(const float *rows, size_t n, size_t dim, const uint8_t *deleted, const float *queries,
                                size_t nq, size_t k, int order, int full_sort, int threads,
                                uint32_t *out_ids, float *out_dist) {
    if (threads < 1) threads = 1;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    size_t next = 0;
    bf_job_t job = {rows, n, dim, queries, nq, k, order, full_sort, out_ids, out_dist, &next, &mu, deleted};
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, bf_worker, &job);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

double so_bench_brute_force(const float *rows, size_t n, size_t dim, const float *queries,
                            size_t nq, size_t k, int order, int full_sort, int threads,
                            uint32_t *out_ids, float *out_dist) {
    return so_bench_brute_force_del(rows, n, dim, NULL, queries, nq, k, order, full_sort, threads, out_ids, out_dist);
}

/* A copy of `src` whose pages are first touched by `threads` worker threads, page p by thread p % threads: on a multi-socket
 * host the copy ends up spread over the NUMA nodes the workers run on instead of sitting on the node of the thread that
 * produced `src` (bench.py's multi-threaded CPU baseline streams the corpus from every core). Free with so_free. */
typedef struct { const char *src; char *dst; size_t bytes; int t, nt; } ic_job_t;
static void *ic_worker(void *arg) {
    ic_job_t *j = (ic_job_t *)arg;
    const size_t page = 1u << 21;          /* 2 MiB: one THP / a few hundred 4K pages per stripe */
    for (size_t off = (size_t)j->t * page; off < j->bytes; off += (size_t)j->nt * page) {
        size_t len = j->bytes - off < page ? j->bytes - off : page;
        memcpy(j->dst + off, j->src + off, len);
    }
    return NULL;
}
float *so_interleaved_copy(const float *src, size_t n_floats, int threads) {
    if (threads < 1) threads = 1;
    void *dst = NULL;
    if (posix_memalign(&dst, 1u << 21, n_floats * 4 + 64) != 0) return NULL;
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    ic_job_t *jobs = (ic_job_t *)malloc((size_t)threads * sizeof(ic_job_t));
    for (int t = 0; t < threads; ++t) {
        jobs[t].src = (const char *)src; jobs[t].dst = (char *)dst; jobs[t].bytes = n_floats * 4; jobs[t].t = t; jobs[t].nt = threads;
        pthread_create(&th[t], NULL, ic_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return (float *)dst;
}
void so_free(void *p) { free(p); }

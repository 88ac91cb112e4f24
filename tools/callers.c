/* callers.c -- bench / test driver for the reference's real call pattern: T host threads, each calling the C ABI with ONE item per call in a closed
 * loop on ONE handle (recall.rs:512-513 spawn_blocking + one search; minilm.rs:889-897 one encode behind the session mutex). Built with gcc by
 * bench.py / the tests at run time; the ABI entry points arrive as function pointers (ctypes addresses), so this file links against nothing.
 * Every result is compared byte for byte with the expected (solo) answer of its item when one is given. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*search_fn)(void *idx, const float *q, uint32_t nq, uint32_t k, uint32_t *ids, float *dist, uint32_t *counts);
typedef int (*encode_fn)(void *emb, const int32_t *ids, const uint8_t *mask, uint32_t b, float *out);

typedef struct {
    double wall_s, p50_us, p99_us, mean_us, max_us;
    uint64_t calls, mismatches, errors;
} callers_result;

typedef struct {
    int kind;                 /* 0 search, 1 encode */
    void *fn, *handle;
    const float *queries; uint32_t n_items, dim, k;
    const uint32_t *expect_ids; const float *expect_dist;          /* [n_items][k] or NULL */
    const int32_t *tok_ids; const uint8_t *tok_mask; uint32_t max_len, hidden;
    const float *expect_vec;                                        /* [n_items][hidden] or NULL */
    uint32_t threads, calls, warmup, tid;
    pthread_barrier_t *bar;
    double *lat_us;           /* [calls] of this thread */
    uint64_t mismatches, errors;
    double t_start, t_end;
} worker;

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3;
}

static void one_call(worker *w, uint32_t item, uint32_t *ids, float *dist, float *vec, int check) {
    if (w->kind == 0) {
        uint32_t cnt = 0;
        const int rc = ((search_fn)w->fn)(w->handle, w->queries + (size_t)item * w->dim, 1, w->k, ids, dist, &cnt);
        if (rc != 0) { w->errors++; return; }
        if (check && w->expect_ids &&
            (memcmp(ids, w->expect_ids + (size_t)item * w->k, (size_t)w->k * 4) != 0 || memcmp(dist, w->expect_dist + (size_t)item * w->k, (size_t)w->k * 4) != 0)) w->mismatches++;
    } else {
        const int rc = ((encode_fn)w->fn)(w->handle, w->tok_ids + (size_t)item * w->max_len, w->tok_mask + (size_t)item * w->max_len, 1, vec);
        if (rc != 0) { w->errors++; return; }
        if (check && w->expect_vec && memcmp(vec, w->expect_vec + (size_t)item * w->hidden, (size_t)w->hidden * 4) != 0) w->mismatches++;
    }
}

static void *run(void *arg) {
    worker *w = (worker *)arg;
    uint32_t *ids = (uint32_t *)malloc((size_t)(w->k ? w->k : 1) * 4);
    float *dist = (float *)malloc((size_t)(w->k ? w->k : 1) * 4);
    float *vec = (float *)malloc((size_t)(w->hidden ? w->hidden : 1) * 4);
    uint32_t item = (w->tid * 7919u) % w->n_items;
    for (uint32_t i = 0; i < w->warmup; ++i) { one_call(w, item, ids, dist, vec, 1); item = (item + w->threads) % w->n_items; }
    pthread_barrier_wait(w->bar);
    w->t_start = now_us();
    for (uint32_t i = 0; i < w->calls; ++i) {
        const double t0 = now_us();
        one_call(w, item, ids, dist, vec, 1);
        w->lat_us[i] = now_us() - t0;
        item = (item + w->threads) % w->n_items;
    }
    w->t_end = now_us();
    free(ids); free(dist); free(vec);
    return 0;
}

static int cmp_d(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

static int drive(worker proto, uint32_t threads, uint32_t calls, uint32_t warmup, callers_result *out) {
    if (!out || threads == 0 || calls == 0 || proto.n_items == 0) return -1;
    worker *w = (worker *)calloc(threads, sizeof(worker));
    pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
    double *lat = (double *)malloc((size_t)threads * calls * sizeof(double));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, 0, threads);
    for (uint32_t t = 0; t < threads; ++t) {
        w[t] = proto; w[t].threads = threads; w[t].calls = calls; w[t].warmup = warmup; w[t].tid = t; w[t].bar = &bar; w[t].lat_us = lat + (size_t)t * calls;
        pthread_create(&th[t], 0, run, &w[t]);
    }
    double t0 = 1e300, t1 = 0;
    memset(out, 0, sizeof(*out));
    for (uint32_t t = 0; t < threads; ++t) {
        pthread_join(th[t], 0);
        if (w[t].t_start < t0) t0 = w[t].t_start;
        if (w[t].t_end > t1) t1 = w[t].t_end;
        out->mismatches += w[t].mismatches; out->errors += w[t].errors;
    }
    const size_t n = (size_t)threads * calls;
    qsort(lat, n, sizeof(double), cmp_d);
    double sum = 0;
    for (size_t i = 0; i < n; ++i) sum += lat[i];
    out->wall_s = (t1 - t0) * 1e-6; out->calls = n; out->p50_us = lat[n / 2]; out->p99_us = lat[(size_t)((double)n * 0.99)]; out->mean_us = sum / (double)n; out->max_us = lat[n - 1];
    pthread_barrier_destroy(&bar);
    free(w); free(th); free(lat);
    return 0;
}

int callers_search(void *fn, void *idx, const float *queries, uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t threads, uint32_t calls_per_thread,
                   uint32_t warmup_per_thread, const uint32_t *expect_ids, const float *expect_dist, callers_result *out) {
    worker p;
    memset(&p, 0, sizeof(p));
    p.kind = 0; p.fn = fn; p.handle = idx; p.queries = queries; p.n_items = n_queries; p.dim = dim; p.k = k; p.expect_ids = expect_ids; p.expect_dist = expect_dist;
    return drive(p, threads, calls_per_thread, warmup_per_thread, out);
}

int callers_encode(void *fn, void *emb, const int32_t *ids, const uint8_t *mask, uint32_t n_texts, uint32_t max_len, uint32_t hidden, uint32_t threads,
                   uint32_t calls_per_thread, uint32_t warmup_per_thread, const float *expect_vec, callers_result *out) {
    worker p;
    memset(&p, 0, sizeof(p));
    p.kind = 1; p.fn = fn; p.handle = emb; p.tok_ids = ids; p.tok_mask = mask; p.n_items = n_texts; p.max_len = max_len; p.hidden = hidden; p.expect_vec = expect_vec;
    return drive(p, threads, calls_per_thread, warmup_per_thread, out);
}

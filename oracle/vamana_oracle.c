/* vamana_oracle.c -- CPU restatement of the Vamana GRAPH side of VamanaIndex (TEST INFRASTRUCTURE ONLY, like shodh_oracle.c; it is
 * compiled into the same libshodh_oracle.so):
 *   greedy_search            src/vector_db/vamana.rs:576-657
 *   search (ANN path)        :764-808   (over-fetch for tombstones, filter, take k)
 *   add_vector               :853-974   (incremental insert: k-nearest by greedy_search, back edges pruned by distance)
 *   robust_prune             :665-746   (alpha-RNG on the (1 + d) scale for NormalizedDotProduct)
 *   find_medoid              :407-441
 *   build                    :200-284   GIVEN the initial random graph (the reference draws it from thread_rng, :287-312)
 * Note: an index that only ever grows through add_vector (what `remember` does) has a fully deterministic graph -- no RNG on that
 * path, medoid 0 -- so the reference's default (non-exact) `search` is reproducible for it, bit for bit.
 * SearchCandidate's order is total: (distance total_cmp, id) (:1664-1673). Keys are unique (ids are), so BinaryHeap's internal
 * order never shows: any priority queue over the same keys pops the same sequence. Distances: -dot in the configured order. */
#include "shodh_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t *a; size_t n, cap; } heap_t;

static uint64_t vkey(float d, uint32_t id) { return ((uint64_t)so_total_order_key(d) << 32) | (uint64_t)id; }
static float key_dist(uint64_t k) {
    uint32_t b = (uint32_t)(k >> 32);
    b = (b & 0x80000000u) ? (b ^ 0x80000000u) : ~b;
    float f; memcpy(&f, &b, 4); return f;
}
static void heap_push(heap_t *h, uint64_t k, int is_max) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->a = (uint64_t *)realloc(h->a, h->cap * 8); }
    size_t i = h->n++;
    h->a[i] = k;
    while (i) {
        size_t p = (i - 1) / 2;
        int up = is_max ? (h->a[i] > h->a[p]) : (h->a[i] < h->a[p]);
        if (!up) break;
        uint64_t t = h->a[i]; h->a[i] = h->a[p]; h->a[p] = t; i = p;
    }
}
static uint64_t heap_pop(heap_t *h, int is_max) {
    uint64_t top = h->a[0];
    h->a[0] = h->a[--h->n];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->n && (is_max ? h->a[l] > h->a[m] : h->a[l] < h->a[m])) m = l;
        if (r < h->n && (is_max ? h->a[r] > h->a[m] : h->a[r] < h->a[m])) m = r;
        if (m == i) break;
        uint64_t t = h->a[i]; h->a[i] = h->a[m]; h->a[m] = t; i = m;
    }
    return top;
}

static float vdist(const float *a, const float *b, size_t dim, int order) { return so_normalized_distance(a, b, dim, order); }

/* greedy_search (vamana.rs:576-657). Graph: deg[n], nbr[n][cap]. Returns the number of results (<= k), ascending (distance, id). */
size_t so_vamana_greedy_search(const float *vecs, size_t n, size_t dim, const uint32_t *deg, const uint32_t *nbr, size_t cap,
                               const float *q, size_t k, uint32_t entry, int order, uint32_t *out_ids, float *out_dist) {
    if (n == 0 || entry >= n) return 0;
    uint8_t *visited = (uint8_t *)calloc(n, 1);
    heap_t cand = {0, 0, 0}, w = {0, 0, 0};
    const float ed = vdist(q, vecs + (size_t)entry * dim, dim, order);
    heap_push(&cand, vkey(ed, entry), 0);
    heap_push(&w, vkey(ed, entry), 1);
    visited[entry] = 1;
    while (cand.n) {
        const uint64_t cur = heap_pop(&cand, 0);
        if (w.n && key_dist(cur) > key_dist(w.a[0])) break;                  /* current.distance > p.distance */
        const uint32_t cid = (uint32_t)cur;
        if (cid >= n) continue;
        for (uint32_t j = 0; j < deg[cid]; ++j) {
            const uint32_t nb = nbr[(size_t)cid * cap + j];
            if (nb >= n) continue;                                            /* (get_slice_from_storage would fail; cannot happen for a consistent graph) */
            if (visited[nb]) continue;
            visited[nb] = 1;
            const float d = vdist(q, vecs + (size_t)nb * dim, dim, order);
            const int should_add = w.n < k || (w.n && d < key_dist(w.a[0]));
            if (should_add) {
                heap_push(&cand, vkey(d, nb), 0);
                heap_push(&w, vkey(d, nb), 1);
                if (w.n > k) heap_pop(&w, 1);
            }
        }
    }
    size_t m = w.n;
    for (size_t i = m; i-- > 0;) { const uint64_t t = heap_pop(&w, 1); out_ids[i] = (uint32_t)t; out_dist[i] = key_dist(t); }
    free(visited); free(cand.a); free(w.a);
    return m;
}

/* VamanaIndex::search without SHODH_VECTOR_EXACT (vamana.rs:764-808) */
size_t so_vamana_search(const float *vecs, size_t n, size_t dim, const uint32_t *deg, const uint32_t *nbr, size_t cap, uint32_t medoid,
                        const uint8_t *deleted, const float *q, size_t k, int order, uint32_t *out_ids, float *out_dist) {
    if (n == 0) return 0;
    size_t dc = 0;
    if (deleted) for (size_t i = 0; i < n; ++i) dc += deleted[i] != 0;
    const size_t search_k = dc > 0 ? k + (dc < 2 * k ? dc : 2 * k) : k;
    uint32_t *ids = (uint32_t *)malloc((search_k ? search_k : 1) * 4);
    float *dist = (float *)malloc((search_k ? search_k : 1) * 4);
    const size_t m = so_vamana_greedy_search(vecs, n, dim, deg, nbr, cap, q, search_k, medoid, order, ids, dist);
    size_t o = 0;
    for (size_t i = 0; i < m && o < k; ++i)
        if (!(deleted && deleted[ids[i]])) { out_ids[o] = ids[i]; out_dist[o] = dist[i]; ++o; }
    free(ids); free(dist);
    return o;
}

typedef struct { uint32_t id; float d; } idd2_t;
static int idd2_cmp(const void *pa, const void *pb) {
    const idd2_t *a = (const idd2_t *)pa, *b = (const idd2_t *)pb;
    const int c = so_total_cmp(a->d, b->d);
    if (c) return c;
    return a->id < b->id ? -1 : (a->id > b->id ? 1 : 0);
}

/* add_vector (vamana.rs:853-974): vecs already holds row `id` = n_before. deg/nbr have room for n_before + 1 nodes. */
void so_vamana_add_vector(const float *vecs, size_t n_before, size_t dim, uint32_t *deg, uint32_t *nbr, size_t cap, size_t R,
                          uint32_t medoid, int order) {
    const uint32_t id = (uint32_t)n_before;
    deg[id] = 0;
    if (n_before == 0) return;                                    /* the first vector: a node without neighbours, medoid 0 */
    uint32_t *ids = (uint32_t *)malloc((R ? R : 1) * 4);
    float *dist = (float *)malloc((R ? R : 1) * 4);
    const size_t m = so_vamana_greedy_search(vecs, n_before, dim, deg, nbr, cap, vecs + (size_t)id * dim, R, medoid, order, ids, dist);
    const size_t take = m < R ? m : R;
    for (size_t i = 0; i < take; ++i) nbr[(size_t)id * cap + i] = ids[i];
    deg[id] = (uint32_t)take;
    for (size_t i = 0; i < take; ++i) {
        const uint32_t nb = ids[i];
        nbr[(size_t)nb * cap + deg[nb]++] = id;                    /* cap >= R + 1 */
        if (deg[nb] > R) {
            idd2_t *nd = (idd2_t *)malloc(deg[nb] * sizeof(idd2_t));
            for (uint32_t j = 0; j < deg[nb]; ++j) {
                const uint32_t x = nbr[(size_t)nb * cap + j];
                nd[j].id = x; nd[j].d = vdist(vecs + (size_t)nb * dim, vecs + (size_t)x * dim, dim, order);
            }
            qsort(nd, deg[nb], sizeof(idd2_t), idd2_cmp);          /* (distance total_cmp, id): a total order, so any sort agrees with sort_by */
            for (size_t j = 0; j < R; ++j) nbr[(size_t)nb * cap + j] = nd[j].id;
            deg[nb] = (uint32_t)R;
            free(nd);
        }
    }
    free(ids); free(dist);
}

/* robust_prune (vamana.rs:665-746). cand: n_c (id, distance) pairs. Returns the pruned neighbour count (<= R) in out. */
size_t so_vamana_robust_prune(const float *vecs, size_t dim, uint32_t node, const uint32_t *c_ids, const float *c_dist, size_t n_c,
                              size_t R, float alpha, int order, uint32_t *out) {
    if (n_c == 0) return 0;
    idd2_t *sc = (idd2_t *)malloc(n_c * sizeof(idd2_t));
    for (size_t i = 0; i < n_c; ++i) { sc[i].id = c_ids[i]; sc[i].d = c_dist[i]; }
    qsort(sc, n_c, sizeof(idd2_t), idd2_cmp);
    float *dist_ne = (float *)malloc((R ? R : 1) * 4);
    size_t np = 0;
    const float off = 1.0f;                                       /* NormalizedDotProduct: compare on the nonnegative (1 + d) scale */
    const float *nv = vecs + (size_t)node * dim;
    for (size_t i = 0; i < n_c && np < R; ++i) {
        if (sc[i].id == node) continue;
        const float *cv = vecs + (size_t)sc[i].id * dim;
        const float dist_nc = vdist(nv, cv, dim, order);
        int add = 1;
        for (size_t j = 0; j < np; ++j) {
            const float dist_ce = vdist(cv, vecs + (size_t)out[j] * dim, dim, order);
            if (alpha * (dist_ce + off) <= (dist_nc + off) && dist_ce <= dist_ne[j]) { add = 0; break; }
        }
        if (add) { out[np] = sc[i].id; dist_ne[np] = dist_nc; ++np; }
    }
    free(sc); free(dist_ne);
    return np;
}

/* find_medoid (vamana.rs:407-441): the row closest (strict '<', first wins) to the mean vector */
uint32_t so_vamana_find_medoid(const float *vecs, size_t n, size_t dim, int order) {
    if (n == 0) return 0;
    float *c = (float *)calloc(dim, 4);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < dim; ++j) c[j] = c[j] + vecs[i * dim + j];
    for (size_t j = 0; j < dim; ++j) c[j] = c[j] / (float)n;
    uint32_t best = 0;
    float bd = 3.4028235e38f;
    for (size_t i = 0; i < n; ++i) {
        const float d = vdist(vecs + i * dim, c, dim, order);
        if (d < bd) { bd = d; best = (uint32_t)i; }
    }
    free(c);
    return best;
}

/* build (vamana.rs:200-284) GIVEN the initial graph in deg/nbr. Returns the medoid; deg/nbr hold the built graph. */
uint32_t so_vamana_build(const float *vecs, size_t n, size_t dim, uint32_t *deg, uint32_t *nbr, size_t cap, size_t R, size_t L, float alpha,
                         int order) {
    if (n == 0) return 0;
    const uint32_t medoid = so_vamana_find_medoid(vecs, n, dim, order);
    uint32_t *ids = (uint32_t *)malloc((L ? L : 1) * 4), *pr = (uint32_t *)malloc((R + 1) * 4), *pr2 = (uint32_t *)malloc((R + 1) * 4);
    float *dist = (float *)malloc((L ? L : 1) * 4), *zero = (float *)calloc(cap + 1, 4);
    for (int iteration = 1;; ++iteration) {
        size_t updates = 0;
        for (size_t node = 0; node < n; ++node) {
            const size_t m = so_vamana_greedy_search(vecs, n, dim, deg, nbr, cap, vecs + node * dim, L, medoid, order, ids, dist);
            const size_t np = so_vamana_robust_prune(vecs, dim, (uint32_t)node, ids, dist, m, R, alpha, order, pr);
            int same = (np == deg[node]);
            for (size_t j = 0; same && j < np; ++j) same = nbr[node * cap + j] == pr[j];
            if (same) continue;
            ++updates;
            for (size_t j = 0; j < np; ++j) nbr[node * cap + j] = pr[j];
            deg[node] = (uint32_t)np;
            for (size_t j = 0; j < np; ++j) {
                const uint32_t nb = pr[j];
                if (nb >= n) continue;
                int has = 0;
                for (uint32_t t = 0; t < deg[nb]; ++t) has |= nbr[(size_t)nb * cap + t] == (uint32_t)node;
                if (has) continue;
                nbr[(size_t)nb * cap + deg[nb]++] = (uint32_t)node;
                if (deg[nb] > R) {
                    const size_t q2 = so_vamana_robust_prune(vecs, dim, nb, nbr + (size_t)nb * cap, zero, deg[nb], R, alpha, order, pr2);
                    for (size_t t = 0; t < q2; ++t) nbr[(size_t)nb * cap + t] = pr2[t];
                    deg[nb] = (uint32_t)q2;
                }
            }
        }
        if (updates == 0 || iteration >= 2) break;
    }
    free(ids); free(pr); free(pr2); free(dist); free(zero);
    return medoid;
}

/* incremental_repair (vamana.rs:1033-1115) for nodes [start, n): greedy_search(L) from the medoid, robust_prune, and when the list
 * changed: stale back edges removed (retain), new back edges pushed and truncated to R. Returns the number of nodes whose list changed. */
size_t so_vamana_incremental_repair(const float *vecs, size_t n, size_t dim, uint32_t *deg, uint32_t *nbr, size_t cap, size_t R, size_t L,
                                    float alpha, uint32_t medoid, uint32_t start, int order) {
    if (n == 0) return 0;
    uint32_t *ids = (uint32_t *)malloc((L ? L : 1) * 4), *pr = (uint32_t *)malloc((R + 1) * 4), *old = (uint32_t *)malloc((cap + 1) * 4);
    float *dist = (float *)malloc((L ? L : 1) * 4);
    size_t repaired = 0;
    for (uint32_t node = start; node < (uint32_t)n; ++node) {
        const size_t m = so_vamana_greedy_search(vecs, n, dim, deg, nbr, cap, vecs + (size_t)node * dim, L, medoid, order, ids, dist);
        const size_t np = so_vamana_robust_prune(vecs, dim, node, ids, dist, m, R, alpha, order, pr);
        const uint32_t od = deg[node];
        int same = (np == od);
        for (size_t j = 0; same && j < np; ++j) same = nbr[(size_t)node * cap + j] == pr[j];
        if (same) continue;
        for (uint32_t j = 0; j < od; ++j) old[j] = nbr[(size_t)node * cap + j];
        for (size_t j = 0; j < np; ++j) nbr[(size_t)node * cap + j] = pr[j];
        deg[node] = (uint32_t)np;
        ++repaired;
        for (uint32_t j = 0; j < od; ++j) {                       /* back edges of neighbours that are gone */
            const uint32_t o = old[j];
            int kept = 0;
            for (size_t t = 0; t < np; ++t) kept |= pr[t] == o;
            if (kept || o >= n) continue;
            uint32_t w = 0;
            for (uint32_t t = 0; t < deg[o]; ++t) { const uint32_t x = nbr[(size_t)o * cap + t]; if (x != node) nbr[(size_t)o * cap + w++] = x; }
            deg[o] = w;
        }
        for (size_t j = 0; j < np; ++j) {                         /* back edges to the new neighbours: push, truncate to R */
            const uint32_t p = pr[j];
            if (p >= n) continue;
            int has = 0;
            for (uint32_t t = 0; t < deg[p]; ++t) has |= nbr[(size_t)p * cap + t] == node;
            if (has) continue;
            if (deg[p] < cap) nbr[(size_t)p * cap + deg[p]] = node;
            deg[p] += 1;
            if (deg[p] > R) deg[p] = (uint32_t)R;
        }
    }
    free(ids); free(pr); free(old); free(dist);
    return repaired;
}

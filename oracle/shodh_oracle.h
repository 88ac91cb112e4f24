/*
 * shodh_oracle.h -- CPU restatement of the shodh-memory embed-and-recall hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (shodh_memory_amd/, the C-ABI library,
 * bench.py's GPU legs) may include, link, import or execute anything under oracle/.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and only
 * as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line (under /root/reference/) whose arithmetic
 * ORDER it restates.  rustc never contracts a*b+c into an FMA and LLVM may not
 * re-associate float sums, so the written order in the reference IS the executed order;
 * this file must therefore be compiled with -ffp-contract=off (see oracle/Makefile).
 *
 * Parity pinning: the known-answer tests of the reference's own inline unit tests
 * (distance_inline.rs:514-635, similarity.rs:54-123, vamana.rs:1686-1712,
 * retrieval.rs:2430-2529, spann.rs:1121-1237, pq.rs:496-577, relevance.rs:1794-2066,
 * hybrid_search.rs:969-1056, minilm.rs:1399-1439) are restated in tests/test_oracle_kats.py.
 * The MiniLM encoder itself (ONNX Runtime 1.23.2 + HF weights, not in the reference tree)
 * is NOT restated here: encoder parity is unpinned, see DESIGN.md.
 */
#ifndef SHODH_ORACLE_H
#define SHODH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* accumulation orders of dot_product_inline (distance_inline.rs:28-61) */
enum { SO_ORDER_SCALAR4 = 0, SO_ORDER_AVX2 = 1 };
/* DistanceMetric (vamana.rs DistanceMetric; vamana_persist.rs header byte) */
enum { SO_METRIC_NDP = 0, SO_METRIC_EUCLIDEAN = 1, SO_METRIC_COSINE = 2 };

/* ---- distance_inline.rs ---------------------------------------------------------- */
float so_dot_scalar4(const float *a, const float *b, size_t n);   /* :157-173 */
float so_dot_avx2(const float *a, const float *b, size_t n);      /* :67-111 (fmaf emulation) */
float so_dot_avx2_native(const float *a, const float *b, size_t n); /* same, AVX2 intrinsics; falls back to emulation */
float so_dot(const float *a, const float *b, size_t n, int order);
float so_l2sq_scalar4(const float *a, const float *b, size_t n);  /* :283-305 */
float so_l2sq_avx2(const float *a, const float *b, size_t n);     /* :219-253 */
float so_l2sq(const float *a, const float *b, size_t n, int order);
float so_normsq_scalar4(const float *a, size_t n);                /* :413-431 */
float so_normsq_avx2(const float *a, size_t n);                   /* :355-385 */
float so_normsq(const float *a, size_t n, int order);
float so_l2_norm(const float *a, size_t n, int order);            /* :315-317 */
float so_cosine_similarity_inline(const float *a, const float *b, size_t n, int order); /* :444-458 */
float so_cosine_distance_inline(const float *a, const float *b, size_t n, int order);   /* :464-466 */
float so_normalized_distance(const float *a, const float *b, size_t n, int order);      /* :479-481 */
int   so_is_normalized(const float *a, size_t n, float eps, int order);                 /* :487-490 */
void  so_normalize_inplace(float *a, size_t n, int order);                              /* :494-502 */
float so_metric_distance(const float *a, const float *b, size_t n, int metric, int order); /* vamana.rs:755-761 */

/* ---- similarity.rs ---------------------------------------------------------------- */
float so_cosine_similarity(const float *a, size_t na, const float *b, size_t nb, int order); /* :10-24 */
/* top_k_similar (:27-48): candidates [n][dim]; writes min(k,n) (score, index) pairs, stable
 * descending on OrderedFloat(score). Returns the count. */
size_t so_top_k_similar(const float *query, const float *cands, size_t n, size_t dim, size_t k,
                        int order, float *out_scores, uint32_t *out_index);

/* ---- f32::total_cmp ---------------------------------------------------------------- */
int      so_total_cmp(float a, float b);       /* -1/0/+1 */
uint32_t so_total_order_key(float x);          /* unsigned key, ascending == total_cmp ascending */

/* ---- vamana.rs --------------------------------------------------------------------- */
/* brute_force_search (:1167-1188): rows [n][dim] row-major; deleted = NULL or n bytes
 * (non-zero = tombstoned). Output ascending (dist total_cmp, id). Returns count <= k.
 * This variant does what the reference does: score every live row, full sort, truncate. */
size_t so_brute_force_search(const float *rows, size_t n, size_t dim, const uint8_t *deleted,
                             const float *q, size_t k, int metric, int order,
                             uint32_t *out_ids, float *out_dist);
/* Same result, bounded selection instead of the full sort (fair CPU baseline). */
size_t so_brute_force_search_select(const float *rows, size_t n, size_t dim, const uint8_t *deleted,
                                    const float *q, size_t k, int metric, int order,
                                    uint32_t *out_ids, float *out_dist);

/* ---- retrieval.rs ------------------------------------------------------------------ */
/* search_ids post-processing (:920-963). vec_ids/dists: index.search() output (n_res).
 * vector_to_memory: [n_vectors][16] uuid bytes, an all-0xFF entry means "no mapping".
 * Output: (uuid, similarity) sorted (sim total_cmp desc, uuid bytes asc), truncated to limit. */
size_t so_search_ids_postprocess(const uint32_t *vec_ids, const float *dists, size_t n_res,
                                 const uint8_t *vector_to_memory, size_t n_vectors, size_t limit,
                                 uint8_t *out_uuid /*[limit][16]*/, float *out_sim);

/* ---- pq.rs ------------------------------------------------------------------------- */
/* codebook layout [M][ncent][sub] f32 (sub = 8, M = dim/8, ncent <= 256) */
float so_squared_l2(const float *a, const float *b, size_t n);                      /* :393-395 */
void  so_pq_encode(const float *codebook, size_t M, size_t ncent, size_t sub,
                   const float *v, uint8_t *codes);                                 /* :220-257 */
void  so_pq_decode(const float *codebook, size_t M, size_t ncent, size_t sub,
                   const uint8_t *codes, float *out);                               /* :259-291 */
void  so_pq_build_distance_table(const float *codebook, size_t M, size_t ncent, size_t sub,
                                 const float *q, float *table /*[M][ncent]*/);      /* :329-351 */
float so_pq_distance_with_table(const float *table, size_t M, size_t ncent,
                                const uint8_t *codes, size_t ncodes);               /* :358-368 */
float so_pq_asymmetric_distance(const float *codebook, size_t M, size_t ncent, size_t sub,
                                const float *q, const uint8_t *codes);              /* :297-324 */
/* k-means of pq.rs:152-217 with the shuffled index order supplied (thread_rng is not
 * reproducible): init_perm is a permutation of 0..n-1. vectors [n][dim]. out [k][dim]. */
void  so_pq_kmeans(const float *vectors, size_t n, size_t dim, size_t k, size_t iterations,
                   const uint32_t *init_perm, float *out_centroids);

/* ---- spann.rs ---------------------------------------------------------------------- */
float  so_spann_compute_distance(const float *a, const float *b, size_t n, int metric); /* :562-571 */
size_t so_spann_find_nearest_centroid(const float *v, const float *centroids, size_t P,
                                      size_t dim, int metric);                          /* :545-558 */
size_t so_spann_compute_partitions(size_t num_vectors);                                 /* :136-139 */
/* kmeans_cluster (:466-541) with supplied init permutation; returns iterations run */
size_t so_spann_kmeans(const float *vectors, size_t n, size_t dim, size_t k, size_t iterations,
                       int metric, const uint32_t *init_perm, float *out_centroids);
/* search (:574-693). Postings in CSR form: list_off[P+1], ids[total], codes[total][M]. */
size_t so_spann_search(const float *centroids, size_t P, size_t dim, int metric,
                       const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes,
                       const float *codebook, size_t M, size_t ncent, size_t sub,
                       size_t num_probes, const float *q, size_t k,
                       uint32_t *out_ids, float *out_dist);

/* ---- minilm.rs --------------------------------------------------------------------- */
/* masked mean-pool (:959-978) + finalize_pooled (:846-878, native-384 branch).
 * hidden [S][H] f32, mask [S] (==1 counts). out [H]. */
void so_mean_pool_finalize(const float *hidden, const int64_t *mask, size_t S, size_t H, float *out);
/* generate_embedding_simplified (:777-831): SipHash-1-3 (k0=k1=0) over words + char bigrams */
uint64_t so_siphash13_str(const uint8_t *bytes, size_t len);   /* Hash for str: bytes || 0xFF */
void so_hash_embed(const char *utf8, size_t len, size_t dim, float *out);

/* ---- relevance.rs ------------------------------------------------------------------ */
typedef struct {
    float semantic, entity, tag, importance, momentum, access_count, graph_strength;
    uint32_t update_count;
} so_weights;
void  so_weights_default(so_weights *w);                                   /* :64-100, :383-397 */
void  so_weights_normalize(so_weights *w);                                 /* :401-418 */
void  so_weights_apply_feedback(so_weights *w, int sem, int ent, int tag, int helpful); /* :427-465 */
float so_calibrate_score(float s);                                         /* :601-606 */
float so_fuse_scores_full(const so_weights *w, float sem, float ent, float tag, float imp,
                          float momentum_ema, uint32_t access_count, float graph_strength); /* :529-594 */
float so_fuse_scores(const so_weights *w, float sem, float ent, float tag, float imp);     /* :471-487 */
float so_fuse_scores_with_momentum(const so_weights *w, float sem, float ent, float tag,
                                   float imp, float mom);                                   /* :499-517 */
float so_calculate_tag_score(const char *content, const char *const *tags, size_t n_tags);  /* :680-705 */
float so_apply_recency_boost(float base, int64_t age_hours, uint64_t boost_hours, float mult); /* :1524-1547 */

/* ranking tail of RelevanceEngine::surface_relevant_inner (:801-918): min-importance filter, fuse_scores_full, recency boost,
 * graph boost (entity matches), sort (score desc, created_at desc, id asc), score >= 0.25, truncate. Outputs hold max_results
 * entries: index of the candidate, final score, reason (0 Combined, 1 EntityMatch, 2 SemanticSimilarity, 3 RecentImportant). */
size_t so_rank_surfaced(const so_weights *w, float min_importance, uint64_t recency_boost_hours, float recency_boost_multiplier,
                        float graph_boost_multiplier, size_t max_results, size_t n, const float *semantic, const float *entity,
                        const float *tag, const float *importance, const float *momentum_ema, const uint32_t *access_count,
                        const float *graph_strength, const int64_t *age_hours, const int64_t *created_at_ns, const uint8_t *uuid,
                        uint32_t *out_index, float *out_score, uint8_t *out_reason);

/* ---- hybrid_search.rs -------------------------------------------------------------- */
/* RRFusion::new + fuse (:536-594). lists: n_lists ranked lists of uuids (16 B each),
 * list_len[l] entries each, concatenated in `uuids`. Output sorted (score desc, uuid asc). */
size_t so_rrf_fuse(float k, const float *weights, size_t n_lists, const uint8_t *uuids,
                   const size_t *list_len, uint8_t *out_uuid, float *out_score, size_t out_cap);

/* ---- memory/mod.rs recall Layer 4: hybrid leg + graph leg fusion (:3878-4468) ----------
 * The reference has no unit test with literal values for this block (it lives inside MemorySystem::recall), so this
 * restatement is cross-checked against an independent numpy float32 restatement in tests/test_legfusion_cpu.py and
 * against the density-weight tests graph_retrieval.rs:2447-2472: "parity unpinned" beyond those.
 * Switches (environment variables in the reference) arrive as an int/float parameter block:
 *   sw[0] SHODH_FUSION_V2, sw[1] SHODH_FUSION_FLAT, sw[2] SHODH_FUSION_SUM, sw[3] SHODH_FUSION_RRF,
 *   sw[4] SHODH_LEG (0 unset, 1 vector, 2 bm25, 3 graph), sw[5] SHODH_FLAT_ADAPTIVE, sw[6] SHODH_ADAPT_FEATURE
 *   (0 fitted, 1 agreement, 2 peak), sw[7] SHODH_ADAPT_SYMMETRIC
 *   fp[0] graph_w, fp[1] hybrid_w, fp[2] k, fp[3] flat_consensus, fp[4] adapt_trust_max, fp[5..7] fw graph/vec/bm25,
 *   fp[8..10] agree k/lo/hi, fp[11..12] peak lo/hi */
void so_density_weights(float graph_density, float out3[3]);                 /* graph_retrieval.rs:81-101 */
void so_leg_fusion_weights(int has_density, float graph_density, float graph_weight_override, float graph_w_floor,
                           float *graph_w, float *hybrid_w);                 /* mod.rs:3878-3921; NaN = variable unset */
size_t so_fuse_legs(const int sw[8], const float fp[13], const uint8_t *hybrid_uuid, const float *hybrid_bm25,
                    const float *hybrid_vec, size_t n_hybrid, const uint8_t *graph_uuid, const float *graph_activation,
                    size_t n_graph, size_t query_len, uint8_t *out_uuid, float *out_score, size_t out_cap, float *vec_trust_out);

/* ---- vamana_persist.rs / spann.rs checksums ---------------------------------------- */
uint64_t so_fnv1a64(const uint8_t *data, size_t len);          /* vamana_persist.rs:155-163 */

/* ---- the Vamana GRAPH (vamana_oracle.c): greedy_search :576-657, search (ANN) :764-808, add_vector :853-974, robust_prune :665-746,
 * find_medoid :407-441, build :200-284 given the initial graph. Graph = deg[n] + nbr[n][cap] (cap >= R + 1). -------------------- */
size_t so_vamana_greedy_search(const float *vecs, size_t n, size_t dim, const uint32_t *deg, const uint32_t *nbr, size_t cap,
                               const float *q, size_t k, uint32_t entry, int order, uint32_t *out_ids, float *out_dist);
size_t so_vamana_search(const float *vecs, size_t n, size_t dim, const uint32_t *deg, const uint32_t *nbr, size_t cap, uint32_t medoid,
                        const uint8_t *deleted, const float *q, size_t k, int order, uint32_t *out_ids, float *out_dist);
void so_vamana_add_vector(const float *vecs, size_t n_before, size_t dim, uint32_t *deg, uint32_t *nbr, size_t cap, size_t R,
                          uint32_t medoid, int order);
size_t so_vamana_robust_prune(const float *vecs, size_t dim, uint32_t node, const uint32_t *c_ids, const float *c_dist, size_t n_c,
                              size_t R, float alpha, int order, uint32_t *out);
uint32_t so_vamana_find_medoid(const float *vecs, size_t n, size_t dim, int order);
size_t so_vamana_incremental_repair(const float *vecs, size_t n, size_t dim, uint32_t *deg, uint32_t *nbr, size_t cap, size_t R, size_t L,
                                    float alpha, uint32_t medoid, uint32_t start, int order);
uint32_t so_vamana_build(const float *vecs, size_t n, size_t dim, uint32_t *deg, uint32_t *nbr, size_t cap, size_t R, size_t L, float alpha,
                         int order);

/* finalize_pooled with the nomic branch (minilm.rs:846-878) */
size_t so_finalize_pooled(const float *pooled, size_t n, int apply_prenorm, size_t out_dim, float *out);

/* ---- multi-threaded CPU baseline driver (bench.py cpu_baseline leg only) ------------ */
/* Runs so_brute_force_search[_select] for nq queries on `threads` pthreads (one query per
 * thread at a time, the reference's "concurrent readers under RwLock" shape). Returns
 * elapsed seconds. */
double so_bench_brute_force(const float *rows, size_t n, size_t dim, const float *queries,
                            size_t nq, size_t k, int order, int full_sort, int threads,
                            uint32_t *out_ids, float *out_dist);
/* the same with tombstones (one byte per row, non-zero = deleted; NULL = none) */
double so_bench_brute_force_del(const float *rows, size_t n, size_t dim, const uint8_t *deleted, const float *queries,
                                size_t nq, size_t k, int order, int full_sort, int threads,
                                uint32_t *out_ids, float *out_dist);
/* copy of `src` first-touched by `threads` threads in 2 MiB stripes (NUMA-spread corpus for the multi-threaded baseline);
 * release with so_free */
float *so_interleaved_copy(const float *src, size_t n_floats, int threads);
void so_free(void *p);

#ifdef __cplusplus
}
#endif
#endif

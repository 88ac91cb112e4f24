/*
 * shodh_hip.h -- C ABI of the MI355X-native embed-and-recall hot path of shodh-memory.
 *
 * This is the drop-in boundary.  The reference has no FFI seam on this
 * path -- MemorySystem owns Arc<MiniLMEmbedder> (src/memory/mod.rs:224) and RetrievalEngine
 * owns Arc<RwLock<VamanaIndex>> (src/memory/retrieval.rs:50-53) as concrete Rust types -- so the
 * seam is the METHOD SET of those types.  Each entry point below names the reference method
 * it replaces (file:line under the reference's src/).  A Rust `extern "C"` shim binding these
 * symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 (SHODH_OK) or a negative shodh_status; shodh_last_error() gives a
 *    thread-local UTF-8 message.  No exception or abort crosses the ABI.
 *  - plain pointers and sizes only.  "host" pointers are ordinary CPU memory; "*_device"
 *    entry points take HIP device pointers + a hipStream_t passed as void* (NULL = default
 *    stream) and are asynchronous on that stream.
 *  - the library owns opaque handles; the caller owns every in/out buffer.
 *  - thread-safety mirrors the reference's RwLock use (retrieval.rs:680,:712,:912):
 *    *_search / *_encode may run concurrently on one handle; *_add / *_build /
 *    *_mark_deleted / *_clear_deleted take the handle exclusively.
 *  - concurrent host-pointer calls that carry ONE item (a query, a text) -- the only call pattern the
 *    reference's `recall` / `remember` have (recall.rs:512-513, retrieval.rs:912-918,
 *    minilm.rs:889-897) -- are coalesced: calls that arrive while a device pass is in flight share
 *    the next pass and every caller gets the bytes its own call would have produced
 *    (*_set_coalesce, SHODH_COALESCE=0 to turn it off). A caller that is alone is not delayed.
 *  - the library is HIP/gfx950 only.  There is no CPU fallback: without a usable device every
 *    create call fails with SHODH_ERR_DEVICE.
 */
#ifndef SHODH_HIP_H
#define SHODH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHODH_HIP_ABI_VERSION 5

typedef enum {
    SHODH_OK = 0,
    SHODH_ERR_INVALID = -1,    /* bad argument / NULL pointer */
    SHODH_ERR_DIM = -2,        /* dimension mismatch (spann.rs:586-592 semantics) */
    SHODH_ERR_DEVICE = -3,     /* HIP error, no device, wrong architecture */
    SHODH_ERR_OOM = -4,
    SHODH_ERR_STATE = -5,      /* e.g. IVF-PQ search before trained state is set (spann.rs:612-617) */
    SHODH_ERR_IO = -6,
    SHODH_ERR_NONFINITE = -7,  /* NaN/Inf in rows or query: out of contract */
    SHODH_ERR_UNSUPPORTED = -8
} shodh_status;

/* DistanceMetric (src/vector_db/vamana.rs; persisted byte in vamana_persist.rs:98-112) */
enum { SHODH_METRIC_NDP = 0, SHODH_METRIC_EUCLIDEAN = 1, SHODH_METRIC_COSINE = 2 };
/* which reference build's accumulation order the exact scores reproduce bit-for-bit:
 * SCALAR4 = dot_product_scalar_inline (distance_inline.rs:157-173; default Linux build),
 * AVX2    = dot_product_avx2_inline   (distance_inline.rs:67-111; -C target-cpu=native build) */
enum { SHODH_ORDER_SCALAR4 = 0, SHODH_ORDER_AVX2 = 1,
       /* distance = 1 - sum(x*y), the sum strictly sequential: SpannIndex::compute_distance (spann.rs:562-571). Used by the
        * library itself for nearest-centroid searches (probe selection, k-means assignment); dist outputs are that distance */
       SHODH_ORDER_SEQ_1M = 2 };
/* index kinds: FLAT = VamanaIndex under SHODH_VECTOR_EXACT (vamana.rs:770-777, :1167-1188);
 * IVFPQ = SpannIndex (spann.rs:574-693) */
enum { SHODH_INDEX_FLAT = 0, SHODH_INDEX_IVFPQ = 1 };
/* scan strategy of the FLAT index. All three return identical results (bit-exact ids+dist):
 * EXACT = every score computed in reference order on the f32 rows (HBM-bound for nq <= ~16);
 * MFMA  = fp16 matrix-core pre-scan with a proven error bound + reference-order re-score of the
 *         surviving candidates; AUTO picks per call. */
enum { SHODH_SCAN_AUTO = 0, SHODH_SCAN_EXACT = 1, SHODH_SCAN_MFMA = 2,
       /* GRAPH = the reference's DEFAULT search path (no SHODH_VECTOR_EXACT): the Vamana graph is maintained on the device exactly as
        * add_vector (vamana.rs:853-974) / build (:200-284) maintain it, and search walks it like greedy_search (:576-657, :764-808).
        * Results are those of the reference's graph walk -- approximate by nature, bit-identical to the reference for an index
        * grown through add (a deterministic path) or carrying a reference-built graph (shodh_index_set_graph / a VAMA v1 file). */
       SHODH_SCAN_GRAPH = 3 };

typedef struct shodh_index shodh_index;
typedef struct shodh_embedder shodh_embedder;

typedef struct {
    uint32_t dim;           /* VamanaConfig.dimension (384; SHODH_TEXT_DIM 128..1024, multiple of 8) */
    uint32_t metric;        /* FLAT accepts SHODH_METRIC_NDP only, like RetrievalEngine (retrieval.rs:188-193) */
    uint32_t kind;          /* SHODH_INDEX_* */
    uint32_t order;         /* SHODH_ORDER_* */
    int32_t  device;        /* HIP device ordinal */
    uint32_t scan_mode;     /* SHODH_SCAN_* */
    uint64_t reserve_rows;  /* rows of HBM to reserve up front (grows by doubling) */
    uint64_t id_base;       /* global id of local row 0 (row-sharded multi-GPU corpora); ids returned = id_base + local */
    uint32_t nprobe;        /* IVFPQ: SpannConfig.num_probes (default 10; BackendConfig 20); capped at the partition count; at most 1024 lists per query */
    uint32_t reserved;
    /* SHODH_SCAN_GRAPH only (VamanaConfig, vamana.rs:120-166): */
    uint32_t max_degree;        /* 32 */
    uint32_t search_list_size;  /* 75 (VamanaConfig::default); the beam of build() */
    float    alpha;             /* 1.2 */
    uint32_t reserved2;
} shodh_index_cfg;

/* ---- library ------------------------------------------------------------------------------- */
const char *shodh_last_error(void);
int shodh_abi_version(void);
/* number of HIP devices visible; <0 on error */
int shodh_device_count(void);
/* fills name (<= cap bytes) and the gcnArchName of device `dev` */
int shodh_device_info(int dev, char *name, size_t cap, uint64_t *hbm_bytes, uint32_t *compute_units);

/* ---- index: VamanaIndex / VectorIndexBackend method set ------------------------------------- */
void shodh_index_cfg_default(shodh_index_cfg *cfg);                       /* BackendConfig::default, vector_db/mod.rs:74-86 */
int shodh_index_create(const shodh_index_cfg *cfg, shodh_index **out);    /* VamanaIndex::new vamana.rs:170-172 */
void shodh_index_destroy(shodh_index *idx);
/* add_vector (vamana.rs:853-974): appends n rows, ids are dense and sequential; *first_id_out =
 * id of rows[0] (= len() before the call, + id_base). rows: host [n][dim] row-major f32.
 * SHODH_SCAN_GRAPH only: when a walk meets thousands of equidistant rows and its frontier overflows, the rows ARE added and the call returns
 * SHODH_OK (an error status would invite a retry that adds them twice); shodh_index_graph_overflowed() then reports 1 and
 * shodh_last_error() holds the message -- see below. */
int shodh_index_add(shodh_index *idx, const float *rows, uint64_t n, uint32_t *first_id_out);
/* SHODH_SCAN_GRAPH: 1 when some add since the last build met a walk whose frontier outgrew its array (thousands of equidistant rows): the rows
 * WERE added and the add call returned SHODH_OK -- a walk cannot be undone, and an error status would invite a retry that adds them twice --
 * but the graph may differ from the reference's from there on (searches stay correct answers of THAT graph). Cleared by build. */
int shodh_index_graph_overflowed(const shodh_index *idx);
int shodh_index_add_device(shodh_index *idx, const float *d_rows, uint64_t n, uint32_t *first_id_out);
/* build (vamana.rs:200-284) / rebuild_from_vectors (:1363-1462): replaces the contents; ids 0..n-1;
 * tombstones cleared. */
int shodh_index_build(shodh_index *idx, const float *rows, uint64_t n);
int shodh_index_build_device(shodh_index *idx, const float *d_rows, uint64_t n);
/* search (vamana.rs:764-808 under SHODH_VECTOR_EXACT -> brute_force_search :1167-1188;
 * spann.rs:574-693 for IVFPQ). q: [nq][dim]. ids/dist: [nq][k], row i holds counts[i] valid
 * entries in ascending (dist by f32::total_cmp, id) order; the rest is filled with
 * 0xFFFFFFFF / +inf. Empty index -> counts 0, SHODH_OK (vamana.rs:766-768). k up to 7936 (the per-query selection buffers live
 * in LDS; the reference's callers ask for limit * 12 at most, retrieval.rs:913-918): more is SHODH_ERR_UNSUPPORTED. */
int shodh_index_search(shodh_index *idx, const float *q, uint32_t nq, uint32_t k,
                       uint32_t *ids, float *dist, uint32_t *counts);
/* VamanaIndex::brute_force_search (vamana.rs:1167-1188) whatever the scan mode: the exact scan of a FLAT index, graph mode included
 * (what the reference's estimate_recall compares its ANN answers with, vamana.rs:1128-1165) */
int shodh_index_brute_force_search(shodh_index *idx, const float *q, uint32_t nq, uint32_t k,
                                   uint32_t *ids, float *dist, uint32_t *counts);
/* SHODH_SCAN_GRAPH through the device-pointer entry point: bit 31 of d_counts[i] is set if walk i overflowed its frontier (thousands of
 * equidistant rows; the answer may then differ from the reference's) -- mask it off; the host-pointer entry point reports it as an error */
int shodh_index_search_device(shodh_index *idx, const float *d_q, uint32_t nq, uint32_t k,
                              uint32_t *d_ids, float *d_dist, uint32_t *d_counts, void *stream);
/* Coalescing front of the host-pointer search (on by default): concurrent shodh_index_search calls with nq <= 32 on a FLAT (non-graph) or IVF-PQ
 * index that arrive while a pass is in flight are gathered into ONE pass of up to 256 queries (k = the largest asked; a caller with a smaller k
 * gets the first k entries of its rows, which is its exact answer) and fanned back out -- `recall` is one query per call from many threads
 * (recall.rs:512-513, retrieval.rs:912-918). linger_us: how long a pass that could start waits for the callers of the pass that just ended
 * (default 30; a caller that is alone never waits). stats6: passes, calls served, largest pass, passes that lingered, microseconds spent inside
 * passes, microseconds spent lingering -- since the last reset. */
int shodh_index_set_coalesce(shodh_index *idx, int enabled, uint32_t linger_us);
int shodh_index_coalesce_stats(shodh_index *idx, uint64_t *stats6, int reset);
int shodh_index_mark_deleted(shodh_index *idx, uint32_t id, int *was_valid);   /* vamana.rs:813-820 */
/* mark_deleted for n ids in one call (one bitmask upload, one kernel over the shadow rows): *n_marked_out = ids that were
 * valid and not yet tombstoned. Same result as n mark_deleted calls. */
int shodh_index_mark_deleted_batch(shodh_index *idx, const uint32_t *ids, uint64_t n, uint64_t *n_marked_out);
int shodh_index_is_deleted(const shodh_index *idx, uint32_t id);               /* :823-825 (1/0, <0 error) */
uint64_t shodh_index_len(const shodh_index *idx);                              /* :184-186; IVF-PQ: postings held (SpannIndex::len) */
uint64_t shodh_index_deleted_count(const shodh_index *idx);                    /* :828-830 */
float shodh_index_deletion_ratio(const shodh_index *idx);                      /* :834-840 */
int shodh_index_needs_compaction(const shodh_index *idx);                      /* :843-845 (ratio >= 0.30) */
int shodh_index_clear_deleted(shodh_index *idx);                               /* :848-850 */
/* extract_all_vectors (vamana.rs; retrieval.rs:2504-2516: row i is returned bit-for-bit) */
int shodh_index_extract_rows(const shodh_index *idx, uint64_t first, uint64_t n, float *out_rows);
/* extract_live_vectors: rows that are not tombstoned, in id order; ids_out may be NULL */
int shodh_index_extract_live_rows(const shodh_index *idx, float *out_rows, uint32_t *ids_out, uint64_t cap, uint64_t *n_out);
uint32_t shodh_index_dim(const shodh_index *idx);
/* last search's device-side stage timings in microseconds (StageTiming.vector_search_us,
 * memory/mod.rs:3626-3630): scan, select, rerank, total */
int shodh_index_stage_timings(const shodh_index *idx, float *us4);
/* mean / min duration (microseconds) of the dominant scan kernel (MFMA emit scan, or the exact
 * scan) over the searches issued since the last reset, from HIP events recorded on the stream each
 * search ran on. The caller must have synchronised those streams. Used by bench.py's roofline. */
int shodh_index_kernel_timing(shodh_index *idx, int reset, float *mean_us, float *min_us, uint32_t *count);
/* diagnostics of the last host-pointer MFMA-path search, 8 values: [0]=rows sampled for the thresholds, [1]=candidates emitted
 * by the pre-scan, [2]=candidates re-scored in reference order, [3]=queries that fell back to the exact scan of the corpus,
 * [4]=queries whose fp16 window was narrowed by the level-2 f32 filter (dense corpora), [5..7] reserved (0) */
int shodh_index_scan_stats(const shodh_index *idx, uint64_t *stats8);

/* ---- SHODH_SCAN_GRAPH: the Vamana graph itself ------------------------------------------------------------------------------- */
/* attach a graph to the rows the index holds (e.g. the degree / neighbour arrays of a VAMA v1 file, vamana_persist.rs:290-391):
 * deg [len], nbr [len][stride] (entries beyond deg[i] ignored), medoid = entry point */
int shodh_index_set_graph(shodh_index *idx, const uint32_t *deg, const uint32_t *nbr, uint32_t stride, uint32_t medoid);
/* replace the contents by `rows` WITH their graph, nothing constructed (VamanaIndex::load_from_file, vamana_persist.rs:290-424) */
int shodh_index_build_with_graph(shodh_index *idx, const float *rows, uint64_t n, const uint32_t *deg, const uint32_t *nbr, uint32_t stride, uint32_t medoid);
/* read it back: deg [len], nbr [len][stride] with stride >= max_degree + 1; any output may be NULL */
int shodh_index_get_graph(const shodh_index *idx, uint32_t *deg, uint32_t *nbr, uint32_t stride, uint32_t *medoid);
/* VamanaIndex::build's graph construction over the rows the index holds (vamana.rs:200-284): find_medoid, then up to two passes of
 * greedy_search(search_list_size) + robust_prune(alpha) + back edges, node after node. The reference starts from a random graph
 * drawn from thread_rng (:287-312): pass it as init_deg / init_nbr [len][init_stride] (each list <= max_degree) to reproduce a
 * given run bit for bit, or NULL to draw one here from `seed`. */
int shodh_index_vamana_build(shodh_index *idx, uint64_t seed, const uint32_t *init_deg, const uint32_t *init_nbr, uint32_t init_stride);
/* VamanaIndex::incremental_repair (vamana.rs:1033-1115) for nodes [first_node, first_node + count): walk(search_list_size) from the
 * medoid, robust_prune(alpha), and where a list changed its stale back edges are removed and the new ones pushed (truncated to max_degree).
 * repaired_out = nodes whose list changed. Which nodes to repair and the insert counter are the caller's (the reference: the last
 * min(inserts, 1000) nodes once 1000 inserts have accumulated). */
int shodh_index_incremental_repair(shodh_index *idx, uint32_t first_node, uint32_t count, uint32_t *repaired_out);

/* ---- multi-GPU: merge of per-shard results ------------------------------------------------------- */
/* Row-sharded corpora: every rank searches its shard (ids carry id_base), the
 * per-shard (ids, dist) blocks are all-gathered (RCCL) into [n_lists][nq][k] and merged here by
 * (dist total_cmp, id) -- the same comparator as vamana.rs:1185, so the merged list equals a
 * single-device search of the concatenated corpus. Padding entries have id 0xFFFFFFFF. */
int shodh_topk_merge_device(const uint32_t *d_in_ids, const float *d_in_dist, uint32_t n_lists, uint32_t nq, uint32_t k,
                            uint32_t *d_ids, float *d_dist, uint32_t *d_counts, void *stream);
/* the same with `list_stride` elements between the blocks of consecutive lists: lets every rank send ONE packed
 * [ids | dist] buffer through ONE all-gather (in_dist = in_ids + nq*k reinterpreted, list_stride = 2*nq*k) */
int shodh_topk_merge_strided_device(const uint32_t *d_in_ids, const float *d_in_dist, uint64_t list_stride, uint32_t n_lists, uint32_t nq, uint32_t k,
                                    uint32_t *d_ids, float *d_dist, uint32_t *d_counts, void *stream);

/* ---- multi-GPU: ONE index over the GPUs of a node, one host process (SURVEY.md 8e / 8b `devices[], n_devices`) ---------------
 * The VamanaIndex / SpannIndex method set again, over G shards, each an ordinary shodh_index on its own device. A search runs the
 * query batch on every shard at once, exchanges the per-shard top-k ([ids | dist], 8*nq*k bytes per shard) with RCCL ncclAllGather
 * (communicators from ncclCommInitAll; librccl is bound at run time) and merges with the comparator of vamana.rs:1185 on the
 * first device: results are bit-identical to one index holding the whole corpus. Duplicate device ordinals are allowed (several
 * shards on one GPU, e.g. for testing on a single-GPU host); the exchange then uses device-to-device copies.
 * FLAT: ids stay dense and sequential (vamana.rs:854-855) and are dealt to the shards in blocks of 2^block_log2 rows, round robin,
 * so appends stay balanced. IVFPQ: every posting list is cut into G pieces, shard g holds piece g of every list.
 * Searches may run concurrently on one handle: every call takes its own set of streams and exchange buffers on every shard (up to
 * SHODH_SHARD_SLOTS = 4 in flight); only the enqueue of a call's commands is serialised (RCCL wants one issue order per communicator).
 * Concurrent host-pointer searches of a few queries are coalesced like shodh_index_search's (one pass over the shards, ONE exchange). */
enum { SHODH_EXCHANGE_AUTO = 0,    /* RCCL when the devices are distinct and librccl loads, device copies otherwise */
       SHODH_EXCHANGE_RCCL = 1,    /* RCCL or fail */
       SHODH_EXCHANGE_COPY = 2 };  /* hipMemcpyAsync device-to-device into the first device */
typedef struct shodh_sharded_index shodh_sharded_index;
typedef struct {
    uint32_t dim, metric, kind, order, scan_mode, nprobe;   /* as in shodh_index_cfg, applied to every shard */
    uint32_t block_log2;             /* FLAT: rows per round-robin block = 2^block_log2 (default 16) */
    uint32_t exchange;               /* SHODH_EXCHANGE_* */
    uint64_t reserve_rows_per_shard;
} shodh_sharded_cfg;
void shodh_sharded_cfg_default(shodh_sharded_cfg *cfg);
int shodh_sharded_index_create(const shodh_sharded_cfg *cfg, const int32_t *devices, uint32_t n_devices, shodh_sharded_index **out);
void shodh_sharded_index_destroy(shodh_sharded_index *s);
uint32_t shodh_sharded_index_shards(const shodh_sharded_index *s);
int shodh_sharded_index_uses_rccl(const shodh_sharded_index *s);                 /* 1 = the exchange is an RCCL all-gather */
uint64_t shodh_sharded_index_len(const shodh_sharded_index *s);                  /* vamana.rs:184-186 / SpannIndex::len */
uint64_t shodh_sharded_index_shard_len(const shodh_sharded_index *s, uint32_t shard);
int shodh_sharded_index_build(shodh_sharded_index *s, const float *rows, uint64_t n);                            /* vamana.rs:200-284 */
int shodh_sharded_index_add(shodh_sharded_index *s, const float *rows, uint64_t n, uint32_t *first_id_out);      /* :853-974 */
int shodh_sharded_index_search(shodh_sharded_index *s, const float *q, uint32_t nq, uint32_t k,
                               uint32_t *ids, float *dist, uint32_t *counts);
/* the same with device buffers ON THE FIRST DEVICE of the index (devices[0]): d_q [nq][dim], d_ids / d_dist [nq][k], d_counts [nq];
 * asynchronous on `stream` (a hipStream_t of that device, NULL = its default stream): the queries reach the other shards by peer copies,
 * the merged lists land in the caller's buffers, no host round trip. Queries are not screened for NaN / Inf (the host-pointer form is). */
int shodh_sharded_index_search_device(shodh_sharded_index *s, const float *d_q, uint32_t nq, uint32_t k, uint32_t *d_ids, float *d_dist,
                                      uint32_t *d_counts, void *stream);                                    /* :764-808, :1167-1188; spann.rs:574-693 */
int shodh_sharded_index_mark_deleted(shodh_sharded_index *s, uint32_t id, int *was_valid);                        /* :813-820 */
int shodh_sharded_index_mark_deleted_batch(shodh_sharded_index *s, const uint32_t *ids, uint64_t n, uint64_t *n_marked_out);
int shodh_sharded_index_is_deleted(shodh_sharded_index *s, uint32_t id);
uint64_t shodh_sharded_index_deleted_count(shodh_sharded_index *s);
int shodh_sharded_index_clear_deleted(shodh_sharded_index *s);
int shodh_sharded_index_extract_rows(shodh_sharded_index *s, uint64_t first, uint64_t n, float *out_rows);        /* by global id, bit-for-bit */
int shodh_sharded_index_set_ivfpq(shodh_sharded_index *s, const float *centroids, uint32_t P, const float *codebook, uint32_t M,
                                  uint32_t ncent, const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes);
int shodh_sharded_index_ivfpq_insert(shodh_sharded_index *s, uint32_t vector_id, const float *row);             /* spann.rs:1006-1051, shard = id % G */
int shodh_sharded_index_set_coalesce(shodh_sharded_index *s, int enabled, uint32_t linger_us);     /* see shodh_index_set_coalesce */
int shodh_sharded_index_coalesce_stats(shodh_sharded_index *s, uint64_t *stats6, int reset);
/* host wall clock of the last search, microseconds: enqueue of the shard searches, exchange enqueue, merge + wait, total */
int shodh_sharded_index_host_timings(const shodh_sharded_index *s, float *us4);
/* which librccl was bound and its version ("<path> version <n>"); SHODH_ERR_DEVICE if none could be loaded */
int shodh_rccl_info(char *buf, size_t cap);

/* ---- IVF-PQ trained state (SpannIndex given centroids/codebooks/postings) -------------------- */
/* The reference's k-means is unseeded (spann.rs:472-474, pq.rs:155-157) so parity is defined
 * GIVEN trained state. centroids [P][dim]; codebook [M][ncent][8] (M = dim/8, ncent <= 256);
 * postings in CSR form: list_off[P+1], ids[total], codes[total][M] in insertion order. */
int shodh_index_set_ivfpq(shodh_index *idx, const float *centroids, uint32_t P,
                          const float *codebook, uint32_t M, uint32_t ncent,
                          const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes);
/* SpannIndex::insert (spann.rs:1006-1051): nearest centroid (strict '<') + PQ encode, appended
 * to that partition's posting list. */
int shodh_index_ivfpq_insert(shodh_index *idx, uint32_t vector_id, const float *row);
/* find_nearest_centroid + ProductQuantizer::encode for n rows on the device (spann.rs:545-558,
 * pq.rs:220-257): assign_out[n], codes_out[n][M] */
int shodh_index_ivfpq_encode(shodh_index *idx, const float *rows, uint64_t n, uint32_t *assign_out, uint8_t *codes_out);
/* Lloyd k-means on the device following spann.rs:466-541 / pq.rs:152-217 operation for operation: GIVEN the initial
 * shuffles (init_perm_ivf[n], init_perm_pq[M][n], M = dim/8; the reference draws them from thread_rng) centroids_out
 * [P][dim] and codebook_out [M][256][8] are bit-identical to the reference's. rows: host [n][dim]. IVF stops early
 * when no assignment changed; PQ runs all pq_iters iterations; empty clusters keep their centroid. */
int shodh_ivfpq_train(int device, const float *rows, uint64_t n, uint32_t dim, uint32_t P, uint32_t ivf_iters,
                      uint32_t pq_iters, const uint32_t *init_perm_ivf, const uint32_t *init_perm_pq,
                      float *centroids_out, float *codebook_out);

/* ---- pairwise similarity (src/similarity.rs:10-48) -------------------------------------------- */
/* cosine_similarity for n pairs: a,b host [n][dim]; out[n]. (reference order, clamp to [-1,1],
 * 0 on zero norm) */
int shodh_cosine_similarity_batch(int device, const float *a, const float *b, uint64_t n, uint32_t dim,
                                  uint32_t order, float *out);
/* top_k_similar (similarity.rs:27-48): cosine_similarity(query, cands[i]) for i < n (0.0 for every candidate when
 * query_dim != dim, :11-13), STABLE sort by OrderedFloat score descending (equal scores keep input order; NaN sorts first,
 * -0.0 == +0.0), first min(k, n) kept. out_scores/out_index hold k entries; out_index[i] = position of the candidate in
 * `cands` (the caller's items T are looked up with it); *count_out = entries written. */
int shodh_top_k_similar(int device, const float *query, uint32_t query_dim, const float *cands /*[n][dim]*/, uint64_t n,
                        uint32_t dim, uint64_t k, uint32_t order, float *out_scores, uint32_t *out_index, uint64_t *count_out);

/* ---- embedder: trait Embedder / MiniLMEmbedder (src/embeddings/mod.rs:52-88, minilm.rs) ------- */
enum { SHODH_DTYPE_FP32 = 0, SHODH_DTYPE_BF16 = 1,
       /* the reference's DEFAULT model is the ONNX Runtime dynamic-quantisation export model_quint8_avx2.onnx (downloader.rs:31,
        * minilm.rs:212-220): 8-bit weights and word table with the file's own scales and zero points (per tensor or per channel; see
        * shodh_embedder_load_file), DynamicQuantizeLinear uint8 activations per dense layer, MatMulInteger int32 accumulation on
        * v_mfma_i32_32x32x32_i8, fp32 softmax / GELU / LayerNorm. Parity with that file is unpinned (no ONNX Runtime / checkpoint
        * offline): what is implemented is the published semantics of those ONNX operators */
       SHODH_DTYPE_INT8 = 2 };
typedef struct {
    int32_t  device;
    uint32_t dtype;          /* GEMM operand type (accumulation is always fp32) */
    uint32_t max_len;        /* EmbeddingConfig.max_length = 256 (minilm.rs:225) */
    uint32_t vocab, hidden, layers, heads, intermediate, max_pos, type_vocab;  /* 30522,384,6,12,1536,512,2 */
    float    ln_eps;         /* 1e-12 */
    uint32_t compute_padded; /* 0: only real tokens (exact in fp32/bf16, minilm.rs:153-154); 1 (INT8 only): all max_len positions of every
                              * non-empty text, as the reference's tensor has them -- DynamicQuantizeLinear takes its range over
                              * the padded tensor, so the padding is part of the INT8 embedding function (minilm.rs:588-593) */
    uint32_t quant_scope;    /* INT8 only: the tensor a DynamicQuantizeLinear range spans when encode_ids gets b > 1 texts (SHODH_QUANT_SCOPE_*, below).
                              * BATCH (0, default) = the reference's encode_batch: one session.run on [B, max_len] (minilm.rs:996-1115), ranges over the
                              * whole batch tensor. PER_TEXT (1) = B calls of the reference's encode(): one session.run on [1, max_len] each
                              * (minilm.rs:883-982) -- what remember / index_memory / recall compute (memory/mod.rs:1037, retrieval.rs:673, :708,
                              * :878): every range spans ONE text's padded tensor, a batch of N texts is bit-identical to N calls with b = 1 */
    const char *weights_path; /* NULL, or the model file shodh_embedder_create loads (EmbeddingConfig.model_path, minilm.rs:212-220): model.safetensors,
                              * model.onnx, or the dynamic-quantisation export model_quantized.onnx -- see shodh_embedder_load_file */
} shodh_embed_cfg;
enum { SHODH_QUANT_SCOPE_BATCH = 0, SHODH_QUANT_SCOPE_PER_TEXT = 1 };
void shodh_embed_cfg_default(shodh_embed_cfg *cfg);
int shodh_embedder_create(const shodh_embed_cfg *cfg, shodh_embedder **out);   /* MiniLMEmbedder::new minilm.rs:652-690 */
void shodh_embedder_destroy(shodh_embedder *e);
/* number of f32 parameters expected by shodh_embedder_load_weights, in HF BertModel order
 * (see DESIGN.md "encoder weight blob") */
uint64_t shodh_embedder_param_count(const shodh_embedder *e);
int shodh_embedder_load_weights(shodh_embedder *e, const float *blob, uint64_t n_floats);
/* ---- weights as the reference's files hold them (minilm.rs:96-98 hands the .onnx file to ONNX Runtime; downloader.rs:29-53 names the
 * files: onnx/model.onnx, onnx/model_quint8_avx2.onnx; the checkpoint itself is model.safetensors) -------------------------------------
 * shodh_embedder_load_file reads
 *   .safetensors  F32 / F16 / BF16 tensors under their HF BertModel names (any "prefix." in front, pooler ignored);
 *   .onnx         the fp32 export, or the onnxruntime dynamic-quantisation export: initialisers by HF name where the exporter kept them,
 *                 MatMul / MatMulInteger constants ([K][N], anonymous) by the bias their result is added to, <w>_quantized / <w>_scale /
 *                 <w>_zero_point triples per tensor or per output channel, uint8 or int8.
 * With SHODH_DTYPE_INT8 an export's quantised tensors are used AS THEY ARE (bytes, scales, zero points: MatMulInteger's
 * sum (a - a_zp)(b - b_zp) with b_zp of any value); tensors that arrive as floats are quantised by this library (per tensor, symmetric:
 * scale = 2 max|w| / 255) -- a fallback for checkpoints without an export, not the export's arithmetic. fp32 / bf16 modes take
 * the dequantised values of quantised tensors. */
int shodh_embedder_load_file(shodh_embedder *e, const char *path);
/* the same hand-over one tensor at a time (a host that parses the model file itself): names are HF BertModel parameter names
 * ("embeddings.word_embeddings.weight", "encoder.layer.0.attention.self.query.weight", ...; see shodh_weight_file_* for the list).
 * load_tensor: f32, [rows][cols] as HF stores it, or [cols][rows] with transposed != 0 (ONNX MatMul constants are [K][N]).
 * load_quantized: uint8 (is_signed = 0) or int8 bytes in the same layouts, n_scale = 1 (per tensor) or N (per output channel) scales and
 * zero points (zero_point NULL = 0; same integer type as the bytes); allowed for the six dense weights per layer and the word table.
 * finish_weights checks that every parameter arrived and builds the device copies; encode calls fail with SHODH_ERR_STATE before it. */
int shodh_embedder_load_tensor(shodh_embedder *e, const char *name, const float *data, uint64_t n, uint32_t transposed);
int shodh_embedder_load_quantized(shodh_embedder *e, const char *name, const void *q, uint32_t is_signed, uint32_t transposed,
                                  const float *scale, const void *zero_point, uint32_t n_scale);
int shodh_embedder_finish_weights(shodh_embedder *e);
/* where a parameter's device copy came from: ABSENT (nothing loaded), F32, EXPORT_Q8 (INT8 mode multiplies the file's own bytes),
 * SELF_Q8 (INT8 mode quantised f32 values itself) */
enum { SHODH_WEIGHT_ABSENT = 0, SHODH_WEIGHT_F32 = 1, SHODH_WEIGHT_EXPORT_Q8 = 2, SHODH_WEIGHT_SELF_Q8 = 3 };
int shodh_embedder_weight_source(const shodh_embedder *e, const char *name, uint32_t *source_out);
/* host-only view of a weight file (no device needed; what shodh_embedder_load_file parses): the f32 blob in shodh_embedder_load_weights
 * order (quantised tensors dequantised, (q - zp) * scale), and per name the quantised form in this library's storage convention --
 * signed bytes [N][K] (uint8 sources minus 128), n_scale scales, zero points in the same signed terms. n_scale_out = 0: stored as floats. */
typedef struct shodh_weight_file shodh_weight_file;
int shodh_weight_file_open(const char *path, const shodh_embed_cfg *cfg, shodh_weight_file **out);
void shodh_weight_file_close(shodh_weight_file *f);
int shodh_weight_file_blob(const shodh_weight_file *f, float *blob_out, uint64_t n_floats);
int shodh_weight_file_quantized(const shodh_weight_file *f, const char *name, int8_t *q_out, uint64_t q_len, float *scale_out,
                                int32_t *zero_point_out, uint32_t scale_cap, uint32_t *n_scale_out);
/* deterministic synthetic weights (normal std 0.02, LayerNorm gamma 1 beta 0) from a seed; the
 * same blob is returned to the host when blob_out != NULL so a checker can mirror it */
int shodh_embedder_init_synthetic(shodh_embedder *e, uint64_t seed, float *blob_out, uint64_t n_floats);
/* host-only helpers (no device needed): parameter count of a configuration and the synthetic blob */
uint64_t shodh_embed_param_count(const shodh_embed_cfg *cfg);
int shodh_embedder_synthetic_weights(const shodh_embed_cfg *cfg, uint64_t seed, float *blob_out, uint64_t n_floats);
uint32_t shodh_embedder_dimension(const shodh_embedder *e);                    /* Embedder::dimension mod.rs:63 */
/* encode_batch after tokenisation (minilm.rs:996-1115): ids int32 [b][max_len], mask uint8
 * [b][max_len] (1 = real token). out host [b][hidden]: masked mean-pool, NaN/Inf scrub,
 * L2-normalise (minilm.rs:959-981, :846-878). A row whose mask is all zero yields zeros
 * (empty text, minilm.rs:1123-1125). */
int shodh_embedder_encode_ids(shodh_embedder *e, const int32_t *ids, const uint8_t *mask, uint32_t b, float *out);
int shodh_embedder_encode_ids_device(shodh_embedder *e, const int32_t *d_ids, const uint8_t *d_mask, uint32_t b,
                                     float *d_out, void *stream);
/* The same with the scope of THIS call as an argument (SHODH_QUANT_SCOPE_*), read nowhere else: concurrent callers with different scopes cannot
 * disturb each other (set_quant_scope + encode + restore cannot be made atomic by a caller). INT8: which tensor the DynamicQuantizeLinear ranges
 * span. fp32 / bf16: PER_TEXT (and every call with b = 1) runs the kernel forms a single text takes whatever the size of the batch, so a text's
 * embedding is the same BYTES alone, in encode_each, or coalesced with other callers; BATCH lets the library pick the fastest forms for the
 * batch (same function, bf16 rounding-level differences between forms). */
int shodh_embedder_encode_ids_scoped(shodh_embedder *e, const int32_t *ids, const uint8_t *mask, uint32_t b, uint32_t scope, float *out);
int shodh_embedder_encode_ids_device_scoped(shodh_embedder *e, const int32_t *d_ids, const uint8_t *d_mask, uint32_t b, uint32_t scope,
                                            float *d_out, void *stream);
/* Coalescing front of the host-pointer encode (on by default): concurrent calls with b = 1 -- `remember` / `recall` embed one text per call behind
 * Mutex<Session> (minilm.rs:889-897) -- share ONE per-text forward (a one-text call is the same function under either scope, and PER_TEXT makes a
 * forward over N texts N x that function): every caller gets the bytes its own call would have produced. Up to SHODH_ENC_SLOTS = 2 forwards are in
 * flight per handle (calls with b > 1 run on their own). stats6 as for shodh_index_coalesce_stats. */
int shodh_embedder_set_coalesce(shodh_embedder *e, int enabled, uint32_t linger_us);
int shodh_embedder_coalesce_stats(shodh_embedder *e, uint64_t *stats6, int reset);
/* switch the quantisation scope of later encode calls (cfg.quant_scope; no reallocation): a host calls PER_TEXT for bulk `remember` ingest
 * (N x encode()) and BATCH where the reference itself calls encode_batch (memory/mod.rs:8443, :8838). INT8: which tensor the activation ranges
 * span. fp32 / bf16: the NUMBERS of a text never depend on its batch mates, but PER_TEXT selects the kernel forms a one-text call takes (three-kernel
 * feed-forward, K-split down projection) for every batch size, so that encode_each(texts)[i] == encode(texts[i]) byte for byte; it is slower than
 * BATCH for bulk bf16 ingest (no fused feed-forward) and runs in sub-batches of 1024 texts to bound the K-split scratch. Leave BATCH where byte
 * equality with one-text calls is not needed. */
int shodh_embedder_set_quant_scope(shodh_embedder *e, uint32_t scope);
uint32_t shodh_embedder_quant_scope(const shodh_embedder *e);
int shodh_embedder_stage_timings(const shodh_embedder *e, float *us2);         /* StageTiming.embedding_us */
/* One dynamically quantised dense layer, y = dequant(MatMulInteger(DynamicQuantizeLinear(x), quantise(w))) + bias, on host data:
 * x [M][K], w [N][K] (quantised per tensor, symmetric 8 bit), bias [N] or NULL, y [M][N]; N % 128 == 0, K % 128 == 0.
 * Optional outputs: acc_out [M][N] the exact int32 accumulators sum_k (a - a_zp)(w_q), a_scale / a_zp the activation
 * quantisation parameters, w_scale the weight scale. The building block of SHODH_DTYPE_INT8. */
int shodh_int8_dense(int device, const float *x, const float *w, const float *bias, uint32_t M, uint32_t N, uint32_t K,
                     float *y, int32_t *acc_out, float *a_scale, int32_t *a_zp, float *w_scale);
/* the same layer on a weight that is ALREADY quantised (an export's tensor): wq uint8 (is_signed = 0) or int8 [N][K], n_scale = 1 or N
 * scales and zero points (w_zero_point NULL = 0, same integer type as wq). acc_out = sum_k (a - a_zp)(wq - w_zp), exact (MatMulInteger). */
int shodh_int8_dense_quantized(int device, const float *x, const void *wq, uint32_t is_signed, const float *w_scale, const void *w_zero_point,
                               uint32_t n_scale, const float *bias, uint32_t M, uint32_t N, uint32_t K, float *y, int32_t *acc_out,
                               float *a_scale, int32_t *a_zp);

/* ---- host-side glue of the same path (string / uuid work: stays on the host by design) ------------------ */
/* MiniLMEmbedder::new_simplified / generate_embedding_simplified (minilm.rs:777-831): SipHash-1-3
 * (DefaultHasher, zero keys) over whitespace words and char bigrams -> dim-d unit vector; zeros if
 * normalisation fails. The reference's unit tests index with this embedder (retrieval.rs:2451-2455). */
int shodh_hash_embed(const char *utf8, size_t len, uint32_t dim, float *out);
/* MiniLMEmbedder::finalize_pooled (minilm.rs:846-878) on a pooled vector, both branches: NaN/Inf scrub; apply_prenorm != 0 (the
 * nomic recipe, SHODH_EMBEDDER=nomic) adds the parameter-free LayerNorm over the full width n; truncation to out_dim (Matryoshka);
 * L2 over the kept prefix. out holds min(n, out_dim) floats; returns that length. (The device encoder applies the MiniLM branch
 * itself; a host running another ONNX tower can finish its pooled vectors here.) */
size_t shodh_finalize_pooled(const float *pooled, size_t n, int apply_prenorm, size_t out_dim, float *out);
/* RetrievalEngine::search_ids post-processing (retrieval.rs:920-963): vector ids -> memory ids
 * (vector_to_memory: [n_vectors][16] uuid bytes, all-0xFF = unmapped), similarity = -distance, per-memory
 * max over chunks (strict '>'), sort (similarity total_cmp desc, uuid asc), truncate. Returns the count. */
size_t shodh_search_ids_postprocess(const uint32_t *vec_ids, const float *dists, size_t n_res,
                                    const uint8_t *vector_to_memory, size_t n_vectors, size_t limit,
                                    uint8_t *out_uuid /*[limit][16]*/, float *out_sim /*[limit]*/);
/* RRFusion::new + fuse (memory/hybrid_search.rs:536-594): n_lists ranked uuid lists concatenated in
 * `uuids` (list l has list_len[l] entries); output sorted (score desc, uuid asc). Returns the count. */
size_t shodh_rrf_fuse(float k, const float *weights, size_t n_lists, const uint8_t *uuids, const size_t *list_len,
                      uint8_t *out_uuid, float *out_score, size_t out_cap);

/* ---- recall Layer 4: fusion of the hybrid (vector + BM25) leg with the graph leg (memory/mod.rs:3878-4468) ------
 * The step right after search_ids on the recall path. Pure arithmetic over at most a few hundred
 * candidates per request: host code. The reference reads its switches from environment variables inside recall; here they
 * are fields, and shodh_leg_fusion_cfg_default() gives the reference's behaviour with every variable unset. */
typedef struct {
    /* fusion mode switches: SHODH_FUSION_V2 / _FLAT / _SUM / _RRF (mod.rs:3951-3953, :3964-3966, :4020-4022, :4041-4044).
     * flat = fusion_flat || !(fusion_rrf || fusion_v2 || fusion_sum); the legs test them in the reference's order */
    uint8_t fusion_v2, fusion_flat, fusion_sum, fusion_rrf;
    uint8_t isolate_leg;          /* SHODH_LEG (:3975-3988): 0 unset, 1 vector, 2 bm25, 3 graph */
    uint8_t flat_adaptive;        /* SHODH_FLAT_ADAPTIVE, default on (:4063-4065) */
    uint8_t adapt_feature;        /* SHODH_ADAPT_FEATURE: 0 fitted (default), 1 agreement, 2 peak (:4090-4092) */
    uint8_t adapt_symmetric;      /* SHODH_ADAPT_SYMMETRIC, default on (:4237-4244) */
    float   graph_w, hybrid_w;    /* density weights, see shodh_leg_fusion_weights (defaults 0.3 / 0.7, :3878-3880, :3921) */
    float   rrf_k;                /* RRF_K_GRAPH_FUSION = 30 (constants.rs:1198) */
    float   flat_consensus;       /* SHODH_FLAT_CONSENSUS 0.3, clamped to [0,1] (:3990-3994) */
    float   adapt_trust_max;      /* SHODH_ADAPT_TRUST_MAX 2.0 (:4067) */
    float   fw_graph, fw_vec, fw_bm25;          /* SHODH_FW_* 0.3 / 0.6 / 0.4 (:4029-4031) */
    float   agree_k, agree_lo, agree_hi;        /* SHODH_ADAPT_AGREE_* 10 / 0.1 / 0.5 (:4170-4172) */
    float   peak_lo, peak_hi;                   /* SHODH_ADAPT_PEAK_* 2 / 6 (:4207-4208) */
} shodh_leg_fusion_cfg;
void shodh_leg_fusion_cfg_default(shodh_leg_fusion_cfg *c);
/* calculate_density_weights (memory/graph_retrieval.rs:81-101): out = {semantic_w, graph_w, linguistic_w} */
void shodh_density_weights(float graph_density, float *out3 /*[3]*/);
/* mod.rs:3878-3921: (semantic, graph, linguistic) = density weights, or (0.6, 0.3, 0.1) without a density (has_density = 0);
 * graph_weight_override (SHODH_GRAPH_FUSION_WEIGHT) and graph_w_floor (SHODH_GRAPH_W_FLOOR) are NaN when unset;
 * hybrid_w = semantic_w + linguistic_w */
void shodh_leg_fusion_weights(int has_density, float graph_density, float graph_weight_override, float graph_w_floor,
                              float *graph_w, float *hybrid_w);
/* The fusion itself. hybrid leg: n_hybrid candidates in hybrid rank order, uuid [n][16] with their BM25 and vector component
 * scores (HybridSearchResult.bm25_score / vector_score, 0 when absent; a repeated uuid keeps its LAST components, like the
 * HashMap insert at :3832-3838). graph leg: n_graph candidates in activation rank order with their activation.
 * query_len = query_text.len() in bytes (a feature of the fitted gate, :4162). Output: every fused candidate with its score,
 * sorted (score total_cmp desc, uuid asc) -- the reference keeps them in a HashMap, the order is ours. vec_trust_out
 * (may be NULL) receives effective_vec_trust. Returns the number of fused candidates (only the first out_cap are written). */
size_t shodh_fuse_legs(const shodh_leg_fusion_cfg *c, const uint8_t *hybrid_uuid, const float *hybrid_bm25, const float *hybrid_vec,
                       size_t n_hybrid, const uint8_t *graph_uuid, const float *graph_activation, size_t n_graph, size_t query_len,
                       uint8_t *out_uuid /*[out_cap][16]*/, float *out_score /*[out_cap]*/, size_t out_cap, float *vec_trust_out);

/* ---- on-disk index formats of the reference (host code; both little-endian, FNV-1a-64 over everything after the header) --- */
/* VAMA v1: src/vector_db/vamana_persist.rs:6-34 layout, :98-112 header bytes, :175-284 save_to_file, :290-391 load_from_file */
typedef struct {
    uint64_t num_vectors;
    uint32_t dimension, max_degree, medoid, deleted_count;
    uint64_t incremental_inserts;
    uint64_t graph_edges;          /* sum of the per-node neighbour counts (size of the `neighbors` array) */
    uint8_t  distance_metric;      /* 0 NormalizedDotProduct, 1 Euclidean, 2 Cosine (:87-91) */
} shodh_vama_info;
int shodh_vama_info_read(const char *path, shodh_vama_info *out);   /* validates magic, version and checksum (verify_index_file, :410-424) */
/* any output may be NULL; sizes come from shodh_vama_info_read: vectors [num_vectors*dimension], deleted [deleted_count],
 * degree [num_vectors], neighbors [graph_edges] */
int shodh_vama_load(const char *path, float *vectors, uint32_t *deleted, uint16_t *degree, uint32_t *neighbors);
/* degree == NULL writes an empty adjacency for every node (this library searches exactly; the graph is not built) */
int shodh_vama_save(const char *path, const float *vectors, uint64_t n, uint32_t dim, uint32_t max_degree, uint32_t medoid,
                    uint8_t metric, const uint32_t *deleted, uint32_t deleted_count, uint64_t incremental_inserts,
                    const uint16_t *degree, const uint32_t *neighbors);
/* SPAN v1: src/vector_db/spann.rs:13-52 layout, :221-252 header bytes, :750-876 save_to_file, :879-1003 load_from_file */
typedef struct {
    uint64_t num_vectors, total_postings;
    uint32_t num_partitions, dimension, pq_subvectors, pq_num_centroids, pq_subvec_dim;
    uint8_t  pq_enabled, distance_metric;
} shodh_span_info;
int shodh_span_info_read(const char *path, shodh_span_info *out);
/* outputs (any may be NULL): centroids [P*dim], codebook [M*256*8], list_off [P+1] (CSR), ids [total_postings],
 * codes [total_postings*M] -- exactly the arrays shodh_index_set_ivfpq takes */
int shodh_span_load(const char *path, float *centroids, float *codebook, uint64_t *list_off, uint32_t *ids, uint8_t *codes);
/* codebook == NULL writes a PQ-less file (ids only) */
int shodh_span_save(const char *path, uint64_t num_vectors, uint32_t num_partitions, uint32_t dim, uint32_t pq_subvectors, uint8_t metric,
                    const float *centroids, const float *codebook, const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes);

/* ---- fusion: LearnedWeights (src/relevance.rs:343-606) ----------------------------------------- */
typedef struct {
    float semantic, entity, tag, importance, momentum, access_count, graph_strength;
    uint32_t update_count;
} shodh_weights;
void  shodh_weights_default(shodh_weights *w);                                  /* relevance.rs:383-397 */
void  shodh_weights_normalize(shodh_weights *w);                                /* :401-418 */
void  shodh_weights_apply_feedback(shodh_weights *w, int semantic_contributed, int entity_contributed,
                                   int tag_contributed, int helpful);           /* :427-465 */
float shodh_calibrate_score(float score);                                       /* :601-606 */
float shodh_fuse_scores(const shodh_weights *w, float sem, float ent, float tag, float imp);               /* :471-487 */
float shodh_fuse_scores_with_momentum(const shodh_weights *w, float sem, float ent, float tag, float imp,
                                      float momentum_ema);                      /* :499-517 */
float shodh_fuse_scores_full(const shodh_weights *w, float sem, float ent, float tag, float imp,
                             float momentum_ema, uint32_t access_count, float graph_strength);              /* :529-594 */
/* the same for n candidates on the device (signals host arrays [n]); used by the surfacing tail
 * relevance.rs:847-855 when candidate lists are large */
int shodh_fuse_scores_full_batch(int device, const shodh_weights *w, uint64_t n, const float *sem, const float *ent,
                                 const float *tag, const float *imp, const float *momentum_ema,
                                 const uint32_t *access_count, const float *graph_strength, float *out);

/* ---- the ranking tail of RelevanceEngine::surface_relevant_inner (src/relevance.rs:801-918), host code ------ */
/* calculate_tag_score (:680-705): share of the tags that occur in the context (substring, or a context word that starts
 * with the tag / is a prefix of it); both sides lower-cased like str::to_lowercase (full Unicode mappings + Final_Sigma,
 * from the Unicode 13 database), words split at Unicode White_Space. */
float shodh_calculate_tag_score(const char *context_utf8, const char *const *tags_utf8, size_t n_tags);
/* the lower-casing step of it alone (str::to_lowercase, relevance.rs:685-689): writes at most cap - 1 bytes + NUL, returns the full length */
size_t shodh_to_lowercase(const char *utf8, char *out, size_t cap);
/* apply_recency_boost (:1524-1547): age_hours = (now - created_at).num_hours(), cast to u64 like the reference does
 * (a negative age becomes huge and gets no boost) */
float shodh_apply_recency_boost(float base_score, int64_t age_hours, uint64_t boost_hours, float multiplier);
typedef struct {
    float    min_importance;              /* 0.3  (:167-169) */
    uint64_t recency_boost_hours;         /* 24   (:171-173) */
    float    recency_boost_multiplier;    /* 1.2  (:175-177) */
    float    graph_boost_multiplier;      /* 1.15 (:147-149) */
    uint32_t max_results;                 /* 5    (:159-161) */
} shodh_relevance_cfg;
void shodh_relevance_cfg_default(shodh_relevance_cfg *c);
enum { SHODH_REASON_COMBINED = 0, SHODH_REASON_ENTITY_MATCH = 1, SHODH_REASON_SEMANTIC_SIMILARITY = 2, SHODH_REASON_RECENT_IMPORTANT = 3 };
/* Phase 3 of surface_relevant_inner for n candidates: drop importance < min_importance (:809-812), fuse_scores_full
 * (:847-855), recency boost (:869-874), graph boost for entity matches, capped at 1 (:877-881), sort (score total_cmp desc,
 * created_at desc, id asc) (:904-909), keep score >= 0.25 (:913-914), truncate to max_results (:917). The memory_types filter
 * (:815-825) is string matching on the caller's side. uuid: [n][16] (the id tie-break compares Uuid::to_string(), which
 * orders like the bytes). Outputs (max_results entries each): index into the inputs, final score, reason. Returns the count. */
size_t shodh_rank_surfaced(const shodh_weights *w, const shodh_relevance_cfg *cfg, size_t n, const float *semantic, const float *entity,
                           const float *tag, const float *importance, const float *momentum_ema, const uint32_t *access_count,
                           const float *graph_strength, const int64_t *age_hours, const int64_t *created_at_ns, const uint8_t *uuid,
                           uint32_t *out_index, float *out_score, uint8_t *out_reason);

/* ---- allocation guard (diagnostics; shodh_memory_amd/csrc/guard.h) --------------------------------------------------
 * SHODH_GUARD=1|2|3 in the environment (read at the library's first allocation) turns every device / pinned-host allocation into its own
 * mapping fenced by unmapped pages, end-aligned (1: 16 B, 3: 256 B) or start-aligned (2): an out-of-bounds or use-after-free access by any
 * kernel is a GPU page fault on every run. SHODH_GUARD_LOG=<file> logs every allocation for matching the fault address. Off (0): plain
 * hipMalloc / hipHostMalloc. Not part of the reference's interface -- there is no counterpart in src/vector_db. */
int shodh_guard_mode(void);
int shodh_guard_stats(uint64_t *allocations, uint64_t *frees, uint64_t *live_device_bytes);
/* torch.cuda.memory.CUDAPluggableAllocator entry points (tests/conftest.py installs them under SHODH_GUARD so that the callers' device buffers
 * are fenced too); plain hipMalloc / hipFree when the guard is off */
void *shodh_guard_torch_alloc(int64_t bytes, int device, void *stream);
void shodh_guard_torch_free(void *p, int64_t bytes, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHODH_HIP_H */

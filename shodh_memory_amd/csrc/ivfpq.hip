// ivfpq.hip -- SpannIndex (IVF + product quantisation) search / insert / encode on the device,
// bit-faithful to src/vector_db/spann.rs:545-693 and src/vector_db/pq.rs:220-368 GIVEN trained
// state (the reference's k-means is unseeded: spann.rs:472-474, pq.rs:155-157).
//
// Every distance here is a per-candidate sum in a fixed order with no cross-candidate reduction,
// so the device reproduces the reference bit-for-bit by running the same order per thread:
//   probe select  : d_p = 1 - sum_i q_i c_pi (strictly sequential, spann.rs:562-571), the nprobe
//                   smallest by (d total_cmp, p) (spann.rs:595-607)  -> flat_exact kernels, op SEQ
//   ADC table     : T[m][c] = sum_{j<8} (q[8m+j] - cb[m][c][j])^2 sequential (pq.rs:329-351), built
//                   per query straight into LDS (M*ncent*4 = 48 KiB at 384-d)
//   list scan     : d_e = sum_{m<M} T[m][code_e[m]] sequential from 0 (pq.rs:358-368; a code beyond
//                   the table gives f32::MAX), candidates over the probed lists in probe order
//   result        : the k smallest by (d, id) -- identical to the reference's max-heap of size k on
//                   (OrderedFloat(d), id) followed by the (total_cmp, id) sort (spann.rs:625-690),
//                   because d is a finite non-negative sum (no NaN, no -0).
// Roofline: HBM/L2 -- 48-byte codes + 4-byte ids per posting, read coalesced (one entry per lane);
// the LUT gathers hit LDS.
#include <algorithm>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include <hipcub/hipcub.hpp>

#include <vector>

#include "common.h"
#include "topk.h"

#pragma clang fp contract(off)

namespace shodh {

constexpr int EX_OP_SEQ_ONE_MINUS_DOT = 2;
constexpr int EX_OP_SEQ_L2 = 3;

uint32_t topk_capacity(uint32_t k);
size_t exact_partial_bytes(uint32_t nq, uint32_t dim, uint32_t k, uint32_t grid_x);
uint32_t exact_grid_x(uint64_t n_rows, uint32_t nq, uint32_t k, int cus);
int launch_flat_exact(const float *rows, uint64_t n_rows, uint32_t dim, const uint32_t *deleted,
                      const float *d_queries, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                      uint64_t *partial, uint32_t grid_x, uint32_t *d_ids, float *d_dist, uint32_t *d_counts,
                      const uint32_t *qlist, const uint32_t *qcount, hipStream_t st);

struct IvfpqState {
    int device = 0, cus = 256;
    uint32_t dim = 0, P = 0, M = 0, ncent = 0, metric = 0;
    float *centroids = nullptr;      // [P][dim]
    shodh_index *cent_idx = nullptr; // flat index over the centroid table in SHODH_ORDER_SEQ_1M (NormalizedDotProduct only): nearest-centroid
                                     // searches run as MFMA pre-scan + exact re-score instead of the exact-order scan when the table is big enough
    float *codebook = nullptr;       // [M][ncent][8]
    uint64_t *list_off = nullptr;    // [P+1]
    uint32_t *ids = nullptr;         // [total]
    uint8_t *codes = nullptr;        // [total][M]
    uint64_t total = 0, cap_total = 0, max_list_len = 0;
    // host mirror of the postings for incremental insert (SpannIndex::insert appends to a Vec)
    std::vector<std::vector<uint32_t>> h_ids;
    std::vector<std::vector<uint8_t>> h_codes;
    bool dirty = false;
    std::mutex mu;                   // guards the lazy re-upload and the scratch buffers
    unsigned char *scratch = nullptr;
    size_t scratch_bytes = 0;
};

// SpannIndex::len (spann.rs: num_vectors): the postings held, pending inserts included
uint64_t ivfpq_len(IvfpqState *s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> g(s->mu);
    uint64_t n = 0;
    for (const auto &l : s->h_ids) n += l.size();
    return n;
}

void ivfpq_destroy(IvfpqState *s) {
    if (!s) return;
    hipSetDevice(s->device);
    if (s->cent_idx) shodh_index_destroy(s->cent_idx);
    dev_free(s->centroids); dev_free(s->codebook); dev_free(s->list_off); dev_free(s->ids); dev_free(s->codes); dev_free(s->scratch);
    delete s;
}

static int ensure_scratch(IvfpqState *s, size_t bytes) {
    if (bytes <= s->scratch_bytes) return SHODH_OK;
    if (s->scratch) dev_free(s->scratch);
    s->scratch = nullptr; s->scratch_bytes = 0;
    SHODH_HIP_TRY(dev_alloc((void **)&s->scratch, bytes));
    s->scratch_bytes = bytes;
    return SHODH_OK;
}

// ---- nearest centroids through a flat index (SpannIndex::find_nearest_centroid / probe selection, spann.rs:545-571, :595-607) ----
// The k smallest (compute_distance, index) pairs are exactly what a flat search in SHODH_ORDER_SEQ_1M returns: strict '<' in
// the reference keeps the first minimum, i.e. the smallest index among equal distances.
void index_set_probe_set_mode(shodh_index *idx, bool on);      // index.hip
static int make_centroid_index(int device, uint32_t dim, const float *d_centroids, uint32_t P, shodh_index **out) {
    shodh_index_cfg c;
    shodh_index_cfg_default(&c);
    c.dim = dim; c.order = SHODH_ORDER_SEQ_1M; c.device = device; c.reserve_rows = P;
    if (!*out) SHODH_TRY(shodh_index_create(&c, out));
    // only WHICH partitions are probed reaches a result (their postings are merged by (distance, id) afterwards, spann.rs:625-652, :689-690): the centroid
    // index answers with the set of the nprobe nearest, exact distances only where membership hangs on them. SHODH_PROBE_SET=0: the full exact lists.
    static const bool probe_set = !(getenv("SHODH_PROBE_SET") && atoi(getenv("SHODH_PROBE_SET")) == 0);
    index_set_probe_set_mode(*out, probe_set);
    return shodh_index_build_device(*out, d_centroids, P);
}
static int nearest_centroids(shodh_index *ci, const float *d_q, uint64_t nq, uint32_t dim, uint32_t k, uint32_t *d_ids, float *d_dist,
                             uint32_t *d_cnt, hipStream_t st) {
    const uint64_t CH = 8192;             // queries per call: bounds the pre-scan's candidate workspace (32 KiB per query)
    for (uint64_t b = 0; b < nq; b += CH) {
        const uint32_t m = (uint32_t)((nq - b) < CH ? (nq - b) : CH);
        SHODH_TRY(shodh_index_search_device(ci, d_q + b * dim, m, k, d_ids + b * k, d_dist + b * k, d_cnt + b, (void *)st));
    }
    return SHODH_OK;
}

// ---- ADC list scan -------------------------------------------------------------------------------------
struct AdcArgs {
    const float *q;            // [nq][dim]
    const float *codebook;     // [M][ncent][8]
    const uint64_t *list_off;
    const uint32_t *ids;
    const uint8_t *codes;
    const uint32_t *probes;    // [nq][nprobe_k] partitions in probe order (0xFFFFFFFF = none)
    const uint32_t *probe_cnt; // [nq]
    uint32_t nq, dim, M, ncent, nprobe_k, k, cap, split;
    uint64_t *partial;         // [nq][split][k]
    const uint32_t *redo_list; // or null. Not null: the queries to scan are redo_list[0 .. *redo_n) (the list-major path's candidate list overflowed for
    const uint32_t *redo_n;    //   them: this kernel recomputes their result from scratch), workgroup column x takes every gridDim.x-th of them
    const float *tables;       // or null. Not null: [nq][M * ncent] distance tables already built (adc_table_kernel): copied instead of computed
};

constexpr int ADC_U = 1;        // postings per thread and iteration
constexpr int ADC_NT = 1024;    // threads per workgroup: the 48 KiB table allows two workgroups per CU; 2 x 16 waves keep the LDS gathers busy
                                // (measured at 10M rows, batch 1024: 256 threads x 2 postings 2.36 ms, 512 x 2 1.87 ms, 1024 x 1 1.73 ms)
constexpr int ADC_MAXP = 1024;  // probed lists per query (nprobe is capped at P and at this; the reference's default is 20)
// FULL256: 256 codewords per sub-quantiser (always the case for a reference-built index, pq.rs:27): an 8-bit code cannot
// leave the table, so the range check disappears and the table row becomes an immediate offset of the LDS read.
template <bool FULL256, int NT, bool REDO = false>
__global__ __launch_bounds__(NT) void adc_scan_kernel(AdcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *table = reinterpret_cast<float *>(smem);                              // [M][ncent]
    uint64_t *keys = reinterpret_cast<uint64_t *>(table + (size_t)a.M * a.ncent);  // [cap]
    uint64_t *thr = keys + a.cap;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    float *qs = reinterpret_cast<float *>(cnt + 1);                              // [dim]
    uint64_t *seg_base = reinterpret_cast<uint64_t *>((reinterpret_cast<uintptr_t>(qs + a.dim) + 7) & ~(uintptr_t)7);   // [ADC_MAXP] first posting of each segment
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(seg_base + ADC_MAXP);        // [ADC_MAXP + 1] start of each segment in the block's posting space
    const int tid = threadIdx.x;
    const uint32_t part = blockIdx.y;
    // redo mode (the list-major path's overflowed queries): workgroup column x takes entries x, x + gridDim.x, ... of redo_list
    // (a template parameter, not a run-time test: as a run-time loop the compiler hoists the lane's table offsets out of it and the kernel's 50 registers
    // become 95 -- one workgroup per CU instead of two, 1.41 -> 1.73 ms at configs[3])
    const uint32_t n_redo = REDO ? *a.redo_n : 0;
    for (uint32_t qi = blockIdx.x, it = 0; REDO ? qi < n_redo : it == 0; qi += gridDim.x, ++it) {
    const uint32_t q = REDO ? a.redo_list[qi] : qi;
    if (REDO) __syncthreads();                                  // the previous query's LDS is free
    for (uint32_t i = tid; i < a.dim; i += NT) qs[i] = a.q[(size_t)q * a.dim + i];
    if (tid == 0) { *cnt = 0; *thr = a.k ? KEY_NONE : 0; }
    __syncthreads();
    // build_distance_table (pq.rs:329-351): one (m, c) entry per thread-iteration, 8 sequential terms
    if (a.tables) {
        const float *tq = a.tables + (size_t)q * a.M * a.ncent;
        for (uint32_t e = tid; e < a.M * a.ncent; e += NT) table[e] = tq[e];
    } else
    for (uint32_t e = tid; e < a.M * a.ncent; e += NT) {
        const uint32_t m = e / a.ncent;
        const float *cb = a.codebook + (size_t)e * 8;
        const float *qq = qs + m * 8;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = qq[j] - cb[j]; sum = sum + d * d; }
        table[e] = sum;
    }
    __syncthreads();
    TopKBuf buf{keys, cnt, thr, a.cap, a.k};
    const uint32_t np = a.probe_cnt[q] < ADC_MAXP ? a.probe_cnt[q] : (uint32_t)ADC_MAXP;
    // This block's share of every probed list, [lo + part*len/split, lo + (part+1)*len/split), laid end to end as one
    // posting space [0, total): segment starts in LDS, a posting index is mapped back by binary search. Every iteration
    // is then full (ADC_U postings per thread) whatever the list lengths are, and the codes of the NEXT iteration are
    // requested before the current one is scored (the loads used to be issued and awaited inside every 256-posting
    // iteration: 122 exposed round trips per block at 4M rows). Scoring order does not matter: the result is the k
    // smallest (distance, id) keys and every distance is its own fixed-order sum.
    for (uint32_t i = tid; i < np; i += NT) {
        const uint32_t p = a.probes[(size_t)q * a.nprobe_k + i];
        const uint64_t lo = a.list_off[p], len = a.list_off[p + 1] - lo;
        const uint64_t b0 = lo + len * part / a.split, b1 = lo + len * (part + 1) / a.split;
        seg_base[i] = b0;
        seg_start[i] = (uint32_t)(b1 - b0);         // length for now; scanned below
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < np; ++i) { const uint32_t l = seg_start[i]; seg_start[i] = acc; acc += l; }
        seg_start[np] = acc;
    }
    __syncthreads();
    const uint32_t total = seg_start[np];
    auto locate = [&](uint32_t g) -> uint64_t {      // posting g of the block's space -> index into codes / ids
        uint32_t lo_ = 0, hi_ = np;                   // largest i with seg_start[i] <= g
        while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (seg_start[mid] <= g) lo_ = mid; else hi_ = mid; }
        return seg_base[lo_] + (g - seg_start[lo_]);
    };
    const bool wide = (a.M == 48);                   // 384-d: 3 x 16-byte code loads per posting
    uint4 cw[ADC_U][3];
    uint32_t idv[ADC_U];
    uint64_t ev[ADC_U];
    auto fetch = [&](uint32_t g0) {
#pragma unroll
        for (int u = 0; u < ADC_U; ++u) {
            const uint32_t g = g0 + u * NT + tid;
            const uint64_t e = locate(g < total ? g : (total ? total - 1 : 0));      // clamped: unconditional loads
            ev[u] = e;
            if (wide) {
                const uint4 *cp = reinterpret_cast<const uint4 *>(a.codes + e * 48);
                cw[u][0] = cp[0]; cw[u][1] = cp[1]; cw[u][2] = cp[2];
            }
            idv[u] = a.ids[e];
        }
    };
    if (total) fetch(0);
    for (uint32_t g0 = 0; g0 < total; g0 += NT * ADC_U) {
        uint4 cc[ADC_U][3];
        uint32_t idc[ADC_U];
        uint64_t ec[ADC_U];
#pragma unroll
        for (int u = 0; u < ADC_U; ++u) { cc[u][0] = cw[u][0]; cc[u][1] = cw[u][1]; cc[u][2] = cw[u][2]; idc[u] = idv[u]; ec[u] = ev[u]; }
        if (g0 + NT * ADC_U < total) fetch(g0 + NT * ADC_U);
#pragma unroll
        for (int u = 0; u < ADC_U; ++u) {
            const uint32_t g = g0 + u * NT + tid;
            if (g < total) {
                float totald = 0.0f;
                bool bad = false;
                // distance_with_table (pq.rs:358-368): strictly m = 0, 1, ... ; a code beyond the table -> f32::MAX
                if (wide) {
                    // all 16 table reads of a code word are issued before the first add (a read behind the `bad` branch was
                    // one exposed LDS round trip per sub-quantiser: 48 per posting); the adds stay strictly in m order
                    const uint32_t cmax = a.ncent - 1;
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        const uint32_t word[4] = {cc[u][w].x, cc[u][w].y, cc[u][w].z, cc[u][w].w};
                        float tv[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const uint32_t c = (word[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                            if (FULL256) {
                                tv[j] = table[(w * 16 + j) * 256 + c];
                            } else {
                                bad |= c > cmax;
                                tv[j] = table[(w * 16 + j) * a.ncent + (c > cmax ? cmax : c)];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j) totald = totald + tv[j];
                    }
                } else {
                    const uint8_t *code = a.codes + ec[u] * a.M;
                    for (uint32_t m = 0; m < a.M; ++m) {
                        const uint32_t c = code[m];
                        if (c >= a.ncent) { bad = true; break; }
                        totald = totald + table[m * a.ncent + c];
                    }
                }
                if (bad) totald = 3.4028234663852886e38f;       // f32::MAX
                topk_push(buf, make_key(totald, idc[u]));
            }
        }
        topk_compact_if_short<NT>(buf, NT * ADC_U);
    }
    __syncthreads();
    topk_compact<NT>(buf);
    uint64_t *out = a.partial + ((size_t)q * a.split + part) * a.k;
    const uint32_t m = *buf.cnt;
    for (uint32_t i = tid; i < a.k; i += NT) out[i] = (i < m) ? buf.keys[i] : KEY_NONE;
    }
}

struct AdcMergeArgs {
    const uint64_t *partial;   // [nq][split][k]
    uint32_t split, k, cap;
    uint32_t *ids; float *dist; uint32_t *counts;
};
__global__ __launch_bounds__(256) void adc_merge_kernel(AdcMergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *mins = keys + a.cap;
    uint64_t *thr = mins + 512;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    TopKBuf buf{keys, cnt, thr, a.cap, a.k};
    const uint64_t *base = a.partial + (size_t)q * a.split * a.k;
    auto key_at = [&](uint64_t i) -> uint64_t { return base[i]; };
    const uint32_t m = block_select_topk<256>(key_at, (uint64_t)a.split * a.k, buf, mins);
    for (uint32_t i = tid; i < a.k; i += 256) {
        if (i < m) {
            const uint64_t key = buf.keys[i];
            a.ids[(size_t)q * a.k + i] = (uint32_t)key;
            a.dist[(size_t)q * a.k + i] = order_key_inv((uint32_t)(key >> 32));
        } else {
            a.ids[(size_t)q * a.k + i] = 0xFFFFFFFFu;
            a.dist[(size_t)q * a.k + i] = __builtin_inff();
        }
    }
    if (tid == 0) a.counts[q] = m;
}

// ---- list-major batch scan (round 4) ------------------------------------------------------------------------------
// A batch of queries probes the same lists over and over (configs[3]: 1024 queries x 32 probes over 4096 lists = 8 queries per list),
// and the query-major kernel above is bound by its LDS gathers, not by HBM: a 64-lane ds_read_b32 with random 8-bit codes takes ~6.8
// cycles (bank conflicts; tools/ubench/lds_gather.hip: 5.8 T lookups/s over the chip), two thirds of the kernel's cycles. Here:
//   1. a bound B(q) on every query's k-th best key: the k-th best key among the first <= 3072 postings of its nearest lists (adc_bound_kernel:
//      one query per workgroup against its own table). The nearest list holds most of the final answer, so the bound is close to the final one;
//   2. ALL (query, probe) pairs are grouped by LIST. A workgroup takes a chunk of <= LM_NT * LM_U postings of one list and a block of the
//      queries that probe it: the chunk's codes and ids are read ONCE into registers and scored against every query of the block (HBM: the
//      probed postings once per block instead of once per pair). The distance tables (T[m][c], 48 KiB per query, built once per query by
//      adc_table_kernel) of TWO queries are interleaved entry by entry in LDS, so one ds_read_b64 gather serves two lookups at the
//      instruction rate of one (the same benchmark: 11.3 T lookups/s); they arrive in 24-sub-quantiser tiles (48 KiB) staged through
//      registers one tile ahead. Every distance is still its own sum in m order from 0 (pq.rs:358-368): the same bits. A key at or under
//      B(q) is appended to the query's candidate list in HBM (a hundred or two per query); nothing is selected inside this kernel;
//   3. a query whose candidate list overflowed (an outlier whose nearest lists say little about the others) is recomputed from scratch by the
//      query-major kernel (SHODH_ADC_LM_CAP forces that in the tests); lm_merge_kernel picks the k best of the candidates.
// Measured and dropped on the way: the k best of every (query, list) pair selected inside the list kernel (group minima, bound, gather, rank, two
// more barriers per pair: the selection cost more than the scoring, 1.80 ms against the query-major kernel's 1.31); the nearest lists scanned WHOLE
// by the bound kernel and left out of the list-major pass (the nearest list is the long one -- 4 500 postings on average where the mean list has
// 2 441 -- and one query per workgroup gathers at half the rate: 110 us where a 3 072-posting prefix takes 30).
typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef float f32x2q __attribute__((ext_vector_type(2)));
#ifndef SHODH_LM_U      // (diagnostic builds time other shapes)
#define SHODH_LM_U 3
#endif
#ifndef SHODH_LM_QB
#define SHODH_LM_QB 4
#endif
#ifndef SHODH_LM_PRUNE
#define SHODH_LM_PRUNE 1
#endif
constexpr int LM_NT = 1024, LM_U = SHODH_LM_U, LM_QB = SHODH_LM_QB, LM_MT = 24;
constexpr bool LM_PRUNE = SHODH_LM_PRUNE;      // partial-sum pruning (see adc_list_kernel)
constexpr int LM_TILE = LM_MT * 256 * 8;                       // 48 KiB: 24 sub-quantisers x 256 entries x 2 queries
constexpr int LM_CANDS = 4096;                                  // candidate keys per query (HBM). (Round 6, 32768 instead: the one query in a thousand with a loose bound --
                                                                // 29 727 candidates at configs[3] -- then costs the merge kernel 84 us where its redo costs 45 + 10.)
#ifndef SHODH_LM_WG_PER_CU
#define SHODH_LM_WG_PER_CU 1
#endif
constexpr int LM_WG_PER_CU = SHODH_LM_WG_PER_CU;               // resident workgroups of the list scan per CU (its LDS admits one)
// A work item of the list-major scan = (a chunk of <= LM_NT * LM_U postings of one list) x (a block of <= 2 * LM_QB of the queries that probe it), described
// in one 128-byte record written by lm_fill_kernel. The scan reads the record with one 32-lane load.
struct LmItem { uint64_t c0; uint32_t clen, pqn; uint32_t q[2 * LM_QB]; uint32_t pad[12 - 2 * LM_QB]; uint64_t bound[2 * LM_QB]; uint64_t pad2[8 - 2 * LM_QB]; };
static_assert(sizeof(LmItem) == 128 && LM_QB <= 4, "one descriptor = 32 dwords");
constexpr int LM_LDS = 2 * LM_TILE + 2 * (int)sizeof(LmItem) + 16;

constexpr int LM_TQ = 8;      // queries per workgroup of the table kernel (a codebook row is loaded once for all of them)
__global__ __launch_bounds__(1024) void adc_table_kernel(const float *__restrict__ q, const float *__restrict__ codebook, uint32_t dim, uint32_t nq, float *__restrict__ tables, uint32_t *__restrict__ zero, uint32_t n_zero) {
    // (also clears the pass's counters: the first kernel of the pass, a memset would be two launches of its own)
    { const uint32_t z = (blockIdx.y * gridDim.x + blockIdx.x) * 1024 + threadIdx.x; if (z < n_zero) zero[z] = 0; }
    // build_distance_table (pq.rs:329-351) for queries blockIdx.x * LM_TQ ..: entry e = m * 256 + c, 8 sequential terms (the arithmetic of adc_scan_kernel's table)
    __shared__ float qs[LM_TQ][32];                              // the four sub-vectors (m = 4 blockIdx.y ..) of each query
    const uint32_t e = blockIdx.y * 1024 + threadIdx.x, m = e >> 8, q0 = blockIdx.x * LM_TQ;
    if (threadIdx.x < LM_TQ * 32) {
        const uint32_t qi = threadIdx.x >> 5, j = threadIdx.x & 31;
        qs[qi][j] = (q0 + qi < nq) ? q[(size_t)(q0 + qi) * dim + blockIdx.y * 32 + j] : 0.0f;
    }
    float cb[8];
    {
        const f32x4q *cp = reinterpret_cast<const f32x4q *>(codebook + (size_t)e * 8);
        const f32x4q c0 = cp[0], c1 = cp[1];
        cb[0] = c0[0]; cb[1] = c0[1]; cb[2] = c0[2]; cb[3] = c0[3]; cb[4] = c1[0]; cb[5] = c1[1]; cb[6] = c1[2]; cb[7] = c1[3];
    }
    __syncthreads();
    const float *qq0 = &qs[0][(m & 3) * 8];
#pragma unroll
    for (int qi = 0; qi < LM_TQ; ++qi) {
        if (q0 + qi < nq) {
            const float *qq = qq0 + qi * 32;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = qq[j] - cb[j]; sum = sum + d * d; }
            tables[(size_t)(q0 + qi) * 12288 + e] = sum;
        }
    }
}

// pairs (query, probed list) grouped by list: counted by adc_bound_kernel; scan (pair and work-item offsets of every list), fill (the work items' records)
__global__ __launch_bounds__(1024) void lm_scan_kernel(const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ list_off, uint32_t P, uint32_t *__restrict__ pair_off, uint32_t *__restrict__ cursor, uint32_t *__restrict__ item_off) {
    // exclusive scans over the lists: pairs, and work items (a block of <= 2 * LM_QB queries of one list x a chunk of <= LM_NT * LM_U of its postings)
    const uint32_t tid = threadIdx.x, per = (P + 1023) / 1024, b0 = tid * per < P ? tid * per : P, b1 = (b0 + per < P) ? b0 + per : P;
    auto chunks_of = [&](uint32_t p) -> uint32_t { return (uint32_t)((list_off[p + 1] - list_off[p] + (uint64_t)LM_NT * LM_U - 1) / ((uint64_t)LM_NT * LM_U)); };
    auto items_of = [&](uint32_t p) -> uint32_t { return ((cnt[p] + 2 * LM_QB - 1) / (2 * LM_QB)) * chunks_of(p); };
    uint32_t a = 0, it = 0;
    for (uint32_t p = b0; p < b1; ++p) { a += cnt[p]; it += items_of(p); }
    typedef hipcub::BlockScan<uint32_t, 1024> Scan;
    __shared__ typename Scan::TempStorage tmp;
    uint32_t ta, ti;
    Scan(tmp).ExclusiveSum(a, a, ta);
    __syncthreads();
    Scan(tmp).ExclusiveSum(it, it, ti);
    if (tid == 0) { pair_off[P] = ta; item_off[P] = ti; }
    for (uint32_t p = b0; p < b1; ++p) { pair_off[p] = a; cursor[p] = a; item_off[p] = it; a += cnt[p]; it += items_of(p); }
}
__global__ void lm_fill_kernel(const uint32_t *__restrict__ probes, const uint32_t *__restrict__ probe_cnt, uint32_t nq, uint32_t nprobe, uint32_t P, const uint64_t *__restrict__ list_off,
                               const uint32_t *__restrict__ pair_off, const uint32_t *__restrict__ item_off, const uint64_t *__restrict__ bound, uint32_t *__restrict__ cursor, LmItem *__restrict__ items) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * nprobe) return;
    const uint32_t q = i / nprobe, r = i - q * nprobe;
    if (r >= probe_cnt[q]) return;
    const uint32_t p = probes[i];
    if (p >= P) return;
    // the query's place among the list's pairs (the order inside a list is arbitrary: every query's result is computed on its own) = a slot of one block:
    // into the record of every chunk of that block
    const uint32_t slot = atomicAdd(&cursor[p], 1u) - pair_off[p], blk = slot / (2 * LM_QB), sl = slot - blk * 2 * LM_QB;
    const uint64_t lo = list_off[p], len = list_off[p + 1] - lo;
    const uint32_t nchunk = (uint32_t)((len + (uint64_t)LM_NT * LM_U - 1) / ((uint64_t)LM_NT * LM_U));
    const uint64_t b = bound[q];
    LmItem *d = items + item_off[p] + blk * nchunk;
    for (uint32_t ch = 0; ch < nchunk; ++ch) { d[ch].q[sl] = q; d[ch].bound[sl] = b; }
    if (sl == 0) {                                                // the block's first query also describes its chunks
        const uint32_t left = pair_off[p + 1] - pair_off[p] - blk * 2 * LM_QB, pqn = left < 2u * LM_QB ? left : 2u * LM_QB;
        for (uint32_t ch = 0; ch < nchunk; ++ch) {
            const uint64_t c0 = lo + len * ch / nchunk;
            d[ch].c0 = c0; d[ch].clen = (uint32_t)(lo + len * (ch + 1) / nchunk - c0); d[ch].pqn = pqn;
        }
    }
}

#ifdef SHODH_LMPROF      // diagnostic build: phase timers of a few workgroups
#define LPROF_DECL long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq_ = clock64(); const long long t00_ = tq_;
#define LPROF_T(i) { const long long t_ = clock64(); pt_[i] += t_ - tq_; tq_ = t_; }
#define NPROF_END if (tid == 0 && (blockIdx.x % 257) == 3) printf("boundprof q %d lists %u postings %u total %lld : table+walk %lld | codes+score %lld | bound %lld | push %lld | sort %lld | tail %lld\n", (int)blockIdx.x, r0, total, clock64() - t00_, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5]);
#define LPROF_END if (lane == 0 && (wave == 0 || wave == 9) && (blockIdx.x % 61) == 5) printf("lmprof blk %d wave %d total %lld : codes %lld | wait-first-tile %lld | score %lld | store %lld | bar %lld | push %lld\n", (int)blockIdx.x, wave, clock64() - t00_, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5]);
#else
#define LPROF_DECL
#define LPROF_T(i)
#define LPROF_END
#define NPROF_END
#endif
// The bound: the first <= max_postings (<= LM_NEAR_NT * LM_NEAR_U) postings of the query's nearest lists -- the nearest list's head, or several short
// lists -- scored against its own table (plain ds_read_b32 gathers: one query per workgroup, a few per cent of the postings), keys in registers (all
// code loads of a lane in flight at once). The k-th smallest of G >= k group minima bounds the segment's k-th best key; the few keys at or under it go
// through the TopKBuf and are ranked by counting: bound[q] = the k-th best key seen (KEY_NONE: fewer than k postings seen, no bound). The fewer postings
// stand behind the bound, the more keys of the other lists pass it (about k * probed postings / postings behind the bound on featureless data).
// The kernel also counts the query's pairs per list for the list-major pass (every probed list: the postings scored here are scored again there).
#ifndef SHODH_LM_NEAR_U      // (diagnostic builds: postings per lane behind the bound)
#define SHODH_LM_NEAR_U 6
#endif
constexpr int LM_NEAR_NT = 512, LM_NEAR_U = SHODH_LM_NEAR_U, LM_NEAR_MAXL = 64;      // (75 KiB of LDS at k = 10: two workgroups per CU)
struct NearArgs {
    const float *tables; const uint64_t *list_off; const uint32_t *ids; const uint8_t *codes;
    const uint32_t *probes; const uint32_t *probe_cnt;
    uint32_t nprobe_k, k, cap, P, max_postings;
    uint64_t *bound; uint32_t *list_cnt;
};
__global__ __launch_bounds__(LM_NEAR_NT) void adc_bound_kernel(const NearArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *table = reinterpret_cast<float *>(smem);                                   // [48][256]
    uint64_t *bkeys = reinterpret_cast<uint64_t *>(smem + 49152);                      // [cap] TopKBuf (cap >= k + 2 LM_NEAR_NT)
    uint64_t *mins = bkeys + a.cap;                                                    // [2 LM_NEAR_NT] group minima; rank-sort output
    uint64_t *seg_base = mins + 2 * LM_NEAR_NT;                                        // [LM_NEAR_MAXL]
    uint64_t *thr = seg_base + LM_NEAR_MAXL;
    uint32_t *seg_start = reinterpret_cast<uint32_t *>(thr + 1);                       // [LM_NEAR_MAXL + 1]
    uint32_t *cnt = seg_start + LM_NEAR_MAXL + 1;
    uint32_t *nl = cnt + 1;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x, k = a.k;
    LPROF_DECL
    {
        const f32x4q *tq = reinterpret_cast<const f32x4q *>(a.tables + (size_t)q * 12288);
#pragma unroll
        for (int i = 0; i < 3072 / LM_NEAR_NT; ++i) reinterpret_cast<f32x4q *>(table)[i * LM_NEAR_NT + tid] = tq[i * LM_NEAR_NT + tid];
    }
    const uint32_t npq = a.probe_cnt[q];
    uint32_t np = npq < (uint32_t)LM_NEAR_MAXL ? npq : (uint32_t)LM_NEAR_MAXL;
    // the lengths of the nearest lists (one lane each), then lane 0 lays the first of them end to end until max_postings are covered
    if ((uint32_t)tid < np) {
        const uint32_t p = a.probes[(size_t)q * a.nprobe_k + tid];
        uint64_t lo = 0, len = 0;
        if (p < a.P) { lo = a.list_off[p]; len = a.list_off[p + 1] - lo; }
        seg_base[tid] = lo; seg_start[tid] = (uint32_t)(len < 0xFFFFFFFFull ? len : 0xFFFFFFFFull);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0, r = 0;
        while (r < np && acc < a.max_postings) { const uint32_t len = seg_start[r]; seg_start[r] = acc; acc = (len > a.max_postings - acc) ? a.max_postings : acc + len; ++r; }
        seg_start[r] = acc; *nl = r;
    }
    __syncthreads();
    const uint32_t r0 = *nl, total = seg_start[r0], n = total;                        // n <= max_postings <= LM_NEAR_NT * LM_NEAR_U
    LPROF_T(0)
    TopKBuf buf{bkeys, cnt, thr, a.cap, k};
    uint64_t bound = KEY_NONE;
    if (n) {
        uint4 c3[LM_NEAR_U][3];
        uint32_t idv[LM_NEAR_U];
        uint64_t rk[LM_NEAR_U];
#pragma unroll
        for (int u = 0; u < LM_NEAR_U; ++u) {
            const uint32_t i = u * LM_NEAR_NT + tid;
            const uint32_t g = i < n ? i : n - 1;
            uint32_t lo_ = 0, hi_ = r0;                 // largest r with seg_start[r] <= g
            while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (seg_start[mid] <= g) lo_ = mid; else hi_ = mid; }
            const uint64_t e = seg_base[lo_] + (g - seg_start[lo_]);
            const uint4 *cp = reinterpret_cast<const uint4 *>(a.codes + e * 48);
            c3[u][0] = cp[0]; c3[u][1] = cp[1]; c3[u][2] = cp[2];
            idv[u] = a.ids[e];
        }
        uint64_t tmin = KEY_NONE;
#pragma unroll
        for (int u = 0; u < LM_NEAR_U; ++u) {
            float d = 0.0f;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const uint32_t word[4] = {c3[u][w].x, c3[u][w].y, c3[u][w].z, c3[u][w].w};
                float tv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) tv[j] = table[(w * 16 + j) * 256 + ((word[j >> 2] >> ((j & 3) * 8)) & 0xFFu)];
#pragma unroll
                for (int j = 0; j < 16; ++j) d = d + tv[j];                    // strictly in m order (pq.rs:358-368)
            }
            rk[u] = (u * LM_NEAR_NT + tid < n) ? make_key(d, idv[u]) : KEY_NONE;
            tmin = rk[u] < tmin ? rk[u] : tmin;
        }
        LPROF_T(1)
        // the k-th smallest of G >= k group minima (G = 16 .. 512 groups of 32 .. 1 lanes) bounds the k-th best key from above. (A rank over all the lane
        // minima -- block_select_topk's form -- costs thousands of broadcast reads; 16 group minima cost nothing and let ~1.5 k keys through instead of k.)
        uint64_t T = KEY_NONE;
        if (k <= (uint32_t)LM_NEAR_NT) {
            uint32_t G = 16; while (G < k) G <<= 1;
            const uint32_t gw = LM_NEAR_NT / G;         // lanes per group (32 .. 1)
            uint64_t gm = tmin;
            for (uint32_t ofs = 1; ofs < gw; ofs <<= 1) {
                const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(gm >> 32), (int)ofs) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)gm, (int)ofs);
                gm = o < gm ? o : gm;
            }
            if ((tid & (gw - 1)) == 0) mins[tid / gw] = gm;
            if (tid == 0) *thr = KEY_NONE;
            __syncthreads();
            if ((uint32_t)tid < G) {
                const uint64_t v = mins[tid];
                uint32_t r = 0;
                for (uint32_t j = 0; j < G; ++j) { const uint64_t w = mins[j]; r += (w < v) || (w == v && j < (uint32_t)tid); }
                if (r == k - 1) *thr = v;               // exactly one group holds rank k - 1 (KEY_NONE when fewer than k groups have a key)
            }
            __syncthreads();
            T = *thr;
            __syncthreads();
        }
        if (tid == 0) { *cnt = 0; *thr = (T == KEY_NONE) ? KEY_NONE : T + 1; }
        __syncthreads();
        LPROF_T(2)
#pragma unroll
        for (int u = 0; u < LM_NEAR_U; ++u) {
            if (rk[u] != KEY_NONE) topk_push(buf, rk[u]);
            if ((u & 1) == 1) topk_compact_if_short<LM_NEAR_NT>(buf, 2 * LM_NEAR_NT);      // at most 2 LM_NEAR_NT pushes between checks
        }
        const uint32_t left = *buf.cnt;
        __syncthreads();
        LPROF_T(3)
        if (left <= 2u * LM_NEAR_NT) {                  // the usual case (a few dozen keys): order them by counting, no sorting network (21 barriers for 64 keys)
            rank_sort_lds<LM_NEAR_NT>(bkeys, mins, left);
            if (left >= k) bound = mins[k - 1];
        } else {
            topk_compact<LM_NEAR_NT>(buf);
            if (*buf.cnt >= k) bound = bkeys[k - 1];
        }
        LPROF_T(4)
    }
    if (tid == 0) a.bound[q] = bound;
    // the query's probed lists: one more pair for each of them (the list-major pass groups the pairs by list)
    for (uint32_t r = tid; r < npq; r += LM_NEAR_NT) { const uint32_t p = a.probes[(size_t)q * a.nprobe_k + r]; if (p < a.P) atomicAdd(&a.list_cnt[p], 1u); }
    LPROF_T(5)
    NPROF_END
}

struct LmArgs {
    const float *tables;       // [nq][48 * 256]
    const uint32_t *ids;
    const uint8_t *codes;
    const LmItem *items;       // the work items (lm_scan_kernel, lm_fill_kernel)
    const uint32_t *n_items;   // their number
    uint32_t *work;            // the next item nobody has taken yet, less the grid size (zero at launch)
    uint32_t cand_cap;
    uint64_t *cand;            // [nq][cand_cap] keys under the bound
    uint32_t *cand_cnt;        // [nq] (may exceed cand_cap: the query is redone)
};

// Persistent (round 6): one workgroup per CU walks the work items -- the first by its index, the following ones from a shared counter -- and never waits for a
// descriptor: wave 0 asks for the item after the next (an atomic) and for the next item's record (one 128-byte load) at the top of an item and hands the record
// to the others through LDS during the item's last query pair, whose second half then stages the NEXT item's first table tile exactly as it would stage the
// next pair's. Between two items only the chunk's codes are waited for. (One workgroup per item, round 4 / 5: item index -> list -> pair offsets -> queries ->
// bounds -> tables was a chain of five dependent loads, 9 000 cycles with nothing to overlap them, a quarter of an average item.)
__global__ __launch_bounds__(LM_NT) void adc_list_kernel(const LmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LmItem *ctl = reinterpret_cast<LmItem *>(smem + 2 * LM_TILE);                       // [2] this item's record, the next one's
    uint32_t *has_next = reinterpret_cast<uint32_t *>(ctl + 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_items = *a.n_items;
    if (blockIdx.x >= n_items) return;
    LPROF_DECL
    uint32_t nxt = 0, dreg = 0;                                                          // wave 0: the next item, dword `lane` of its record
    // (the counter's offset passes through a VGPR the compiler knows nothing about: with a provably uniform address its atomic optimizer turns the add into a
    // wave-wide reduction that reads the result back at once -- the one wait this kernel is built to avoid)
    uint32_t wofs = 0;
    asm volatile("" : "+v"(wofs));
    uint32_t *work = a.work + wofs;
    if (wave == 0) {
        if (lane < 32) reinterpret_cast<uint32_t *>(&ctl[0])[lane] = reinterpret_cast<const uint32_t *>(a.items + blockIdx.x)[lane];
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(work, 1u);
        nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + gridDim.x;
    }
    __syncthreads();
    f32x4q sa[2], sb[2];
    // tile (record d, pair j, half t) -> registers: 1536 groups of four entries per query; thread i stages group i and, in waves 0-7, group 1024 + i
    // (an odd block repeats its last query: scored, not pushed twice)
    auto tile_load = [&](const LmItem *d, uint32_t j, int t) {
        const uint32_t last = d->pqn - 1, qa = d->q[2 * j], qb = d->q[2 * j + 1 < last ? 2 * j + 1 : last];
        const float *ta = a.tables + (size_t)qa * 12288 + t * 6144, *tb = a.tables + (size_t)qb * 12288 + t * 6144;
        sa[0] = *reinterpret_cast<const f32x4q *>(ta + tid * 4); sb[0] = *reinterpret_cast<const f32x4q *>(tb + tid * 4);
        if (wave < 8) { sa[1] = *reinterpret_cast<const f32x4q *>(ta + 4096 + tid * 4); sb[1] = *reinterpret_cast<const f32x4q *>(tb + 4096 + tid * 4); }
    };
    auto tile_store = [&](int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0 || wave < 8) {
                f32x4q w0, w1;
                w0[0] = sa[h][0]; w0[1] = sb[h][0]; w0[2] = sa[h][1]; w0[3] = sb[h][1];
                w1[0] = sa[h][2]; w1[1] = sb[h][2]; w1[2] = sa[h][3]; w1[3] = sb[h][3];
                unsigned char *d = smem + slot * LM_TILE + (h * 1024 + tid) * 32;
                *reinterpret_cast<f32x4q *>(d) = w0; *reinterpret_cast<f32x4q *>(d + 16) = w1;
            }
        }
    };
    // The chunk's codes (48 bytes per posting, LM_U postings per lane) stay in registers for the whole item. Words 0-5 (sub-quantisers 0-23) are last read in the
    // first half of the item's last pair, words 6-11 in its second half: the NEXT item's words 0-5 are requested during that second half, its words 6-11 and ids at
    // its top (first needed half a pair later) -- the same registers, and no item waits for its codes (round 6: that wait was a quarter of the kernel).
    uint32_t cw[LM_U][12];
    auto codes_lo = [&](const LmItem *d) {
        const uint64_t c0 = d->c0; const uint32_t clen = d->clen;
#pragma unroll
        for (int u = 0; u < LM_U; ++u) {                                                   // (a lane past the chunk's end reads the last posting again: never scored)
            const uint32_t g = u * LM_NT + tid;
            const uint64_t e = c0 + (g < clen ? g : clen - 1);
            const uint4 v = *reinterpret_cast<const uint4 *>(a.codes + e * 48);
            const uint2 w = *reinterpret_cast<const uint2 *>(a.codes + e * 48 + 16);
            cw[u][0] = v.x; cw[u][1] = v.y; cw[u][2] = v.z; cw[u][3] = v.w; cw[u][4] = w.x; cw[u][5] = w.y;
        }
    };
    tile_load(&ctl[0], 0, 0);
    tile_store(0);                                                                       // (visible after the barrier at the first item's top)
#pragma unroll
    for (int u = 0; u < LM_U; ++u)
#pragma unroll
        for (int w = 0; w < 12; ++w) cw[u][w] = 0;
    codes_lo(&ctl[0]);
    int s = 0;
    bool first = true;
    for (;;) {
        const LmItem *it = &ctl[s];
        const uint64_t c0 = it->c0;
        const uint32_t clen = it->clen, pqn = it->pqn, npair = (pqn + 1) >> 1;           // clen <= LM_NT * LM_U, pqn >= 1
        uint32_t t2 = 0;
        if (wave == 0) {                                                                  // (unconditional loads: a load under a branch is waited for where the branch ends)
            dreg = reinterpret_cast<const uint32_t *>(a.items + (nxt < n_items ? nxt : n_items - 1))[lane & 31];
            if (lane == 0) t2 = atomicAdd(work, 1u);
        }
        uint32_t idv[LM_U];
        bool valid[LM_U];
#pragma unroll
        for (int u = 0; u < LM_U; ++u) {
            const uint32_t g = u * LM_NT + tid;
            valid[u] = g < clen;
            const uint64_t e = c0 + (valid[u] ? g : clen - 1);
            const uint2 w = *reinterpret_cast<const uint2 *>(a.codes + e * 48 + 24);
            const uint4 v = *reinterpret_cast<const uint4 *>(a.codes + e * 48 + 32);
            cw[u][6] = w.x; cw[u][7] = w.y; cw[u][8] = v.x; cw[u][9] = v.y; cw[u][10] = v.z; cw[u][11] = v.w;
            idv[u] = a.ids[e];
        }
        LPROF_T(0)
        if (first) { __syncthreads(); first = false; }                                    // the first item's first tile (the others': staged before the last barrier of the item before)
        LPROF_T(1)
        bool more_items = false;
        for (uint32_t j = 0; j < npair; ++j) {
            const uint32_t ia = 2 * j, ib = 2 * j + 1 < pqn ? 2 * j + 1 : pqn - 1;
            const uint32_t qa = it->q[ia], qb = it->q[ib];
            // keys at or under these bounds are candidates; a posting whose PARTIAL sums already lie above both (every table entry is a sum of squares: the
            // sums only grow) can never become one, and a wave whose 64 postings are all in that state skips the rest of their lookups
            const uint64_t ba = it->bound[ia], bb = it->bound[ib];
            const uint32_t da = (uint32_t)(ba >> 32), db = (uint32_t)(bb >> 32);
            const bool two = qb != qa;
            const bool lastpair = j + 1 == npair;
            f32x2q sum[LM_U];
#pragma unroll
            for (int u = 0; u < LM_U; ++u) { sum[u][0] = 0.0f; sum[u][1] = 0.0f; }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bool more = true;
                if (t == 0) {
                    tile_load(it, j, 1);
                } else if (!lastpair) {
                    tile_load(it, j + 1, 0);
                } else {
                    more_items = *has_next != 0;
                    more = more_items;
                    if (more) { tile_load(&ctl[s ^ 1], 0, 0); codes_lo(&ctl[s ^ 1]); }
                }
                const unsigned char *tb = smem + t * LM_TILE;
#pragma unroll
                for (int u = 0; u < LM_U; ++u) {
                    if ((uint32_t)(u * LM_NT + wave * 64) < clen) {              // (wave-uniform: a wave whose lanes all lie past the chunk's end skips the pass)
                        uint32_t word[6] = {cw[u][6 * t], cw[u][6 * t + 1], cw[u][6 * t + 2], cw[u][6 * t + 3], cw[u][6 * t + 4], cw[u][6 * t + 5]};
                        // (the codes do not change from query pair to query pair: without this the compiler hoists all 144 table offsets of a lane out of the pair loop and spills them)
                        asm volatile("" : "+v"(word[0]), "+v"(word[1]), "+v"(word[2]), "+v"(word[3]), "+v"(word[4]), "+v"(word[5]));
#pragma unroll
                        for (int h = 0; h < 3; ++h) {                        // eight gathers in flight, then their adds strictly in m order (pq.rs:358-368)
                            // (from the 16th sub-quantiser on: a lane whose partial sums lie above both bounds stops looking up -- its sums stay partial, above the
                            // bounds, and are never pushed; fewer active lanes are also fewer bank conflicts for the others)
                            const bool alive = !LM_PRUNE || !(t == 1 || h == 2) || (valid[u] && (order_key(sum[u][0]) <= da || order_key(sum[u][1]) <= db));
                            if (alive) {
                                f32x2q tv[8];
#pragma unroll
                                for (int jj = 0; jj < 8; ++jj) {
                                    const uint32_t c = (word[2 * h + (jj >> 2)] >> ((jj & 3) * 8)) & 0xFFu;
                                    tv[jj] = *reinterpret_cast<const f32x2q *>(tb + (8 * h + jj) * 2048 + c * 8);
                                }
#pragma unroll
                                for (int jj = 0; jj < 8; ++jj) sum[u] = sum[u] + tv[jj];
                            }
                        }
                    }
                }
                LPROF_T(2)
                if (more) tile_store(t ^ 1);
                if (t == 0 && lastpair && wave == 0) {                                   // the next item's record: visible to everybody after this half's barrier
                    if (lane < 32) reinterpret_cast<uint32_t *>(&ctl[s ^ 1])[lane] = dreg;    // (here, where the tile's loads are waited for anyway)
                    if (lane == 0) *has_next = nxt < n_items ? 1u : 0u;
                }
                LPROF_T(3)
                __syncthreads();
                LPROF_T(4)
            }
            // keys under the query's bound join its candidates
            // (distance first: the 32-bit order keys decide all but the rare equal-distance case; most lanes have nothing to push)
#pragma unroll
            for (int u = 0; u < LM_U; ++u) {
                if (valid[u]) {
                    if (order_key(sum[u][0]) <= da) {
                        const uint64_t ka = make_key(sum[u][0], idv[u]);
                        if (ka <= ba) { const uint32_t slot = atomicAdd(&a.cand_cnt[qa], 1u); if (slot < a.cand_cap) a.cand[(size_t)qa * a.cand_cap + slot] = ka; }
                    }
                    if (two && order_key(sum[u][1]) <= db) {
                        const uint64_t kb = make_key(sum[u][1], idv[u]);
                        if (kb <= bb) { const uint32_t slot = atomicAdd(&a.cand_cnt[qb], 1u); if (slot < a.cand_cap) a.cand[(size_t)qb * a.cand_cap + slot] = kb; }
                    }
                }
            }
            LPROF_T(5)
        }
        if (!more_items) break;
        s ^= 1;
        if (wave == 0) nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)t2) + gridDim.x;
    }
    LPROF_END
}

// the queries whose candidate list overflowed, in query order
__global__ __launch_bounds__(1024) void lm_overflow_kernel(const uint32_t *__restrict__ cand_cnt, uint32_t nq, uint32_t cand_cap, uint32_t *__restrict__ redo_list, uint32_t *__restrict__ redo_n) {
    __shared__ uint32_t n;
    if (threadIdx.x == 0) n = 0;
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < nq; q += 1024) if (cand_cnt[q] > cand_cap) redo_list[atomicAdd(&n, 1u)] = q;      // (any order: every query is scanned on its own)
    __syncthreads();
    if (threadIdx.x == 0) *redo_n = n;
}

// the k best of a query's candidates (or of its redo blocks)
struct LmMergeArgs { const uint64_t *cand; const uint32_t *cand_cnt; const uint64_t *redo /* [nq][redo_split][k] */; uint32_t k, cap, cand_cap, redo_split; uint32_t *ids; float *dist; uint32_t *counts; };
__global__ __launch_bounds__(256) void lm_merge_kernel(LmMergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *mins = keys + a.cap;
    uint64_t *thr = mins + 512;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    TopKBuf buf{keys, cnt, thr, a.cap, a.k};
    const bool redone = a.cand_cnt[q] > a.cand_cap;                 // its candidate list overflowed: the query-major kernel scanned all its lists again, redo_split k-blocks
    const uint32_t nc = a.cand_cnt[q];
    const uint64_t *c0 = a.cand + (size_t)q * a.cand_cap, *r0 = a.redo + (size_t)q * a.redo_split * a.k;
    auto key_at = [&](uint64_t i) -> uint64_t { return redone ? r0[i] : c0[i]; };
    const uint32_t m = block_select_topk<256>(key_at, redone ? (uint64_t)a.redo_split * a.k : (uint64_t)nc, buf, mins);
    for (uint32_t i = tid; i < a.k; i += 256) {
        if (i < m) {
            const uint64_t key = buf.keys[i];
            a.ids[(size_t)q * a.k + i] = (uint32_t)key;
            a.dist[(size_t)q * a.k + i] = order_key_inv((uint32_t)(key >> 32));
        } else {
            a.ids[(size_t)q * a.k + i] = 0xFFFFFFFFu;
            a.dist[(size_t)q * a.k + i] = __builtin_inff();
        }
    }
    if (tid == 0) a.counts[q] = m;
}

// ---- PQ encode (pq.rs:220-257): nearest of ncent centroids per 8-d subvector, first minimum wins ------------
// One workgroup = 256 rows of ONE subspace m (m is the fastest-varying part of blockIdx.x, so the workgroups that share a
// row's 128-byte lines run next to each other): the centroid addresses are wave-uniform, so the centroids arrive through
// scalar loads and sit in SGPRs; a lane holds its sub-vector in 8 VGPRs. KEYS: write u32 keys [M][n] (coalesced; what the
// k-means member sort wants) instead of the byte codes [n][M].
template <bool KEYS>
__global__ __launch_bounds__(256) void pq_encode_kernel(const float *__restrict__ rows, uint64_t n, uint32_t dim, const float *__restrict__ codebook,
                                                        uint32_t M, uint32_t ncent, uint8_t *__restrict__ codes, uint32_t *__restrict__ keys) {
    const uint32_t m = blockIdx.x % M;
    const uint64_t row = (uint64_t)(blockIdx.x / M) * 256 + threadIdx.x;
    if (row >= n) return;
    const float *v = rows + row * dim + m * 8;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j];
    uint32_t best = 0;
    float best_dist = 3.4028234663852886e38f;      // f32::MAX
    const float *cb = codebook + (size_t)m * ncent * 8;
#pragma unroll 4
    for (uint32_t c = 0; c < ncent; ++c) {
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = x[j] - cb[c * 8 + j]; sum = sum + d * d; }
        if (sum < best_dist) { best_dist = sum; best = c; }
    }
    if (KEYS) keys[(uint64_t)m * n + row] = best;
    else codes[row * M + m] = (uint8_t)best;
}
static inline uint32_t pq_encode_grid(uint64_t n, uint32_t M) { return (uint32_t)(ceil_div(n, 256) * M); }

// ---- host --------------------------------------------------------------------------------------------------------
static int upload_postings(IvfpqState *s) {
    // flatten the host lists into CSR and upload
    std::vector<uint64_t> off(s->P + 1, 0);
    uint64_t longest = 0;
    for (uint32_t p = 0; p < s->P; ++p) { off[p + 1] = off[p] + s->h_ids[p].size(); if (s->h_ids[p].size() > longest) longest = s->h_ids[p].size(); }
    const uint64_t total = off[s->P];
    s->max_list_len = longest;
    std::vector<uint32_t> ids(total ? total : 1);
    std::vector<uint8_t> codes((total ? total : 1) * s->M);
    for (uint32_t p = 0; p < s->P; ++p) {
        std::copy(s->h_ids[p].begin(), s->h_ids[p].end(), ids.begin() + off[p]);
        std::copy(s->h_codes[p].begin(), s->h_codes[p].end(), codes.begin() + off[p] * s->M);
    }
    if (total > s->cap_total) {
        dev_free(s->ids); dev_free(s->codes); s->ids = nullptr; s->codes = nullptr;
        uint64_t nc = total + total / 4 + 1024;
        SHODH_HIP_TRY(dev_alloc((void **)&s->ids, nc * 4));
        SHODH_HIP_TRY(dev_alloc((void **)&s->codes, nc * s->M));
        s->cap_total = nc;
    }
    SHODH_HIP_TRY(hipMemcpy(s->list_off, off.data(), (s->P + 1) * 8, hipMemcpyHostToDevice));
    if (total) {
        SHODH_HIP_TRY(hipMemcpy(s->ids, ids.data(), total * 4, hipMemcpyHostToDevice));
        SHODH_HIP_TRY(hipMemcpy(s->codes, codes.data(), total * s->M, hipMemcpyHostToDevice));
    }
    s->total = total;
    s->dirty = false;
    return SHODH_OK;
}

struct IvfpqLayout { uint32_t nprobe, cap, split, gx; bool list_major; uint32_t lm_chunk; size_t o_probe_ids, o_probe_dist, o_probe_cnt, o_partial, o_flat, o_tables, o_lm, o_pairs, o_cand, o_items, o_redo, o_redo_list, o_r0, bytes; };
// (round 5, configs[3] step with 64 / 32 / 16 / 8 workgroups per redone query: 1.060 / 1.034 / 1.021 / 1.026 ms -- the launch is mostly empty workgroups of
// 1024 threads and ~100 KiB of LDS, which cost their dispatch; an overflowed query is 1 of 2048)
#ifndef SHODH_LM_REDO_SPLIT
#define SHODH_LM_REDO_SPLIT 16
#endif
constexpr uint32_t LM_REDO_SPLIT = SHODH_LM_REDO_SPLIT, LM_REDO_COLS = 16;      // redo launch: 16 x 32 workgroups, a query per column at a time
constexpr uint32_t LM_QCHUNK = 4096;       // queries per list-major pass (their tables: 192 MiB)
// The list-major scan is the default wherever its shapes hold (48 sub-quantisers of 256 codewords: what the reference builds at 384-d). Measured at
// configs[3]'s index against the query-major scan, ms per batch of 1 / 64 / 256 / 1024 / 4096 queries: 0.088 / 0.23 / 0.43 / 1.08 / 3.44 against
// 0.096 / 0.32 / 0.52 / 1.41 / 4.73 -- even one query spreads its 32 lists over 32 workgroups.
static bool use_list_major(const IvfpqState *s, uint32_t nq, uint32_t nprobe, uint32_t k) {
    (void)nq;
    const char *ev = getenv("SHODH_ADC_LIST_MAJOR");                    // diagnostic / tests (read per call): 0 = the query-major scan
    if (ev && atoi(ev) == 0) return false;
    return s->ncent == 256 && s->M == 48 && k >= 1 && k <= 1024 && nprobe >= 1;
}
static IvfpqLayout ivfpq_layout(const IvfpqState *s, const shodh_index_cfg &cfg, uint32_t nq, uint32_t k) {
    IvfpqLayout L{};
    L.nprobe = cfg.nprobe < s->P ? cfg.nprobe : s->P;     // .take(num_probes)
    L.cap = topk_capacity(k);
    // split a query's probed lists over several blocks when the batch alone cannot fill the chip
    L.split = 1;
    while ((uint64_t)nq * L.split < (uint64_t)s->cus * 8 && L.split < 32) L.split <<= 1;    // two workgroups per CU are resident: at least four rounds of them
    static const int env_split = getenv("SHODH_ADC_SPLIT") ? atoi(getenv("SHODH_ADC_SPLIT")) : 0;     // diagnostic
    if (env_split > 0) L.split = (uint32_t)env_split;
    L.gx = exact_grid_x(s->P, nq, L.nprobe, s->cus);
    const size_t part_probe = exact_partial_bytes(nq, s->dim, L.nprobe, L.gx);
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += (b + 255) & ~(size_t)255; return r; };
    L.o_probe_ids = take((size_t)nq * (L.nprobe ? L.nprobe : 1) * 4);
    L.o_probe_dist = take((size_t)nq * (L.nprobe ? L.nprobe : 1) * 4);
    L.o_probe_cnt = take((size_t)nq * 4);
    L.list_major = use_list_major(s, nq, L.nprobe, k);
    L.lm_chunk = nq < LM_QCHUNK ? nq : LM_QCHUNK;
    const uint32_t pslots = L.list_major ? 1 : L.split;                 // partial keys per query: one k-block per split (list-major: the nearest list's k best)
    L.o_partial = take((size_t)nq * pslots * (k ? k : 1) * 8);
    L.o_flat = take(part_probe + 256);
    L.o_tables = L.o_lm = L.o_pairs = L.o_cand = L.o_items = L.o_redo = L.o_redo_list = L.o_r0 = 0;
    if (L.list_major) {
        L.o_tables = take((size_t)L.lm_chunk * 12288 * 4);
        L.o_lm = take((size_t)(s->P + 1 + L.lm_chunk) * 4 + (size_t)3 * (s->P + 1) * 4);      // list counts (+ the scan's work counter) + candidate counts (one memset), pair offsets, fill cursors, item offsets
        L.o_pairs = 0;
        L.o_cand = take((size_t)L.lm_chunk * LM_CANDS * 8);
        const uint64_t max_chunks = (s->max_list_len + (uint64_t)LM_NT * LM_U - 1) / ((uint64_t)LM_NT * LM_U);
        L.o_items = take((((size_t)L.lm_chunk * L.nprobe / (2 * LM_QB) + s->P + 1) * (max_chunks ? max_chunks : 1) + 1) * sizeof(LmItem));
        L.o_redo = take((size_t)L.lm_chunk * LM_REDO_SPLIT * (k ? k : 1) * 8);
        L.o_redo_list = take((size_t)(L.lm_chunk + 1) * 4);
        L.o_r0 = take((size_t)L.lm_chunk * 8);                         // the bounds
    }
    L.bytes = o;
    return L;
}
size_t ivfpq_scratch_bytes(const IvfpqState *s, const shodh_index_cfg &cfg, uint32_t nq, uint32_t k) {
    return s ? ivfpq_layout(s, cfg, nq, k).bytes : 0;
}

// scratch: ivfpq_scratch_bytes() bytes of device memory private to this search
int ivfpq_search(IvfpqState *s, const shodh_index_cfg &cfg, const float *d_q, uint32_t nq, uint32_t k,
                 uint32_t *d_ids, float *d_dist, uint32_t *d_counts, unsigned char *scratch, hipStream_t st) {
    if (!s || s->P == 0) { set_error("IVF-PQ index has no trained state"); return SHODH_ERR_STATE; }
    {
        std::lock_guard<std::mutex> g(s->mu);
        if (s->dirty) { SHODH_HIP_TRY(hipDeviceSynchronize()); SHODH_TRY(upload_postings(s)); }
    }
    const IvfpqLayout L = ivfpq_layout(s, cfg, nq, k);
    const uint32_t nprobe = L.nprobe, cap = L.cap, split = L.split, gx = L.gx;
    const size_t o_probe_ids = L.o_probe_ids, o_probe_dist = L.o_probe_dist, o_probe_cnt = L.o_probe_cnt, o_partial = L.o_partial, o_flat = L.o_flat;
    uint32_t *probe_ids = (uint32_t *)(scratch + o_probe_ids);
    float *probe_dist = (float *)(scratch + o_probe_dist);
    uint32_t *probe_cnt = (uint32_t *)(scratch + o_probe_cnt);
    uint64_t *partial = (uint64_t *)(scratch + o_partial);
    // 1. probe selection: the nprobe nearest centroids by (compute_distance, index)
    const uint32_t op = (s->metric == SHODH_METRIC_EUCLIDEAN) ? EX_OP_SEQ_L2 : EX_OP_SEQ_ONE_MINUS_DOT;
    if (s->cent_idx) {
        SHODH_TRY(nearest_centroids(s->cent_idx, d_q, nq, s->dim, nprobe, probe_ids, probe_dist, probe_cnt, st));
    } else {
        SHODH_TRY(launch_flat_exact(s->centroids, s->P, s->dim, nullptr, d_q, nq, nprobe, op, 0, (uint64_t *)(scratch + o_flat), gx,
                                    probe_ids, probe_dist, probe_cnt, nullptr, nullptr, st));
    }
    if (L.list_major) {
        // 2'. tables, bounds, every pair list-major, overflowed queries query-major, merge
        const char *ecap = getenv("SHODH_ADC_LM_CAP");                  // tests (read per call): a tiny candidate list makes every query overflow
        const int env_cap = ecap ? atoi(ecap) : 0;
        const uint32_t cand_cap = (env_cap > 0 && env_cap < LM_CANDS) ? (uint32_t)env_cap : (uint32_t)LM_CANDS;
        const char *emax = getenv("SHODH_ADC_LM_BOUND_POSTINGS");       // tests: postings behind the bound (fewer: looser bound, more candidates; under k: no bound)
        const uint32_t bound_max = (emax && atoi(emax) > 0 && atoi(emax) < LM_NEAR_NT * LM_NEAR_U) ? (uint32_t)atoi(emax) : (uint32_t)(LM_NEAR_NT * LM_NEAR_U);
        float *tables = (float *)(scratch + L.o_tables);
        uint32_t *lcnt = (uint32_t *)(scratch + L.o_lm), *cand_cnt = lcnt + (s->P + 1), *pair_off = cand_cnt + L.lm_chunk, *cursor = pair_off + (s->P + 1), *item_off = cursor + (s->P + 1);
        uint64_t *cand = (uint64_t *)(scratch + L.o_cand);
        LmItem *items = (LmItem *)(scratch + L.o_items);
        uint64_t *redo = (uint64_t *)(scratch + L.o_redo);
        uint64_t *bound = (uint64_t *)(scratch + L.o_r0);                 // [m]
        uint32_t *redo_list = (uint32_t *)(scratch + L.o_redo_list);     // [m] + the count
        uint32_t cap_redo = next_pow2(k + (uint32_t)(ADC_NT * ADC_U));
        if (cap_redo < cap) cap_redo = cap;
        const size_t lds_redo = (size_t)s->M * s->ncent * 4 + (size_t)cap_redo * 8 + 8 + 8 + (size_t)s->dim * 4 + 8 + (size_t)ADC_MAXP * 8 + (size_t)(ADC_MAXP + 1) * 4 + 16;
        uint32_t cap_near = next_pow2(k + 2u * LM_NEAR_NT);            // the k kept + two rounds of pushes between checks
        const size_t lds_near = 49152 + (size_t)cap_near * 8 + (size_t)2 * LM_NEAR_NT * 8 + (size_t)LM_NEAR_MAXL * 8 + 8 + (size_t)(LM_NEAR_MAXL + 1) * 4 + 4 + 4 + 16;
        if (lds_redo > 160 * 1024 || lds_near > 160 * 1024) { set_error("IVF-PQ: dim/k too large for LDS (%zu B)", lds_redo > lds_near ? lds_redo : lds_near); return SHODH_ERR_UNSUPPORTED; }
        if (nprobe > (uint32_t)ADC_MAXP) { set_error("IVF-PQ: nprobe %u > %d", nprobe, ADC_MAXP); return SHODH_ERR_UNSUPPORTED; }
        SHODH_TRY(ensure_dynamic_lds((const void *)adc_bound_kernel, lds_near));
        SHODH_TRY(ensure_dynamic_lds((const void *)adc_scan_kernel<true, ADC_NT, true>, lds_redo));
        SHODH_TRY(ensure_dynamic_lds((const void *)adc_list_kernel, LM_LDS));
        const size_t mlds2 = (size_t)cap * 8 + 512 * 8 + 8 + 4 + 16;
        SHODH_TRY(ensure_dynamic_lds((const void *)lm_merge_kernel, mlds2));
        const uint32_t max_chunks = (uint32_t)((s->max_list_len + (uint64_t)LM_NT * LM_U - 1) / ((uint64_t)LM_NT * LM_U));
        for (uint32_t b = 0; b < nq; b += L.lm_chunk) {
            const uint32_t m = (nq - b) < L.lm_chunk ? (nq - b) : L.lm_chunk;
            const float *qb = d_q + (size_t)b * s->dim;
            const uint32_t *pids = probe_ids + (size_t)b * nprobe, *pcnt = probe_cnt + b;
            // tables; list counters + work counter + candidate counters (adjacent) cleared: 12 * ceil(m / 8) * 1024 threads, P + 1 + m words
            const uint32_t n_zero = s->P + 1 + m, tgx = (m + LM_TQ - 1) / LM_TQ;
            if ((uint64_t)tgx * 12 * 1024 < n_zero) SHODH_HIP_TRY(hipMemsetAsync(lcnt, 0, (size_t)n_zero * 4, st));      // (few queries, many lists)
            hipLaunchKernelGGL(adc_table_kernel, dim3(tgx, 12), dim3(1024), 0, st, qb, s->codebook, s->dim, m, tables, lcnt, n_zero);
            NearArgs na{tables, s->list_off, s->ids, s->codes, pids, pcnt, nprobe, k, cap_near, s->P, bound_max, bound, lcnt};
            hipLaunchKernelGGL(adc_bound_kernel, dim3(m), dim3(LM_NEAR_NT), lds_near, st, na);
            const uint32_t np = m * nprobe;
            hipLaunchKernelGGL(lm_scan_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)lcnt, s->list_off, s->P, pair_off, cursor, item_off);
            hipLaunchKernelGGL(lm_fill_kernel, dim3((np + 255) / 256), dim3(256), 0, st, pids, pcnt, m, nprobe, s->P, s->list_off, (const uint32_t *)pair_off, (const uint32_t *)item_off, (const uint64_t *)bound, cursor, items);
            // one workgroup per CU (96 KiB of LDS each) walks the items; lcnt[P] (zeroed with the list counters, not a list) is the shared work counter
            LmArgs la{tables, s->ids, s->codes, items, item_off + s->P, lcnt + s->P, cand_cap, cand, cand_cnt};
            const uint64_t max_items = ((uint64_t)np / (2 * LM_QB) + (np < s->P ? np : s->P)) * (max_chunks ? max_chunks : 1);
            const uint32_t lm_grid = (uint32_t)std::min<uint64_t>(max_items ? max_items : 1, (uint64_t)s->cus * LM_WG_PER_CU);
            hipLaunchKernelGGL(adc_list_kernel, dim3(lm_grid), dim3(LM_NT), LM_LDS, st, la);
#ifdef SHODH_LMPROF
            {
                std::vector<uint32_t> hc(m);
                hipStreamSynchronize(st); hipMemcpy(hc.data(), cand_cnt, (size_t)m * 4, hipMemcpyDeviceToHost);
                uint64_t sum = 0; uint32_t mx = 0, over = 0, big = 0, mq = 0;
                for (uint32_t i = 0; i < m; ++i) { const uint32_t v = hc[i]; sum += v; if (v > mx) { mx = v; mq = i; } over += v > cand_cap; big += v > 512; }
                fprintf(stderr, "lmprof candidates: mean %.1f max %u (query %u) over-cap %u (>512: %u) of %u queries\n", (double)sum / m, mx, mq, over, big, m);
            }
#endif
            // queries whose candidate list overflowed (an outlier whose nearest lists say little about the others): from scratch, every probed list,
            // query-major, LM_REDO_SPLIT workgroups per query
            hipLaunchKernelGGL(lm_overflow_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)cand_cnt, m, cand_cap, redo_list, redo_list + m);
            AdcArgs a2{qb, s->codebook, s->list_off, s->ids, s->codes, pids, pcnt, m, s->dim, s->M, s->ncent, nprobe, k, cap_redo, LM_REDO_SPLIT, redo, redo_list, redo_list + m, tables};
            hipLaunchKernelGGL((adc_scan_kernel<true, ADC_NT, true>), dim3(LM_REDO_COLS, LM_REDO_SPLIT), dim3(ADC_NT), lds_redo, st, a2);
            {
                static const bool dbg = getenv("SHODH_ADC_DEBUG") && atoi(getenv("SHODH_ADC_DEBUG")) != 0;      // diagnostics: how many queries were redone, the largest candidate list
                if (dbg) {
                    std::vector<uint32_t> hc(m + 1), hr(1);
                    hipStreamSynchronize(st); hipMemcpy(hc.data(), cand_cnt, (size_t)m * 4, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), redo_list + m, 4, hipMemcpyDeviceToHost);
                    uint32_t mx = 0, mq = 0; uint64_t sum = 0;
                    for (uint32_t i = 0; i < m; ++i) { sum += hc[i]; if (hc[i] > mx) { mx = hc[i]; mq = i; } }
                    fprintf(stderr, "[adc] %u queries: candidates mean %.1f max %u (query %u), cap %u, redone %u\n", m, (double)sum / m, mx, mq, cand_cap, hr[0]);
                }
            }
            LmMergeArgs mm{cand, cand_cnt, redo, k, cap, cand_cap, LM_REDO_SPLIT, d_ids + (size_t)b * k, d_dist + (size_t)b * k, d_counts + b};
            hipLaunchKernelGGL(lm_merge_kernel, dim3(m), dim3(256), mlds2, st, mm);
            SHODH_HIP_TRY(hipGetLastError());
        }
        return SHODH_OK;
    }
    // 2. ADC table + list scan, 3. merge
    uint32_t scan_cap = next_pow2(k + (uint32_t)(ADC_NT * ADC_U));     // room for one iteration's pushes on top of the k kept keys
    if (scan_cap < cap) scan_cap = cap;
    AdcArgs a{d_q, s->codebook, s->list_off, s->ids, s->codes, probe_ids, probe_cnt, nq, s->dim, s->M, s->ncent, nprobe, k, scan_cap, split, partial, nullptr, nullptr, nullptr};
    const size_t lds = (size_t)s->M * s->ncent * 4 + (size_t)scan_cap * 8 + 8 + 8 + (size_t)s->dim * 4 + 8 + (size_t)ADC_MAXP * 8 + (size_t)(ADC_MAXP + 1) * 4 + 16;
    if (lds > 160 * 1024) { set_error("IVF-PQ: dim/k too large for LDS (%zu B)", lds); return SHODH_ERR_UNSUPPORTED; }
    if (nprobe > (uint32_t)ADC_MAXP) { set_error("IVF-PQ: nprobe %u > %d", nprobe, ADC_MAXP); return SHODH_ERR_UNSUPPORTED; }
    if (s->ncent == 256) {
        SHODH_TRY(ensure_dynamic_lds((const void *)adc_scan_kernel<true, ADC_NT>, lds));
        hipLaunchKernelGGL((adc_scan_kernel<true, ADC_NT>), dim3(nq, split), dim3(ADC_NT), lds, st, a);
    } else {
        SHODH_TRY(ensure_dynamic_lds((const void *)adc_scan_kernel<false, ADC_NT>, lds));
        hipLaunchKernelGGL((adc_scan_kernel<false, ADC_NT>), dim3(nq, split), dim3(ADC_NT), lds, st, a);
    }
    SHODH_HIP_TRY(hipGetLastError());
    AdcMergeArgs m{partial, split, k, cap, d_ids, d_dist, d_counts};
    const size_t mlds = (size_t)cap * 8 + 512 * 8 + 8 + 4 + 16;
    SHODH_TRY(ensure_dynamic_lds((const void *)adc_merge_kernel, mlds));
    hipLaunchKernelGGL(adc_merge_kernel, dim3(nq), dim3(256), mlds, st, m);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

IvfpqState *&index_ivfpq_slot(shodh_index *idx);
const shodh_index_cfg &index_cfg(const shodh_index *idx);
std::shared_mutex &index_mutex(shodh_index *idx);

// assign_out / codes_out on the HOST; rows on the host
static int encode_rows(IvfpqState *s, const float *rows, uint64_t n, uint32_t *assign_out, uint8_t *codes_out) {
    if (n == 0) return SHODH_OK;
    std::lock_guard<std::mutex> g(s->mu);
    const uint32_t gx = exact_grid_x(s->P, 1, 1, s->cus);
    const uint64_t CH = 65536;     // rows per chunk (each row is one "query" of the nearest-centroid scan)
    float *d_rows = nullptr; uint8_t *d_codes = nullptr; uint32_t *d_assign = nullptr; float *d_ad = nullptr; uint32_t *d_ac = nullptr; uint64_t *d_part = nullptr;
    const uint64_t ch = n < CH ? n : CH;
    const uint32_t gxc = exact_grid_x(s->P, (uint32_t)ch, 1, s->cus);
    (void)gx;
    const size_t part_bytes = exact_partial_bytes((uint32_t)ch, s->dim, 1, gxc);
    SHODH_HIP_TRY(dev_alloc((void **)&d_rows, ch * s->dim * 4));
    SHODH_HIP_TRY(dev_alloc((void **)&d_codes, ch * s->M));
    SHODH_HIP_TRY(dev_alloc((void **)&d_assign, ch * 4));
    SHODH_HIP_TRY(dev_alloc((void **)&d_ad, ch * 4));
    SHODH_HIP_TRY(dev_alloc((void **)&d_ac, ch * 4));
    SHODH_HIP_TRY(dev_alloc((void **)&d_part, part_bytes + 256));
    int rc = SHODH_OK;
    const uint32_t op = (s->metric == SHODH_METRIC_EUCLIDEAN) ? EX_OP_SEQ_L2 : EX_OP_SEQ_ONE_MINUS_DOT;
    for (uint64_t b = 0; b < n && rc == SHODH_OK; b += ch) {
        const uint64_t m = (n - b) < ch ? (n - b) : ch;
        if (hipMemcpy(d_rows, rows + b * s->dim, m * s->dim * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = SHODH_ERR_DEVICE; break; }
        // find_nearest_centroid (spann.rs:545-558): strict '<' keeps the first minimum == smallest (dist, index)
        rc = s->cent_idx ? nearest_centroids(s->cent_idx, d_rows, m, s->dim, 1, d_assign, d_ad, d_ac, nullptr)
                         : launch_flat_exact(s->centroids, s->P, s->dim, nullptr, d_rows, (uint32_t)m, 1, op, 0, d_part, gxc, d_assign, d_ad, d_ac, nullptr, nullptr, nullptr);
        if (rc != SHODH_OK) break;
        hipLaunchKernelGGL(pq_encode_kernel<false>, dim3(pq_encode_grid(m, s->M)), dim3(256), 0, nullptr, d_rows, m, s->dim, s->codebook, s->M, s->ncent, d_codes, (uint32_t *)nullptr);
        if (hipGetLastError() != hipSuccess) { rc = SHODH_ERR_DEVICE; break; }
        if (hipMemcpy(assign_out + b, d_assign, m * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(codes_out + b * s->M, d_codes, m * s->M, hipMemcpyDeviceToHost) != hipSuccess) { rc = SHODH_ERR_DEVICE; break; }
    }
    if (rc == SHODH_ERR_DEVICE) set_error("IVF-PQ encode failed on device: %s", hipGetErrorString(hipGetLastError()));
    dev_free(d_rows); dev_free(d_codes); dev_free(d_assign); dev_free(d_ad); dev_free(d_ac); dev_free(d_part);
    return rc;
}

}  // namespace shodh

using namespace shodh;

extern "C" {

int shodh_index_set_ivfpq(shodh_index *idx, const float *centroids, uint32_t P, const float *codebook, uint32_t M, uint32_t ncent,
                          const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes) {
    if (!idx || !centroids || !codebook || !list_off || P == 0) { set_error("null argument"); return SHODH_ERR_INVALID; }
    const shodh_index_cfg &cfg = index_cfg(idx);
    if (cfg.kind != SHODH_INDEX_IVFPQ) { set_error("not an IVF-PQ index"); return SHODH_ERR_STATE; }
    if (M * 8 != cfg.dim) { set_error("PQ subvectors %u x 8 != dimension %u", M, cfg.dim); return SHODH_ERR_DIM; }
    if (ncent == 0 || ncent > 256) { set_error("PQ centroids per subspace must be 1..256"); return SHODH_ERR_INVALID; }
    std::unique_lock<std::shared_mutex> lk(index_mutex(idx));
    SHODH_HIP_TRY(hipSetDevice(cfg.device));
    SHODH_HIP_TRY(hipDeviceSynchronize());
    IvfpqState *&slot = index_ivfpq_slot(idx);
    if (slot) { ivfpq_destroy(slot); slot = nullptr; }
    IvfpqState *s = new IvfpqState();
    hipDeviceProp_t prop;
    SHODH_HIP_TRY(hipGetDeviceProperties(&prop, cfg.device));
    s->device = cfg.device; s->cus = prop.multiProcessorCount; s->dim = cfg.dim; s->P = P; s->M = M; s->ncent = ncent; s->metric = cfg.metric;
    const uint64_t total = list_off[P];
    if ((total && (!ids || !codes))) { delete s; set_error("null postings"); return SHODH_ERR_INVALID; }
    SHODH_HIP_TRY(dev_alloc((void **)&s->centroids, (size_t)P * cfg.dim * 4));
    SHODH_HIP_TRY(dev_alloc((void **)&s->codebook, (size_t)M * ncent * 8 * 4));
    SHODH_HIP_TRY(dev_alloc((void **)&s->list_off, (size_t)(P + 1) * 8));
    SHODH_HIP_TRY(hipMemcpy(s->centroids, centroids, (size_t)P * cfg.dim * 4, hipMemcpyHostToDevice));
    SHODH_HIP_TRY(hipMemcpy(s->codebook, codebook, (size_t)M * ncent * 8 * 4, hipMemcpyHostToDevice));
    if (cfg.metric != SHODH_METRIC_EUCLIDEAN) {
        const int rc_ci = make_centroid_index(cfg.device, cfg.dim, s->centroids, P, &s->cent_idx);
        if (rc_ci != SHODH_OK) { ivfpq_destroy(s); return rc_ci; }
    }
    s->h_ids.resize(P); s->h_codes.resize(P);
    for (uint32_t p = 0; p < P; ++p) {
        if (list_off[p + 1] < list_off[p]) { ivfpq_destroy(s); set_error("list_off not monotone"); return SHODH_ERR_INVALID; }
        s->h_ids[p].assign(ids + list_off[p], ids + list_off[p + 1]);
        s->h_codes[p].assign(codes + list_off[p] * M, codes + list_off[p + 1] * M);
    }
    int rc = upload_postings(s);
    if (rc != SHODH_OK) { ivfpq_destroy(s); return rc; }
    slot = s;
    return SHODH_OK;
}

int shodh_index_ivfpq_encode(shodh_index *idx, const float *rows, uint64_t n, uint32_t *assign_out, uint8_t *codes_out) {
    if (!idx || (n && (!rows || !assign_out || !codes_out))) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::shared_lock<std::shared_mutex> lk(index_mutex(idx));
    IvfpqState *s = index_ivfpq_slot(idx);
    if (!s) { set_error("Cannot insert into empty index - build first"); return SHODH_ERR_STATE; }     // spann.rs:1008-1011
    SHODH_HIP_TRY(hipSetDevice(s->device));
    return encode_rows(s, rows, n, assign_out, codes_out);
}

int shodh_index_ivfpq_insert(shodh_index *idx, uint32_t vector_id, const float *row) {
    if (!idx || !row) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::unique_lock<std::shared_mutex> lk(index_mutex(idx));
    IvfpqState *s = index_ivfpq_slot(idx);
    if (!s) { set_error("Cannot insert into empty index - build first"); return SHODH_ERR_STATE; }
    SHODH_HIP_TRY(hipSetDevice(s->device));
    uint32_t part = 0;
    std::vector<uint8_t> code(s->M);
    SHODH_TRY(encode_rows(s, row, 1, &part, code.data()));
    std::lock_guard<std::mutex> g(s->mu);
    if (part < s->P) {                                   // spann.rs:1040-1046
        s->h_ids[part].push_back(vector_id);
        s->h_codes[part].insert(s->h_codes[part].end(), code.begin(), code.end());
        if (s->h_ids[part].size() > s->max_list_len) s->max_list_len = s->h_ids[part].size();      // (the scratch layout of the next search is sized before the re-upload)
        s->dirty = true;
    }
    return SHODH_OK;
}

}  // extern "C"

// =====================================================================================================
// ---- Lloyd k-means of the IVF centroids and the PQ codebooks on the device.
//
// Follows SpannIndex::kmeans_cluster (src/vector_db/spann.rs:466-541) and ProductQuantizer::kmeans
// (src/vector_db/pq.rs:152-217) operation for operation, so that GIVEN the initial shuffles (the reference draws them
// from thread_rng) the trained state is bit-identical to the reference's (tests/test_ivfpq_gpu.py checks it against a CPU restatement):
//   assignment  : find_nearest_centroid (spann.rs:545-558: strictly sequential `1 - sum x*y`, strict '<' keeps the first
//                 minimum) = the exact-order scan kernel over the centroids with k = 1;
//                 PQ: squared L2 over 8 dims, sequential, strict '<' (pq.rs:180-191) = pq_encode_kernel;
//   update      : the reference adds the members of a cluster INTO the new centroid in vector-index order, one f32 add
//                 per member and coordinate, then divides by the count. A parallel tree sum would round differently, so
//                 the members of every cluster are listed in index order (a STABLE radix sort of (cluster, index), rocPRIM)
//                 and one thread per (cluster, coordinate) adds them in that order; the parallelism is across clusters and
//                 coordinates (P x dim, or 48 x 256 x 8 for the codebooks), not inside a sum;
//   empty cluster keeps its previous centroid (spann.rs:521-527, pq.rs:205-212);
//   IVF stops early when no assignment changed (spann.rs:536-539, AFTER the update of that iteration); PQ always runs
//   all its iterations.
namespace shodh {
namespace {

__global__ void gather_rows_kernel(const float *rows, const uint32_t *perm, uint64_t n, uint32_t dim, uint32_t k, float *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)k * dim) return;
    const uint32_t c = (uint32_t)(t / dim), j = (uint32_t)(t % dim);
    out[t] = rows[(uint64_t)perm[c % n] * dim + j];                 // centroids[c] = vectors[indices[c % n]] (spann.rs:476-488)
}
// codebook[m][c][:] = rows[perm[m][c % n]][8m : 8m+8]   (pq.rs:158-172)
__global__ void gather_sub_kernel(const float *rows, const uint32_t *perm, uint64_t n, uint32_t dim, uint32_t M, uint32_t ncent, float *cb) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)M * ncent * 8) return;
    const uint32_t j = (uint32_t)(t % 8), c = (uint32_t)((t / 8) % ncent), m = (uint32_t)(t / (8 * ncent));
    cb[t] = rows[(uint64_t)perm[(uint64_t)m * n + c % n] * dim + m * 8 + j];
}
__global__ void iota_kernel(uint32_t *v, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = (uint32_t)t;
}
__global__ void count_changed_kernel(const uint32_t *a, uint32_t *prev, uint64_t n, unsigned long long *changed) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned c = 0;
    if (t < n) { c = a[t] != prev[t]; prev[t] = a[t]; }
    const unsigned long long b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(c != 0));
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(changed, b);
}
// offsets[c] = first position of the sorted keys that is >= c (c = 0..k): the member list boundaries, nsub key arrays at once
__global__ void offsets_kernel(const uint32_t *sorted_keys, uint64_t n, uint32_t k, uint32_t nsub, uint32_t *offsets /* [nsub][k+1] */) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)nsub * (k + 1)) return;
    const uint32_t sub = (uint32_t)(t / (k + 1)), c = (uint32_t)(t % (k + 1));
    const uint32_t *keys = sorted_keys + (uint64_t)sub * n;
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (keys[mid] < c) lo = mid + 1; else hi = mid; }
    offsets[t] = (uint32_t)lo;
}
// One WAVE per (sub-space, cluster, group of 8 coordinates): the members in index order, one add each, then / count
// (spann.rs:508-527, pq.rs:193-212). The adds of one coordinate are a serial chain by definition, so the parallelism inside a
// cluster is in the LOADS: lane = (member slot 0..7, coordinate 0..7); 16 x 8 members are fetched per step (and the next step's
// are already in flight) and every lane then walks the 128 values of its coordinate in member order through ds_bpermute. Slots
// past the end add +0.0, which leaves an f32 sum that started at +0.0 unchanged. Sub-space `sub` owns columns
// [sub*width, sub*width+width) of the rows (IVF: one sub-space of width dim); width % 8 == 0.
constexpr int MU_U = 16;
__global__ __launch_bounds__(256) void mean_update_kernel(const float *rows, uint64_t n, uint32_t dim, uint32_t width, uint32_t nsub, const uint32_t *members /* [nsub][n] */,
                                                          const uint32_t *offsets /* [nsub][k+1] */, uint32_t k, float *cent /* [nsub][k][width] */) {
    const uint64_t w = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t groups = width >> 3;
    if (w >= (uint64_t)nsub * k * groups) return;                                      // wave-uniform
    const uint32_t lane = threadIdx.x & 63, slot = lane >> 3, j = lane & 7;
    const uint32_t g = (uint32_t)(w % groups), c = (uint32_t)((w / groups) % k), sub = (uint32_t)(w / ((uint64_t)groups * k));
    const uint32_t *mem = members + (uint64_t)sub * n;
    const uint32_t lo = offsets[(uint64_t)sub * (k + 1) + c], hi = offsets[(uint64_t)sub * (k + 1) + c + 1];
    if (hi == lo) return;                                                              // empty cluster keeps its centroid
    const float *col = rows + (uint64_t)sub * width + g * 8 + j;
    float cur[MU_U], nxt[MU_U] = {};
    auto fetch = [&](uint32_t i0, float *v) {                                          // unconditional loads on clamped indices
#pragma unroll
        for (int u = 0; u < MU_U; ++u) {
            const uint32_t i = i0 + u * 8 + slot;
            const float x = col[(uint64_t)mem[i < hi ? i : hi - 1] * dim];
            v[u] = i < hi ? x : 0.0f;
        }
    };
    fetch(lo, cur);
    float sum = 0.0f;
    for (uint32_t i0 = lo; i0 < hi; i0 += MU_U * 8) {
        if (i0 + MU_U * 8 < hi) fetch(i0 + MU_U * 8, nxt);                             // wave-uniform condition
#pragma unroll
        for (int u = 0; u < MU_U; ++u)
#pragma unroll
            for (int s = 0; s < 8; ++s) sum = sum + __shfl(cur[u], s * 8 + (int)j);
#pragma unroll
        for (int u = 0; u < MU_U; ++u) cur[u] = nxt[u];
    }
    if (slot == 0) cent[((uint64_t)sub * k + c) * width + g * 8 + j] = sum / (float)(hi - lo);
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) dev_free(p); }
    int alloc(size_t bytes) { return dev_alloc(&p, bytes ? bytes : 16) == hipSuccess ? SHODH_OK : SHODH_ERR_OOM; }
    template <class T> T *as() { return (T *)p; }
};

// members of every cluster in index order: stable radix sort of (key, index) on the low `bits` bits of the key
int sorted_members(const uint32_t *keys, uint64_t n, uint32_t k, uint32_t *idx_in, uint32_t *keys_out, uint32_t *members, Buf &tmp, size_t &tmp_bytes) {
    int bits = 1;
    while ((1u << bits) < k) ++bits;
    size_t need = 0;
    SHODH_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, keys_out, idx_in, members, (int)n, 0, bits));
    if (need > tmp_bytes) { if (tmp.p) dev_free(tmp.p); tmp.p = nullptr; SHODH_TRY(tmp.alloc(need)); tmp_bytes = need; }
    SHODH_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, need, keys, keys_out, idx_in, members, (int)n, 0, bits));
    return SHODH_OK;
}

}  // namespace
}  // namespace shodh

using namespace shodh;

extern "C" int shodh_ivfpq_train(int device, const float *rows, uint64_t n, uint32_t dim, uint32_t P, uint32_t ivf_iters, uint32_t pq_iters,
                                 const uint32_t *init_perm_ivf, const uint32_t *init_perm_pq, float *centroids_out, float *codebook_out) {
    if (!rows || !init_perm_ivf || !init_perm_pq || !centroids_out || !codebook_out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (n == 0 || P == 0) { set_error("Cannot build index from empty vectors"); return SHODH_ERR_INVALID; }     // spann.rs:364-366
    if (dim % 8 != 0) { set_error("Dimension %u not divisible by 8", dim); return SHODH_ERR_DIM; }               // pq.rs:43-48
    if (n > 0x7FFFFFFFull) { set_error("k-means: more than 2^31 vectors"); return SHODH_ERR_UNSUPPORTED; }
    SHODH_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    SHODH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const uint32_t M = dim / 8, NC = 256;
    Buf d_rows, d_cent, d_cb, d_assign, d_prev, d_idx, d_members, d_off, d_changed, d_perm, d_keys, d_ad, d_ac, tmp;
    size_t tmp_bytes = 0;
    SHODH_TRY(d_rows.alloc(n * dim * 4)); SHODH_TRY(d_cent.alloc((size_t)P * dim * 4)); SHODH_TRY(d_cb.alloc((size_t)M * NC * 8 * 4));
    SHODH_TRY(d_assign.alloc(n * 4)); SHODH_TRY(d_prev.alloc(n * 4)); SHODH_TRY(d_idx.alloc(n * 4));
    SHODH_TRY(d_members.alloc((size_t)M * n * 4)); SHODH_TRY(d_off.alloc(((size_t)(P + 1) + (size_t)M * (NC + 1)) * 4));
    SHODH_TRY(d_changed.alloc(8)); SHODH_TRY(d_perm.alloc((size_t)M * n * 4));
    SHODH_TRY(d_keys.alloc((size_t)M * n * 4));
    SHODH_TRY(d_ad.alloc(n * 4)); SHODH_TRY(d_ac.alloc(n * 4));
    SHODH_HIP_TRY(hipMemcpy(d_rows.p, rows, n * dim * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(iota_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, nullptr, d_idx.as<uint32_t>(), n);
    struct CentIdx { shodh_index *p = nullptr; ~CentIdx() { if (p) shodh_index_destroy(p); } } ci;

    // ---- IVF centroids (spann.rs:466-541) ----
    SHODH_HIP_TRY(hipMemcpy(d_perm.p, init_perm_ivf, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)ceil_div((uint64_t)P * dim, 256)), dim3(256), 0, nullptr, d_rows.as<float>(), d_perm.as<uint32_t>(), n, dim, P, d_cent.as<float>());
    SHODH_HIP_TRY(hipMemset(d_prev.p, 0, n * 4));                   // `let mut assignments = vec![0usize; n]`
    for (uint32_t it = 0; it < ivf_iters; ++it) {
        {
            const int rc_ci = make_centroid_index(device, dim, d_cent.as<float>(), P, &ci.p);
            if (rc_ci != SHODH_OK) return rc_ci;
            SHODH_TRY(nearest_centroids(ci.p, d_rows.as<float>(), n, dim, 1, d_assign.as<uint32_t>(), d_ad.as<float>(), d_ac.as<uint32_t>(), nullptr));
        }
        SHODH_HIP_TRY(hipMemset(d_changed.p, 0, 8));
        hipLaunchKernelGGL(count_changed_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, nullptr, d_assign.as<uint32_t>(), d_prev.as<uint32_t>(), n, d_changed.as<unsigned long long>());
        SHODH_TRY(sorted_members(d_assign.as<uint32_t>(), n, P, d_idx.as<uint32_t>(), d_perm.as<uint32_t>(), d_members.as<uint32_t>(), tmp, tmp_bytes));
        hipLaunchKernelGGL(offsets_kernel, dim3((uint32_t)ceil_div((uint64_t)P + 1, 256)), dim3(256), 0, nullptr, d_perm.as<uint32_t>(), n, P, 1u, d_off.as<uint32_t>());
        hipLaunchKernelGGL(mean_update_kernel, dim3((uint32_t)ceil_div((uint64_t)P * (dim / 8) * 64, 256)), dim3(256), 0, nullptr, d_rows.as<float>(), n, dim, dim, 1u, d_members.as<uint32_t>(), d_off.as<uint32_t>(), P, d_cent.as<float>());
        SHODH_HIP_TRY(hipGetLastError());
        unsigned long long changed = 0;
        SHODH_HIP_TRY(hipMemcpy(&changed, d_changed.p, 8, hipMemcpyDeviceToHost));
        if (changed == 0) break;                                     // "K-means converged" (spann.rs:536-539)
    }
    SHODH_HIP_TRY(hipMemcpy(centroids_out, d_cent.p, (size_t)P * dim * 4, hipMemcpyDeviceToHost));

    // ---- PQ codebooks (pq.rs:131-217): the M subspaces advance in lock-step, they are independent ----
    SHODH_HIP_TRY(hipMemcpy(d_perm.p, init_perm_pq, (size_t)M * n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gather_sub_kernel, dim3((uint32_t)ceil_div((uint64_t)M * NC * 8, 256)), dim3(256), 0, nullptr, d_rows.as<float>(), d_perm.as<uint32_t>(), n, dim, M, NC, d_cb.as<float>());
    for (uint32_t it = 0; it < pq_iters; ++it) {                      // d_perm is free after the gather: it holds the sorted keys from here on
        hipLaunchKernelGGL(pq_encode_kernel<true>, dim3(pq_encode_grid(n, M)), dim3(256), 0, nullptr, d_rows.as<float>(), n, dim, d_cb.as<float>(), M, NC, (uint8_t *)nullptr, d_keys.as<uint32_t>());
        SHODH_HIP_TRY(hipGetLastError());
        for (uint32_t m = 0; m < M; ++m)
            SHODH_TRY(sorted_members(d_keys.as<uint32_t>() + (uint64_t)m * n, n, NC, d_idx.as<uint32_t>(), d_perm.as<uint32_t>() + (uint64_t)m * n, d_members.as<uint32_t>() + (uint64_t)m * n, tmp, tmp_bytes));
        hipLaunchKernelGGL(offsets_kernel, dim3((uint32_t)ceil_div((uint64_t)M * (NC + 1), 256)), dim3(256), 0, nullptr, d_perm.as<uint32_t>(), n, NC, M, d_off.as<uint32_t>());
        hipLaunchKernelGGL(mean_update_kernel, dim3((uint32_t)ceil_div((uint64_t)M * NC * 64, 256)), dim3(256), 0, nullptr, d_rows.as<float>(), n, dim, 8u, M, d_members.as<uint32_t>(), d_off.as<uint32_t>(), NC, d_cb.as<float>());
        SHODH_HIP_TRY(hipGetLastError());
    }
    SHODH_HIP_TRY(hipMemcpy(codebook_out, d_cb.p, (size_t)M * NC * 8 * 4, hipMemcpyDeviceToHost));
    SHODH_HIP_TRY(hipDeviceSynchronize());
    return SHODH_OK;
}

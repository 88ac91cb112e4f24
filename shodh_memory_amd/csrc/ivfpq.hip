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
    uint64_t total = 0, cap_total = 0;
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
    hipFree(s->centroids); hipFree(s->codebook); hipFree(s->list_off); hipFree(s->ids); hipFree(s->codes); hipFree(s->scratch);
    delete s;
}

static int ensure_scratch(IvfpqState *s, size_t bytes) {
    if (bytes <= s->scratch_bytes) return SHODH_OK;
    if (s->scratch) hipFree(s->scratch);
    s->scratch = nullptr; s->scratch_bytes = 0;
    SHODH_HIP_TRY(hipMalloc((void **)&s->scratch, bytes));
    s->scratch_bytes = bytes;
    return SHODH_OK;
}

// ---- nearest centroids through a flat index (SpannIndex::find_nearest_centroid / probe selection, spann.rs:545-571, :595-607) ----
// The k smallest (compute_distance, index) pairs are exactly what a flat search in SHODH_ORDER_SEQ_1M returns: strict '<' in
// the reference keeps the first minimum, i.e. the smallest index among equal distances.
static int make_centroid_index(int device, uint32_t dim, const float *d_centroids, uint32_t P, shodh_index **out) {
    shodh_index_cfg c;
    shodh_index_cfg_default(&c);
    c.dim = dim; c.order = SHODH_ORDER_SEQ_1M; c.device = device; c.reserve_rows = P;
    if (!*out) SHODH_TRY(shodh_index_create(&c, out));
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
};

constexpr int ADC_U = 1;        // postings per thread and iteration
constexpr int ADC_NT = 1024;    // threads per workgroup: the 48 KiB table allows two workgroups per CU; 2 x 16 waves keep the LDS gathers busy
                                // (measured at 10M rows, batch 1024: 256 threads x 2 postings 2.36 ms, 512 x 2 1.87 ms, 1024 x 1 1.73 ms)
constexpr int ADC_MAXP = 1024;  // probed lists per query (nprobe is capped at P and at this; the reference's default is 20)
// FULL256: 256 codewords per sub-quantiser (always the case for a reference-built index, pq.rs:27): an 8-bit code cannot
// leave the table, so the range check disappears and the table row becomes an immediate offset of the LDS read.
template <bool FULL256, int NT>
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
    const uint32_t q = blockIdx.x, part = blockIdx.y;
    for (uint32_t i = tid; i < a.dim; i += NT) qs[i] = a.q[(size_t)q * a.dim + i];
    if (tid == 0) { *cnt = 0; *thr = a.k ? KEY_NONE : 0; }
    __syncthreads();
    // build_distance_table (pq.rs:329-351): one (m, c) entry per thread-iteration, 8 sequential terms
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
    for (uint32_t p = 0; p < s->P; ++p) off[p + 1] = off[p] + s->h_ids[p].size();
    const uint64_t total = off[s->P];
    std::vector<uint32_t> ids(total ? total : 1);
    std::vector<uint8_t> codes((total ? total : 1) * s->M);
    for (uint32_t p = 0; p < s->P; ++p) {
        std::copy(s->h_ids[p].begin(), s->h_ids[p].end(), ids.begin() + off[p]);
        std::copy(s->h_codes[p].begin(), s->h_codes[p].end(), codes.begin() + off[p] * s->M);
    }
    if (total > s->cap_total) {
        hipFree(s->ids); hipFree(s->codes); s->ids = nullptr; s->codes = nullptr;
        uint64_t nc = total + total / 4 + 1024;
        SHODH_HIP_TRY(hipMalloc((void **)&s->ids, nc * 4));
        SHODH_HIP_TRY(hipMalloc((void **)&s->codes, nc * s->M));
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

struct IvfpqLayout { uint32_t nprobe, cap, split, gx; size_t o_probe_ids, o_probe_dist, o_probe_cnt, o_partial, o_flat, bytes; };
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
    L.o_partial = take((size_t)nq * L.split * (k ? k : 1) * 8);
    L.o_flat = take(part_probe + 256);
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
    // 2. ADC table + list scan, 3. merge
    uint32_t scan_cap = next_pow2(k + (uint32_t)(ADC_NT * ADC_U));     // room for one iteration's pushes on top of the k kept keys
    if (scan_cap < cap) scan_cap = cap;
    AdcArgs a{d_q, s->codebook, s->list_off, s->ids, s->codes, probe_ids, probe_cnt, nq, s->dim, s->M, s->ncent, nprobe, k, scan_cap, split, partial};
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
    SHODH_HIP_TRY(hipMalloc((void **)&d_rows, ch * s->dim * 4));
    SHODH_HIP_TRY(hipMalloc((void **)&d_codes, ch * s->M));
    SHODH_HIP_TRY(hipMalloc((void **)&d_assign, ch * 4));
    SHODH_HIP_TRY(hipMalloc((void **)&d_ad, ch * 4));
    SHODH_HIP_TRY(hipMalloc((void **)&d_ac, ch * 4));
    SHODH_HIP_TRY(hipMalloc((void **)&d_part, part_bytes + 256));
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
    hipFree(d_rows); hipFree(d_codes); hipFree(d_assign); hipFree(d_ad); hipFree(d_ac); hipFree(d_part);
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
    SHODH_HIP_TRY(hipMalloc((void **)&s->centroids, (size_t)P * cfg.dim * 4));
    SHODH_HIP_TRY(hipMalloc((void **)&s->codebook, (size_t)M * ncent * 8 * 4));
    SHODH_HIP_TRY(hipMalloc((void **)&s->list_off, (size_t)(P + 1) * 8));
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
    ~Buf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? SHODH_OK : SHODH_ERR_OOM; }
    template <class T> T *as() { return (T *)p; }
};

// members of every cluster in index order: stable radix sort of (key, index) on the low `bits` bits of the key
int sorted_members(const uint32_t *keys, uint64_t n, uint32_t k, uint32_t *idx_in, uint32_t *keys_out, uint32_t *members, Buf &tmp, size_t &tmp_bytes) {
    int bits = 1;
    while ((1u << bits) < k) ++bits;
    size_t need = 0;
    SHODH_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys, keys_out, idx_in, members, (int)n, 0, bits));
    if (need > tmp_bytes) { if (tmp.p) hipFree(tmp.p); tmp.p = nullptr; SHODH_TRY(tmp.alloc(need)); tmp_bytes = need; }
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

// index.hip -- the index object behind the C ABI: VamanaIndex / VectorIndexBackend method set
// (src/vector_db/vamana.rs:168-1645, src/vector_db/mod.rs:98-266) on one MI355X.
//
// HBM layout per index (one device):
//   rows    f32 [cap][dim]   row-major, the master copy (bit-for-bit what add/build received;
//                            extract_all_vectors returns it verbatim, retrieval.rs:2504-2516)
//   rows_h  f16 [cap][dim]   shadow copy fp16(256*x) for the MFMA pre-scan; tombstoned rows zeroed
//   deleted u32 [cap/32]     tombstone bitmask (vamana.rs:813-849 keeps a HashSet<u32>)
// ids are dense and sequential exactly like add_vector (vamana.rs:854-855): id = id_base + row.
#include <algorithm>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "common.h"
#include "combiner.h"

namespace shodh {

// ---- error string ------------------------------------------------------------------------------
static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

const char *last_error_of_this_thread() { return g_err.c_str(); }

int ensure_dynamic_lds(const void *fn, size_t bytes) {
    // hipFuncSetAttribute acts on the CURRENT device's copy of the function: the cache is keyed by (device, function), so
    // a second index on another GPU of the same process raises its own limit
    struct Seen { int dev; const void *fn; size_t bytes; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int dev = 0;
    SHODH_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    for (auto &e : seen)
        if (e.dev == dev && e.fn == fn) {
            if (e.bytes >= bytes) return SHODH_OK;
            SHODH_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            e.bytes = bytes;
            return SHODH_OK;
        }
    SHODH_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    seen.push_back(Seen{dev, fn, bytes});
    return SHODH_OK;
}

// ---- kernels living in other translation units ---------------------------------------------------
uint32_t topk_capacity(uint32_t k);
size_t exact_partial_bytes(uint32_t nq, uint32_t dim, uint32_t k, uint32_t grid_x);
uint32_t exact_grid_x(uint64_t n_rows, uint32_t nq, uint32_t k, int cus);
int launch_flat_exact(const float *rows, uint64_t n_rows, uint32_t dim, const uint32_t *deleted,
                      const float *d_queries, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                      uint64_t *partial, uint32_t grid_x, uint32_t *d_ids, float *d_dist, uint32_t *d_counts,
                      const uint32_t *qlist, const uint32_t *qcount, hipStream_t st);
int launch_flat_exact_arrive(const float *rows, uint64_t n_rows, uint32_t dim, const uint32_t *deleted,
                             const float *d_queries, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                             uint64_t *partial, uint32_t grid_x, uint32_t *d_ids, float *d_dist, uint32_t *d_counts,
                             const uint32_t *qlist, const uint32_t *qcount, uint32_t *arrive, hipStream_t st);

int launch_merge_lists(const uint32_t *in_ids, const float *in_dist, uint64_t list_stride, uint32_t n_lists, uint32_t nq, uint32_t k,
                       uint32_t *ids, float *dist, uint32_t *counts, hipStream_t st);

struct MfmaPlan {
    uint32_t passes, n_slots, ksteps;
    uint32_t tile_stride, n_sel_tiles, J;
    uint64_t n_tiles;
    uint32_t cand_cap, fcap, topk_cap;
    int grid_x;
    uint32_t set_only;      // (scan_mfma.hip: FinalArgs::set_only)
};
bool mfma_supported(uint32_t dim);
MfmaPlan mfma_plan(uint64_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int cus);
size_t mfma_workspace_bytes(const MfmaPlan &p, uint32_t dim, size_t *offs);
bool solo_supported(uint32_t nq, uint32_t k, uint64_t n_rows, int cus, const MfmaPlan &p);
int launch_solo_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim, const uint32_t *deleted, const float *d_q,
                         uint32_t k, uint32_t order, uint32_t id_base, float maxnorm, float maxres, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs,
                         uint32_t *solo_cnt, int cus, uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                         hipEvent_t ev_scan_done, hipEvent_t ev_select_done, hipEvent_t ev0, hipEvent_t ev1, uint32_t *stats_ext, uint32_t *stats_mirror);
bool probe_select_supported(uint64_t n_rows, uint32_t dim, uint32_t k, uint32_t order, bool has_deleted, const MfmaPlan &p);
int launch_probe_select_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim, const float *d_q, uint32_t nq, uint32_t k, uint32_t id_base,
                                 float maxnorm, float maxres, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs, uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                                 uint32_t *stats_ext);
int launch_mfma_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim,
                         const uint32_t *deleted, const float *d_q, uint32_t nq, uint32_t k, uint32_t order,
                         uint32_t id_base, float maxnorm, float maxres, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs,
                         uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                         hipEvent_t ev_scan_done, hipEvent_t ev_select_done, hipEvent_t ev_emit0, hipEvent_t ev_emit1, uint32_t *stats_ext, uint32_t *stats_mirror);
int launch_convert_rows(const float *rows, uint64_t first, uint64_t n, uint32_t dim, _Float16 *rows_h, uint32_t *stats, hipStream_t st);
int launch_count_nonfinite(const float *x, uint64_t n, uint32_t *counter, hipStream_t st);
int launch_shadow_set_row(const float *rows, _Float16 *rows_h, uint64_t row, uint32_t dim, int zero, hipStream_t st);
int launch_shadow_zero_rows(_Float16 *rows_h, const uint32_t *d_list, uint64_t n, uint32_t dim, hipStream_t st);
int launch_shadow_restore_deleted(const float *rows, _Float16 *rows_h, const uint32_t *deleted, uint64_t n, uint32_t dim, hipStream_t st);

// vamana_graph.hip
struct VgGraph { const float *rows; uint32_t dim; uint32_t *deg; uint32_t *nbr; uint32_t stride; uint32_t order; };
struct VgSearchArgs { VgGraph g; uint32_t n, medoid; const float *q; uint32_t nq, k, search_k; const uint32_t *deleted; uint32_t id_base;
                      uint32_t *visited; uint32_t vis_words; uint32_t *ids; float *dist; uint32_t *counts; uint32_t *overflow; };
struct VgInsertArgs { VgGraph g; uint32_t first, count, R, medoid; uint32_t *visited; uint32_t *overflow; };
struct VgBuildArgs { VgGraph g; uint32_t n, R, L, medoid; float alpha; uint32_t *visited; uint32_t *overflow; uint32_t first, count; uint32_t *updates; };
constexpr uint32_t VG_BUILD_CHUNK = 512;
struct VgRepairArgs { VgGraph g; uint32_t n, first, count, R, L, medoid; float alpha; uint32_t *visited; uint32_t *overflow; uint32_t *repaired; };
int vg_launch_repair(const VgRepairArgs &a, hipStream_t st);
int vg_launch_search(const VgSearchArgs &a, hipStream_t st);
int vg_launch_insert(const VgInsertArgs &a, hipStream_t st);
int vg_launch_build(const VgBuildArgs &a, hipStream_t st);
int vg_launch_centroid(const float *rows, uint32_t n, uint32_t dim, float *centroid, hipStream_t st);

struct IvfpqState;   // ivfpq.hip
void ivfpq_destroy(IvfpqState *s);
uint64_t ivfpq_len(IvfpqState *s);
size_t ivfpq_scratch_bytes(const IvfpqState *s, const shodh_index_cfg &cfg, uint32_t nq, uint32_t k);
int ivfpq_search(IvfpqState *s, const shodh_index_cfg &cfg, const float *d_q, uint32_t nq, uint32_t k,
                 uint32_t *d_ids, float *d_dist, uint32_t *d_counts, unsigned char *scratch, hipStream_t st);

// ---- per-search scratch ----------------------------------------------------------------------------
struct Workspace {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // start, scan done, select done, end
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the most recent search that used this workspace
    bool pending = false;                // ... and whether that search may still be running (device-pointer calls)
    unsigned char *buf = nullptr;
    size_t bytes = 0;
    uint32_t *solo_cnt = nullptr;        // single-query scan: its overflow counter, zero between calls (scan_mfma.hip, launch_solo_pipeline)
    // host-pointer calls: queries in, and ONE output block [ids | dist | counts] with a pinned host mirror, so that the results come back in
    // a single copy (three small copies plus their API calls were a visible part of a single query's latency)
    float *h_q = nullptr;                // one-query host calls: the query in pinned, device-visible host memory (the kernels read it from there: no H2D copy command)
    float *h_qb = nullptr; size_t h_qb_floats = 0;   // coalesced calls (combiner.h): the members' queries gathered in pinned memory, one H2D copy
    float *d_q = nullptr; uint32_t *d_out = nullptr, *h_out = nullptr; uint32_t *d_ids = nullptr; float *d_dist = nullptr; uint32_t *d_counts = nullptr, *d_stats = nullptr;
    size_t q_floats = 0, out_words = 0;
    // ring of (start, end) events around the dominant scan kernel of each search, for
    // shodh_index_kernel_timing (bench.py roofline): slot = ring_pos % RING
    static constexpr uint32_t RING = 256;
    hipEvent_t ring[RING][2];
    uint32_t ring_pos = 0;

    int init() {
        SHODH_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (auto &e : ev) SHODH_HIP_TRY(hipEventCreate(&e));
        SHODH_HIP_TRY(hipEventCreateWithFlags(&last_use, hipEventDisableTiming));
        for (auto &r : ring) { r[0] = nullptr; r[1] = nullptr; }      // created on first use (enqueue_flat): 64 concurrent callers used to mean 64 x 512 events up front
        SHODH_HIP_TRY(dev_alloc((void **)&solo_cnt, 256 + 4096));      // + 1024 arrival counters of the exact fallback's in-scan merge (flat_exact.hip, FLAT_ARRIVE_WORDS)
        // hipMemset on device memory is not synchronous with the host and the workspace's stream is non-blocking: without the synchronisation the FIRST search
        // on a new workspace could start before the counter was cleared and have it zeroed under its feet -- survivors lost, a wrong list (seen once in
        // 6400 calls with 64 threads each creating their workspace while the device was busy: bench.py concurrent_callers, round 5)
        SHODH_HIP_TRY(hipMemsetAsync(solo_cnt, 0, 256 + 4096, stream));
        SHODH_HIP_TRY(hipStreamSynchronize(stream));
        SHODH_HIP_TRY(pin_alloc((void **)&h_q, 4096));
        return SHODH_OK;
    }
    int reserve(size_t need) {
        if (need <= bytes) return SHODH_OK;
        if (buf) SHODH_HIP_TRY(dev_free(buf));
        buf = nullptr; bytes = 0;
        SHODH_HIP_TRY(dev_alloc((void **)&buf, need));
        bytes = need;
        return SHODH_OK;
    }
    int reserve_gather(size_t qf) {
        if (qf <= h_qb_floats) return SHODH_OK;
        if (h_qb) pin_free(h_qb);
        h_qb = nullptr; h_qb_floats = 0;
        SHODH_HIP_TRY(pin_alloc((void **)&h_qb, qf * 4));
        h_qb_floats = qf;
        return SHODH_OK;
    }
    int reserve_io(size_t qf, size_t oe, size_t nq) {
        if (qf > q_floats) { if (d_q) dev_free(d_q); d_q = nullptr; q_floats = 0; SHODH_HIP_TRY(dev_alloc((void **)&d_q, qf * 4)); q_floats = qf; }
        const size_t words = 2 * oe + nq + 8;          // + the four pipeline statistics (scan_stats), so that they come back in the same copy (+ the final stage's arrival counter)
        if (words > out_words) {
            if (d_out) dev_free(d_out); if (h_out) pin_free(h_out); d_out = nullptr; h_out = nullptr; out_words = 0;
            SHODH_HIP_TRY(dev_alloc((void **)&d_out, words * 4)); SHODH_HIP_TRY(pin_alloc((void **)&h_out, words * 4)); out_words = words;
        }
        d_ids = d_out; d_dist = reinterpret_cast<float *>(d_out + oe); d_counts = d_out + 2 * oe;      // contiguous for THIS call's sizes
        d_stats = d_out + 2 * oe + nq;
        return SHODH_OK;
    }
    void destroy() {
        if (buf) dev_free(buf);
        if (solo_cnt) dev_free(solo_cnt);
        if (d_q) dev_free(d_q); if (d_out) dev_free(d_out); if (h_out) pin_free(h_out); if (h_q) pin_free(h_q); if (h_qb) pin_free(h_qb);
        for (auto &e : ev) if (e) hipEventDestroy(e);
        if (last_use) hipEventDestroy(last_use);
        for (auto &r : ring) { if (r[0]) hipEventDestroy(r[0]); if (r[1]) hipEventDestroy(r[1]); }
        if (stream) hipStreamDestroy(stream);
    }
};

}  // namespace shodh

using namespace shodh;

struct shodh_index {
    shodh_index_cfg cfg{};
    int cus = 256;
    mutable std::shared_mutex mu;        // search: shared; add/build/delete: exclusive
    float *rows = nullptr;
    _Float16 *rows_h = nullptr;
    uint32_t *deleted = nullptr;         // device bitmask, sized for cap_rows
    uint32_t *stats = nullptr;           // device [4]: max norm^2, max |x|, non-finite count
    std::vector<uint32_t> deleted_host;  // host mirror of the bitmask
    uint64_t n = 0, cap_rows = 0, n_deleted = 0;
    float maxnorm = 0.0f, maxabs = 0.0f;
    float maxres = 0.0f;                 // the largest |row - its fp16 shadow row| (convert_rows_kernel): the corpus side of the pre-scan's error bound (scan_mfma.hip, eps_coefficients)
    bool quantizable = true;             // fp16 shadow usable (max |x| * 256 < 60000)
    bool shadow = false;                 // shadow copy maintained (dim supported by the MFMA kernel)
    std::mutex ws_mu;
    std::vector<Workspace *> ws_free;
    mutable std::mutex stat_mu;
    float last_us[4] = {0, 0, 0, 0};
    uint64_t last_stats[5] = {0, 0, 0, 0, 0};
    IvfpqState *ivfpq = nullptr;
    // SHODH_SCAN_GRAPH: the Vamana graph (vamana_graph.hip)
    uint32_t *g_deg = nullptr, *g_nbr = nullptr;   // [cap_rows], [cap_rows][g_stride]
    uint32_t *g_visited = nullptr;                 // [cap_rows / 32 + 1] visited bits of the insert / build walk
    uint32_t *g_overflow = nullptr;                // device flag: a frontier array overflowed (never in practice)
    bool g_overflowed = false;                     // sticky host copy for graph INSERTS (shodh_index_graph_overflowed): set by add, cleared by build
    uint32_t g_stride = 0, g_medoid = 0;
    uint64_t g_nodes = 0;                          // rows that have a node in the graph (== n when the graph is usable)
    // coalescing front for concurrent host-pointer searches of a few queries each (combiner.h): SHODH_COALESCE=0 / shodh_index_set_coalesce turn it off
    std::atomic<bool> coalesce{true};
    Combiner co;
    bool probe_set_mode = false;         // IVF-PQ's private centroid index: searches return the SET of the k nearest (final_stage_kernel, set mode)
};

namespace shodh {

void index_set_probe_set_mode(shodh_index *idx, bool on) { if (idx) idx->probe_set_mode = on; }

static int set_device(const shodh_index *idx) {
    SHODH_HIP_TRY(hipSetDevice(idx->cfg.device));
    return SHODH_OK;
}

static int grow(shodh_index *idx, uint64_t need_rows) {
    if (need_rows <= idx->cap_rows) return SHODH_OK;
    uint64_t nc = idx->cap_rows ? idx->cap_rows : 1024;
    while (nc < need_rows) nc *= 2;
    nc = (nc + 63) & ~63ull;
    const size_t dim = idx->cfg.dim;
    float *nr = nullptr; _Float16 *nh = nullptr; uint32_t *nd = nullptr;
    if (dev_alloc((void **)&nr, nc * dim * 4) != hipSuccess) { set_error("out of HBM growing index to %llu rows", (unsigned long long)nc); return SHODH_ERR_OOM; }
    if (idx->shadow && dev_alloc((void **)&nh, nc * dim * 2) != hipSuccess) { dev_free(nr); set_error("out of HBM (fp16 shadow)"); return SHODH_ERR_OOM; }
    if (dev_alloc((void **)&nd, (nc / 32 + 1) * 4) != hipSuccess) { dev_free(nr); if (nh) dev_free(nh); set_error("out of HBM (tombstones)"); return SHODH_ERR_OOM; }
    SHODH_HIP_TRY(hipMemset(nd, 0, (nc / 32 + 1) * 4));
    if (idx->n) {
        SHODH_HIP_TRY(hipMemcpy(nr, idx->rows, idx->n * dim * 4, hipMemcpyDeviceToDevice));
        if (nh) SHODH_HIP_TRY(hipMemcpy(nh, idx->rows_h, idx->n * dim * 2, hipMemcpyDeviceToDevice));
        SHODH_HIP_TRY(hipMemcpy(nd, idx->deleted, (idx->cap_rows / 32 + 1) * 4, hipMemcpyDeviceToDevice));
    }
    if (idx->cfg.scan_mode == SHODH_SCAN_GRAPH) {
        uint32_t *gd = nullptr, *gn = nullptr, *gv = nullptr;
        if (dev_alloc((void **)&gd, nc * 4) != hipSuccess || dev_alloc((void **)&gn, nc * (size_t)idx->g_stride * 4) != hipSuccess ||
            dev_alloc((void **)&gv, (nc / 32 + 1) * 4) != hipSuccess) {
            dev_free(gd); dev_free(gn); dev_free(gv); dev_free(nr); if (nh) dev_free(nh); dev_free(nd);      // nothing of the old slab was touched yet
            set_error("out of HBM (graph)"); return SHODH_ERR_OOM;
        }
        SHODH_HIP_TRY(hipMemset(gd, 0, nc * 4));
        if (idx->g_nodes) {
            SHODH_HIP_TRY(hipMemcpy(gd, idx->g_deg, idx->g_nodes * 4, hipMemcpyDeviceToDevice));
            SHODH_HIP_TRY(hipMemcpy(gn, idx->g_nbr, idx->g_nodes * (size_t)idx->g_stride * 4, hipMemcpyDeviceToDevice));
        }
        if (idx->g_deg) dev_free(idx->g_deg);
        if (idx->g_nbr) dev_free(idx->g_nbr);
        if (idx->g_visited) dev_free(idx->g_visited);
        idx->g_deg = gd; idx->g_nbr = gn; idx->g_visited = gv;
    }
    if (idx->rows) dev_free(idx->rows);
    if (idx->rows_h) dev_free(idx->rows_h);
    if (idx->deleted) dev_free(idx->deleted);
    idx->rows = nr; idx->rows_h = nh; idx->deleted = nd; idx->cap_rows = nc;
    idx->deleted_host.resize(nc / 32 + 1, 0);
    return SHODH_OK;
}

// after new rows [first, first+n) are in idx->rows: shadow copy + stats
static int finish_append(shodh_index *idx, uint64_t first, uint64_t n) {
    if (idx->shadow) {
        // the running maxima are restored if the batch is rejected: an Inf row would otherwise leave maxnorm / maxabs at
        // +Inf and silently disable the MFMA path for every later (valid) add
        uint32_t before[4];
        SHODH_HIP_TRY(hipMemcpy(before, idx->stats, sizeof(before), hipMemcpyDeviceToHost));
        SHODH_TRY(launch_convert_rows(idx->rows, first, n, idx->cfg.dim, idx->rows_h, idx->stats, nullptr));
        uint32_t st[4];
        SHODH_HIP_TRY(hipMemcpy(st, idx->stats, sizeof(st), hipMemcpyDeviceToHost));
        float nsq, ma, rsq;
        memcpy(&nsq, &st[0], 4); memcpy(&ma, &st[1], 4); memcpy(&rsq, &st[3], 4);
        if (st[2] != 0) {
            // roll back: the rows are not published (n is not advanced by the caller)
            before[2] = 0;
            hipMemcpy(idx->stats, before, sizeof(before), hipMemcpyHostToDevice);
            set_error("rows contain %u non-finite values (NaN/Inf are out of contract: MiniLM scrubs them, minilm.rs:847-851)", st[2]);
            return SHODH_ERR_NONFINITE;
        }
        idx->maxnorm = sqrtf(nsq) * 1.00001f;
        idx->maxres = sqrtf(rsq) * 1.0001f;
#ifdef SHODH_DIAG_MAXRES0      // (diagnostic builds, results INVALID: the corpus side of the error bound left out -- does a test notice? tools/build_variant.sh with VARIANT_SRC=index)
        idx->maxres = 0.0f;
#endif
        idx->maxabs = ma;
        idx->quantizable = (ma * 256.0f < 60000.0f);
    } else {
        // no shadow copy (dimension without an MFMA kernel): the same finite check, on the device, for host and device rows alike
        uint32_t bad = 0;
        SHODH_TRY(launch_count_nonfinite(idx->rows + first * idx->cfg.dim, n * idx->cfg.dim, idx->stats + 2, nullptr));
        SHODH_HIP_TRY(hipMemcpy(&bad, idx->stats + 2, 4, hipMemcpyDeviceToHost));
        if (bad) {
            uint32_t z = 0;
            hipMemcpy(idx->stats + 2, &z, 4, hipMemcpyHostToDevice);
            set_error("rows contain %u non-finite values (NaN/Inf are out of contract: MiniLM scrubs them, minilm.rs:847-851)", bad);
            return SHODH_ERR_NONFINITE;
        }
    }
    return SHODH_OK;
}

// A workspace that was last used on the same stream needs no synchronisation at all (stream order); one coming from
// another stream gets an event recorded on that stream now and waited for by the new one. (Recording an event after
// every search and waiting for it before the next cost ~5 us per search: every record is a packet between kernels.)
static Workspace *ws_acquire(shodh_index *idx, hipStream_t st, bool own_stream) {
    {
        std::lock_guard<std::mutex> g(idx->ws_mu);
        if (!idx->ws_free.empty()) {
            size_t pick = idx->ws_free.size() - 1;
            if (!own_stream)
                for (size_t i = idx->ws_free.size(); i-- > 0;)
                    if (!idx->ws_free[i]->pending || idx->ws_free[i]->last_stream == st) { pick = i; break; }
            Workspace *w = idx->ws_free[pick];
            idx->ws_free.erase(idx->ws_free.begin() + (long)pick);
            return w;
        }
    }
    Workspace *w = new Workspace();
    if (w->init() != SHODH_OK) { w->destroy(); delete w; return nullptr; }
    return w;
}
static void ws_release(shodh_index *idx, Workspace *w) {
    std::lock_guard<std::mutex> g(idx->ws_mu);
    idx->ws_free.push_back(w);
}

static bool use_mfma(const shodh_index *idx, uint32_t nq, uint32_t k) {
    if (idx->cfg.scan_mode == SHODH_SCAN_EXACT) return false;
    if (!idx->shadow || !idx->quantizable) return false;
    if (k == 0 || k > 2048) return false;
    if (idx->cfg.order == SHODH_ORDER_SEQ_1M) return idx->n >= 512 && idx->n >= 8ull * k;   // centroid tables: thousands of rows, many queries
    if (idx->n < 16384 || idx->n < 64ull * k) return false;     // pre-scan sampling needs a real corpus
    (void)nq;
    // AUTO == MFMA whenever the shadow copy is usable: measured at 1M rows a single query takes 260 us on the
    // fp16 pre-scan path (half the bytes) vs 346 us on the exact-order f32 scan; batches only widen the gap.
    return true;
}

// What enqueue_flat did, for the caller's bookkeeping after the stream has been synchronised
struct FlatCall {
    bool used_mfma = false;      // fp16 pre-scan pipeline (statistics exist)
    bool solo = false;           // ... its single-query form
    bool lean_events = false;    // stage timings come from the scan kernel's own event pair + ev[3] (no packets between the kernels)
    bool deferred = false;       // the exact scan of unresolved queries has NOT been enqueued: statistics word 2 says whether it is needed
    hipEvent_t k0 = nullptr, k1 = nullptr;
    size_t ws_bytes = 0; size_t offs[16]; uint32_t gx = 0;
    uint32_t sampled_rows = 0;
};

// the exact scan of the queries the pre-scan pipeline left in its fallback list (normally none)
static int enqueue_flat_fallback(shodh_index *idx, Workspace *w, const FlatCall &fc, const float *d_q, uint32_t nq, uint32_t k,
                                 uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st) {
    const uint32_t *fb_list = (const uint32_t *)(w->buf + fc.offs[6]);
    const uint32_t *fb_count = (const uint32_t *)(w->buf + fc.offs[7]);
    uint64_t *partial = (uint64_t *)(w->buf + ((fc.ws_bytes + 255) & ~(size_t)255));
    // (the scan merges its own partial lists -- arrival counters behind the single-query counter, zero between launches: one launch, not two, for a list that is nearly always empty)
    return launch_flat_exact_arrive(idx->rows, idx->n, idx->cfg.dim, idx->n_deleted ? idx->deleted : nullptr, d_q, nq, k, idx->cfg.order, (uint32_t)idx->cfg.id_base,
                                    partial, fc.gx, d_ids, d_dist, d_counts, fb_list, fb_count, w->solo_cnt + 64, st);
}

// enqueue a FLAT search on `st` using workspace w (device in/out pointers). host_call: the caller synchronises the stream and reads
// `stats_ext` (four words in its output block) afterwards, so the stage events are recorded and a single query's exact fallback can wait
// for that look at the statistics instead of costing two empty launches on every call.
static int enqueue_flat(shodh_index *idx, Workspace *w, const float *d_q, uint32_t nq, uint32_t k,
                        uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st, FlatCall *fc, bool host_call, uint32_t *stats_ext, uint32_t *stats_mirror = nullptr) {
    // events: the per-stage ones only for host-pointer calls (their timings are read back after the call's own
    // synchronisation); the pair around the scan kernel unless SHODH_KERNEL_EVENTS=0. Each record is a packet in the
    // stream between two kernels.
    static const bool kernel_events = !(getenv("SHODH_KERNEL_EVENTS") && atoi(getenv("SHODH_KERNEL_EVENTS")) == 0);
    const uint32_t dim = idx->cfg.dim;
    const uint32_t idb = (uint32_t)idx->cfg.id_base;
    const uint32_t *del = idx->n_deleted ? idx->deleted : nullptr;
    fc->used_mfma = use_mfma(idx, nq, k);
    MfmaPlan p{};
    if (fc->used_mfma) { p = mfma_plan(idx->n, dim, nq, k, idx->cus); p.set_only = idx->probe_set_mode ? 1u : 0u; }
    // IVF probe selection on the private centroid index: scores of the whole table + one wave per query (scan_mfma.hip, probe_select_kernel)
    const bool psel = fc->used_mfma && probe_select_supported(idx->n, dim, k, idx->cfg.order, del != nullptr, p);
    fc->solo = fc->used_mfma && !psel && solo_supported(nq, k, idx->n, idx->cus, p);
    hipEvent_t *rk = w->ring[w->ring_pos % Workspace::RING];
    hipEvent_t rk0 = nullptr, rk1 = nullptr;
    if (kernel_events) {
        if (!rk[0]) { SHODH_HIP_TRY(hipEventCreate(&rk[0])); SHODH_HIP_TRY(hipEventCreate(&rk[1])); }
        rk0 = rk[0]; rk1 = rk[1]; w->ring_pos++;
    }
    fc->k0 = rk0; fc->k1 = rk1;
    fc->lean_events = host_call && fc->solo && kernel_events;
    const bool stage_events = host_call && !fc->lean_events;
    if (stage_events) SHODH_HIP_TRY(hipEventRecord(w->ev[0], st));
    if (fc->used_mfma) {
        fc->ws_bytes = mfma_workspace_bytes(p, dim, fc->offs);
        fc->gx = exact_grid_x(idx->n, nq, k, idx->cus);
        fc->sampled_rows = fc->solo ? 0u : p.n_sel_tiles * 64u;
        const size_t part_bytes = exact_partial_bytes(nq, dim, k, fc->gx);
        SHODH_TRY(w->reserve(fc->ws_bytes + part_bytes + 256));
        if (psel) {
            fc->sampled_rows = 0;
            SHODH_TRY(launch_probe_select_pipeline(idx->rows, idx->rows_h, idx->n, dim, d_q, nq, k, idb, idx->maxnorm, idx->maxres, p, w->buf, fc->offs, d_ids, d_dist, d_counts, st, stats_ext));
            if (stage_events) { SHODH_HIP_TRY(hipEventRecord(w->ev[1], st)); SHODH_HIP_TRY(hipEventRecord(w->ev[2], st)); }
        } else if (fc->solo) {     // one query: a single pass over the shadow copy with workgroup-local thresholds
            SHODH_TRY(launch_solo_pipeline(idx->rows, idx->rows_h, idx->n, dim, del, d_q, k, idx->cfg.order, idb, idx->maxnorm, idx->maxres, p, w->buf, fc->offs,
                                           w->solo_cnt, idx->cus, d_ids, d_dist, d_counts, st, stage_events ? w->ev[1] : nullptr,
                                            stage_events ? w->ev[2] : nullptr, rk0, rk1, stats_ext, stats_mirror));
        } else {
            SHODH_TRY(launch_mfma_pipeline(idx->rows, idx->rows_h, idx->n, dim, del, d_q, nq, k, idx->cfg.order, idb,
                                           idx->maxnorm, idx->maxres, p, w->buf, fc->offs, d_ids, d_dist, d_counts, st, stage_events ? w->ev[1] : nullptr,
                                            stage_events ? w->ev[2] : nullptr, rk0, rk1, stats_ext, stats_mirror));
        }
        // exact scan of whatever the pre-scan could not settle (device-side list; normally empty). A host-pointer call looks at the
        // statistics after its synchronisation and enqueues it only then, if at all.
        fc->deferred = host_call && stats_ext != nullptr;
        if (!fc->deferred && !psel) SHODH_TRY(enqueue_flat_fallback(idx, w, *fc, d_q, nq, k, d_ids, d_dist, d_counts, st));      // (probe selection settles every query itself)
    } else {
        const uint32_t gx = exact_grid_x(idx->n, nq, k, idx->cus);
        SHODH_TRY(w->reserve(exact_partial_bytes(nq, dim, k, gx) + 256));
        if (rk0) SHODH_HIP_TRY(hipEventRecord(rk0, st));
        SHODH_TRY(launch_flat_exact(idx->rows, idx->n, dim, del, d_q, nq, k, idx->cfg.order, idb, (uint64_t *)w->buf, gx,
                                    d_ids, d_dist, d_counts, nullptr, nullptr, st));
        if (rk1) SHODH_HIP_TRY(hipEventRecord(rk1, st));     // scan + merge (the merge is a few microseconds)
        if (stage_events) { SHODH_HIP_TRY(hipEventRecord(w->ev[1], st)); SHODH_HIP_TRY(hipEventRecord(w->ev[2], st)); }
    }
    if (host_call) SHODH_HIP_TRY(hipEventRecord(w->ev[3], st));
    return SHODH_OK;
}

static void collect_timings(shodh_index *idx, Workspace *w, const FlatCall *fc) {
    float scan = 0, sel = 0, tot = 0;
    if (fc && fc->lean_events) {
        hipEventElapsedTime(&scan, fc->k0, fc->k1);
        hipEventElapsedTime(&tot, fc->k0, w->ev[3]);
        sel = tot - scan;
    } else {
        hipEventElapsedTime(&scan, w->ev[0], w->ev[1]);
        hipEventElapsedTime(&sel, w->ev[1], w->ev[2]);
        hipEventElapsedTime(&tot, w->ev[0], w->ev[3]);
    }
    std::lock_guard<std::mutex> g(idx->stat_mu);
    idx->last_us[0] = scan * 1000.0f;
    idx->last_us[1] = sel * 1000.0f;
    idx->last_us[2] = (tot - scan - sel) * 1000.0f;
    idx->last_us[3] = tot * 1000.0f;
}

}  // namespace shodh

// =====================================================================================================
extern "C" {

const char *shodh_last_error(void) { return g_err.c_str(); }
int shodh_abi_version(void) { return SHODH_HIP_ABI_VERSION; }

int shodh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { set_error("hipGetDeviceCount failed (no ROCm device visible)"); return SHODH_ERR_DEVICE; }
    return n;
}

int shodh_device_info(int dev, char *name, size_t cap, uint64_t *hbm_bytes, uint32_t *compute_units) {
    hipDeviceProp_t p;
    SHODH_HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && cap) snprintf(name, cap, "%s (%s)", p.name, p.gcnArchName);
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (compute_units) *compute_units = (uint32_t)p.multiProcessorCount;
    return SHODH_OK;
}

void shodh_index_cfg_default(shodh_index_cfg *cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->dim = 384;                       // BackendConfig::default (vector_db/mod.rs:74-86)
    cfg->metric = SHODH_METRIC_NDP;
    cfg->kind = SHODH_INDEX_FLAT;
    cfg->order = SHODH_ORDER_SCALAR4;
    cfg->device = 0;
    cfg->scan_mode = SHODH_SCAN_AUTO;
    cfg->reserve_rows = 0;
    cfg->id_base = 0;
    cfg->nprobe = 20;                     // BackendConfig.spann_probes
    cfg->max_degree = 32;                 // VamanaConfig::default (vamana.rs:79-90)
    cfg->search_list_size = 75;
    cfg->alpha = 1.2f;
}

int shodh_index_create(const shodh_index_cfg *cfg, shodh_index **out) {
    if (!cfg || !out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    *out = nullptr;
    if (cfg->dim == 0 || cfg->dim > 4096) { set_error("dimension %u out of range", cfg->dim); return SHODH_ERR_DIM; }
    if (cfg->kind != SHODH_INDEX_FLAT && cfg->kind != SHODH_INDEX_IVFPQ) { set_error("unknown index kind %u", cfg->kind); return SHODH_ERR_INVALID; }
    if (cfg->order > SHODH_ORDER_SEQ_1M) { set_error("unknown accumulation order %u", cfg->order); return SHODH_ERR_INVALID; }
    if (cfg->kind == SHODH_INDEX_FLAT && cfg->metric != SHODH_METRIC_NDP) {
        // RetrievalEngine refuses anything else (retrieval.rs:188-193)
        set_error("FLAT index requires NormalizedDotProduct (vectors are L2-normalised by the embedder)");
        return SHODH_ERR_INVALID;
    }
    if (cfg->kind == SHODH_INDEX_IVFPQ && cfg->dim % 8 != 0) { set_error("IVF-PQ needs dim %% 8 == 0 (pq.rs:43-48)"); return SHODH_ERR_DIM; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: libshodh_hip has no CPU fallback"); return SHODH_ERR_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) { set_error("device %d not present (%d visible)", cfg->device, ndev); return SHODH_ERR_DEVICE; }
    hipDeviceProp_t prop;
    SHODH_HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_error("device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName); return SHODH_ERR_DEVICE; }
    SHODH_HIP_TRY(hipSetDevice(cfg->device));
    shodh_index *idx = new shodh_index();
    idx->cfg = *cfg;
    idx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *cv = getenv("SHODH_COALESCE")) idx->coalesce = atoi(cv) != 0;
    if (const char *lv = getenv("SHODH_COALESCE_LINGER_US")) idx->co.linger_us = (uint32_t)atoi(lv);
    if (const char *qv = getenv("SHODH_COALESCE_QUIET_US")) idx->co.quiet_us = (uint32_t)atoi(qv);      // 0 = wait out the whole linger
    if (const char *tv2 = getenv("SHODH_COALESCE_TRACE")) idx->co.trace = atoi(tv2) != 0;
    if (cfg->scan_mode > SHODH_SCAN_GRAPH) { delete idx; set_error("unknown scan mode %u", cfg->scan_mode); return SHODH_ERR_INVALID; }
    if (cfg->scan_mode == SHODH_SCAN_GRAPH) {
        if (cfg->kind != SHODH_INDEX_FLAT || cfg->order > SHODH_ORDER_AVX2 || cfg->dim % 8 != 0 || cfg->max_degree == 0 || cfg->max_degree > 126 ||
            cfg->search_list_size == 0 || cfg->search_list_size > 1000 || cfg->id_base != 0) {
            delete idx;
            set_error("SHODH_SCAN_GRAPH needs a FLAT index, scalar-4 or AVX2 order, dim %% 8 == 0, 1 <= max_degree <= 126, search_list_size <= 1000, id_base 0");
            return SHODH_ERR_UNSUPPORTED;
        }
        idx->g_stride = cfg->max_degree + 1;
        if (dev_alloc((void **)&idx->g_overflow, 64) != hipSuccess) { delete idx; set_error("hipMalloc failed"); return SHODH_ERR_OOM; }
        hipMemset(idx->g_overflow, 0, 64);
    }
    idx->shadow = (cfg->kind == SHODH_INDEX_FLAT) && cfg->scan_mode != SHODH_SCAN_GRAPH && mfma_supported(cfg->dim);
    if (dev_alloc((void **)&idx->stats, 16) != hipSuccess) { delete idx; set_error("hipMalloc failed"); return SHODH_ERR_OOM; }
    hipMemset(idx->stats, 0, 16);
    if (cfg->kind == SHODH_INDEX_FLAT && cfg->reserve_rows) {
        int s = grow(idx, cfg->reserve_rows);
        if (s != SHODH_OK) { shodh_index_destroy(idx); return s; }
    }
    *out = idx;
    return SHODH_OK;
}

void shodh_index_destroy(shodh_index *idx) {
    if (!idx) return;
    hipSetDevice(idx->cfg.device);
    hipDeviceSynchronize();
    for (Workspace *w : idx->ws_free) { w->destroy(); delete w; }
    if (idx->ivfpq) ivfpq_destroy(idx->ivfpq);
    if (idx->rows) dev_free(idx->rows);
    if (idx->rows_h) dev_free(idx->rows_h);
    if (idx->deleted) dev_free(idx->deleted);
    if (idx->stats) dev_free(idx->stats);
    if (idx->g_deg) dev_free(idx->g_deg);
    if (idx->g_nbr) dev_free(idx->g_nbr);
    if (idx->g_visited) dev_free(idx->g_visited);
    if (idx->g_overflow) dev_free(idx->g_overflow);
    delete idx;
}

static VgGraph graph_of(const shodh_index *idx) {
    return VgGraph{idx->rows, idx->cfg.dim, idx->g_deg, idx->g_nbr, idx->g_stride, idx->cfg.order};
}

// the walk keeps at most VG_C_CAP frontier entries; more can only be LIVE when thousands of rows tie with the worst of the beam
static int check_graph_overflow(shodh_index *idx) {
    uint32_t o = 0;
    SHODH_HIP_TRY(hipMemcpy(&o, idx->g_overflow, 4, hipMemcpyDeviceToHost));
    if (!o) return SHODH_OK;
    hipMemset(idx->g_overflow, 0, 4);
    set_error("graph walk: the frontier overflowed (thousands of equidistant rows); the answer may differ from the reference's");
    return SHODH_ERR_UNSUPPORTED;
}

static int add_impl(shodh_index *idx, const float *rows, uint64_t n, uint32_t *first_id_out, hipMemcpyKind kind, bool skip_graph = false) {
    if (!idx || (!rows && n)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (idx->cfg.kind != SHODH_INDEX_FLAT) { set_error("add is for FLAT indexes; IVF-PQ uses shodh_index_ivfpq_insert"); return SHODH_ERR_STATE; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    if (idx->n + n + idx->cfg.id_base > 0xFFFFFFFEull) { set_error("vector ids are u32: index full"); return SHODH_ERR_INVALID; }
    if (first_id_out) *first_id_out = (uint32_t)(idx->cfg.id_base + idx->n);
    if (n == 0) return SHODH_OK;
    SHODH_TRY(grow(idx, idx->n + n));
    SHODH_HIP_TRY(hipMemcpy(idx->rows + idx->n * idx->cfg.dim, rows, n * idx->cfg.dim * 4, kind));
    SHODH_TRY(finish_append(idx, idx->n, n));
    if (idx->cfg.scan_mode == SHODH_SCAN_GRAPH && !skip_graph) {
        // add_vector (vamana.rs:853-974), one row after the other: each insert walks the graph the previous one left
        if (idx->g_nodes != idx->n) { set_error("the graph does not cover the rows (%llu nodes, %llu rows): build it first", (unsigned long long)idx->g_nodes, (unsigned long long)idx->n); return SHODH_ERR_STATE; }
        for (uint64_t at = 0; at < n; at += 4096) {               // bounded launches: one wave inserts row after row
            VgInsertArgs a{graph_of(idx), (uint32_t)(idx->n + at), (uint32_t)std::min<uint64_t>(4096, n - at), idx->cfg.max_degree, idx->g_medoid, idx->g_visited, idx->g_overflow};
            SHODH_TRY(vg_launch_insert(a, nullptr));
        }
        idx->g_nodes += n;
    }
    SHODH_HIP_TRY(hipDeviceSynchronize());
    idx->n += n;
    if (idx->cfg.scan_mode == SHODH_SCAN_GRAPH && !skip_graph && check_graph_overflow(idx) != SHODH_OK) {
        // The rows and their graph nodes ARE in the index at this point (a walk cannot be undone), so the call SUCCEEDS -- an error status made the
        // wrappers skip their bookkeeping (incremental-insert counters) and invited a retry that would add the rows twice. What happened stays
        // queryable: shodh_index_graph_overflowed() (sticky until the next build), and the message is left in shodh_last_error().
        idx->g_overflowed = true;
        set_error("graph insert: a walk's frontier overflowed (thousands of equidistant rows), so the graph may differ from the reference's; the %llu rows were added (ids from %llu)",
                  (unsigned long long)n, (unsigned long long)(idx->cfg.id_base + idx->n - n));
    }
    return SHODH_OK;
}

int shodh_index_graph_overflowed(const shodh_index *idx) { return idx && idx->g_overflowed ? 1 : 0; }

int shodh_index_add(shodh_index *idx, const float *rows, uint64_t n, uint32_t *first_id_out) {
    return add_impl(idx, rows, n, first_id_out, hipMemcpyHostToDevice);
}
int shodh_index_add_device(shodh_index *idx, const float *d_rows, uint64_t n, uint32_t *first_id_out) {
    return add_impl(idx, d_rows, n, first_id_out, hipMemcpyDeviceToDevice);
}

static int build_impl(shodh_index *idx, const float *rows, uint64_t n, hipMemcpyKind kind, bool construct_graph = true) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    {
        std::unique_lock<std::shared_mutex> lk(idx->mu);
        SHODH_TRY(set_device(idx));
        SHODH_HIP_TRY(hipDeviceSynchronize());
        idx->n = 0;
        idx->n_deleted = 0;
        idx->maxnorm = 0; idx->maxres = 0; idx->maxabs = 0; idx->quantizable = true;
        if (idx->deleted) SHODH_HIP_TRY(hipMemset(idx->deleted, 0, (idx->cap_rows / 32 + 1) * 4));
        std::fill(idx->deleted_host.begin(), idx->deleted_host.end(), 0u);
        SHODH_HIP_TRY(hipMemset(idx->stats, 0, 16));
        idx->g_nodes = 0; idx->g_medoid = 0; idx->g_overflowed = false;
    }
    if (idx->cfg.scan_mode == SHODH_SCAN_GRAPH) {
        // VamanaIndex::build: store the rows, then construct the graph from a random start (vamana.rs:200-284); the start is drawn here
        // from a fixed seed (the reference uses thread_rng) -- shodh_index_vamana_build takes an explicit start or another seed
        SHODH_TRY(add_impl(idx, rows, n, nullptr, kind, true));
        return construct_graph ? shodh_index_vamana_build(idx, 0x5EEDull, nullptr, nullptr, 0) : SHODH_OK;
    }
    return add_impl(idx, rows, n, nullptr, kind);
}
int shodh_index_build(shodh_index *idx, const float *rows, uint64_t n) { return build_impl(idx, rows, n, hipMemcpyHostToDevice); }
int shodh_index_build_device(shodh_index *idx, const float *d_rows, uint64_t n) { return build_impl(idx, d_rows, n, hipMemcpyDeviceToDevice); }

// One caller's part of a search: its queries, its k, its output pointers. A host-pointer call normally is one segment; a coalesced pass
// (combiner.h) carries one segment per member.
struct SearchSeg { const float *q; uint32_t nq, k; uint32_t *ids; float *dist; uint32_t *counts; };

// The segments are searched as ONE batch of nq = sum(nq) queries with k = max(k): a row's top-k' is the first k' entries of its top-k for every
// k' <= k (the order (dist total_cmp, id) is total and both are the exact answer), so a member with a smaller k gets the prefix of its rows.
// Device-pointer calls (sync_host = false) have one segment.
static int search_common(shodh_index *idx, const SearchSeg *segs, size_t n_segs, hipStream_t user_stream, bool sync_host, bool force_exact = false) {
    if (!idx || !segs || n_segs == 0 || (n_segs > 1 && !sync_host)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    uint32_t nq = 0, k = 0;
    for (size_t i = 0; i < n_segs; ++i) {
        const SearchSeg &g = segs[i];
        if ((g.nq && (!g.q || !g.counts)) || (g.nq && g.k && (!g.ids || !g.dist))) { set_error("null argument"); return SHODH_ERR_INVALID; }
        nq += g.nq;
        if (g.nq && g.k > k) k = g.k;
    }
    if (nq == 0) return SHODH_OK;
    const float *q = segs[0].q;
    uint32_t *ids = segs[0].ids; float *dist = segs[0].dist; uint32_t *counts = segs[0].counts;     // (used as such by single-segment calls only)
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    const uint32_t dim = idx->cfg.dim;
    const bool empty = (idx->cfg.kind == SHODH_INDEX_FLAT) ? (idx->n == 0) : (idx->ivfpq == nullptr);
    if (idx->cfg.kind == SHODH_INDEX_IVFPQ && !idx->ivfpq) {
        // SpannIndex::search on an unbuilt index: centroids empty -> Ok(vec![]) (spann.rs:575-578)
    }
    if (empty || k == 0) {
        if (sync_host) {
            for (size_t s = 0; s < n_segs; ++s) {
                const SearchSeg &g = segs[s];
                for (uint32_t i = 0; i < g.nq; ++i) g.counts[i] = 0;
                for (size_t i = 0; i < (size_t)g.nq * g.k; ++i) { g.ids[i] = 0xFFFFFFFFu; g.dist[i] = INFINITY; }
            }
        } else {
            SHODH_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)nq * 4, user_stream));
            if (k) { SHODH_HIP_TRY(hipMemsetAsync(ids, 0xFF, (size_t)nq * k * 4, user_stream)); SHODH_HIP_TRY(hipMemsetAsync(dist, 0x7F, (size_t)nq * k * 4, user_stream)); }
        }
        return SHODH_OK;
    }
    if (sync_host) {
        for (size_t s = 0; s < n_segs; ++s)
            for (size_t i = 0; i < (size_t)segs[s].nq * dim; ++i)
                if (!(fabsf(segs[s].q[i]) <= 3.0e38f)) { set_error("query contains non-finite values"); return SHODH_ERR_NONFINITE; }
    }
    Workspace *w = ws_acquire(idx, user_stream, sync_host);
    if (!w) return SHODH_ERR_DEVICE;
    hipStream_t st = sync_host ? w->stream : user_stream;
    int rc = SHODH_OK;
    FlatCall fc;
    do {
        if (w->pending && w->last_stream != st) {
            // the previous user may still be running on another stream
            if (hipEventRecord(w->last_use, w->last_stream) != hipSuccess || hipStreamWaitEvent(st, w->last_use, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (hipDeviceSynchronize() != hipSuccess) { set_error("cannot order the workspace after its previous use"); rc = SHODH_ERR_DEVICE; break; }
            }
        }
        const float *d_q = q;
        uint32_t *d_ids = ids; float *d_dist = dist; uint32_t *d_counts = counts;
        // One query through host pointers (what `recall` does): the query is placed in pinned host memory the kernels read directly and the result
        // block (<= 1 KiB) is written by the last kernel straight into the pinned mirror -- no H2D and no D2H copy command around ~160 us of kernels.
        // The four statistics words stay in device memory (they are updated with atomics) and the one final-stage workgroup mirrors them at its end.
        static const bool zc_off = getenv("SHODH_ZERO_COPY") && atoi(getenv("SHODH_ZERO_COPY")) == 0;
        const bool flat_scan = idx->cfg.kind == SHODH_INDEX_FLAT && (idx->cfg.scan_mode != SHODH_SCAN_GRAPH || force_exact);
        const bool zero_copy = sync_host && !zc_off && nq == 1 && (size_t)dim * 4 <= 4096 && flat_scan;
        // A FEW queries through host pointers on the pre-scan pipeline (a coalesced pass of <= 128 callers, round 5): the same idea -- the queries are read
        // from the pinned gather block by the two kernels that read them once (query conversion, final stage) and the final stage writes the rows straight
        // into the pinned mirror; its last workgroup mirrors the statistics. A copy command on either side of the kernels is a hand-over between the DMA
        // engine's queue and the compute queue (~15 us each way). Not for the exact-order scan: every workgroup of it reads the queries.
        static const uint32_t zc_max_nq = getenv("SHODH_ZERO_COPY_MAX_NQ") ? (uint32_t)atoi(getenv("SHODH_ZERO_COPY_MAX_NQ")) : 128u;
        const bool zero_copy_multi = sync_host && !zc_off && !zero_copy && nq > 1 && nq <= zc_max_nq && flat_scan && !force_exact && use_mfma(idx, nq, k) &&
                                     !solo_supported(nq, k, idx->n, idx->cus, mfma_plan(idx->n, dim, nq, k, idx->cus));
        uint32_t *stats_mirror = nullptr;
        if (sync_host) {
            if ((rc = w->reserve_io((size_t)nq * dim, (size_t)nq * k, nq)) != SHODH_OK) break;
            if (zero_copy) {
                memcpy(w->h_q, q, (size_t)dim * 4);
                const size_t oe = (size_t)k;
                d_q = w->h_q; d_ids = w->h_out; d_dist = reinterpret_cast<float *>(w->h_out + oe); d_counts = w->h_out + 2 * oe;
                stats_mirror = w->h_out + 2 * oe + nq;
                stats_mirror[0] = stats_mirror[1] = stats_mirror[2] = stats_mirror[3] = 0;       // (paths without a final stage leave them alone)
            } else {
                const float *src = q;
                if (n_segs > 1 || zero_copy_multi) {       // the members' queries gathered in pinned memory (one DMA, or none at all)
                    if ((rc = w->reserve_gather((size_t)nq * dim)) != SHODH_OK) break;
                    size_t at = 0;
                    for (size_t s = 0; s < n_segs; ++s) { memcpy(w->h_qb + at, segs[s].q, (size_t)segs[s].nq * dim * 4); at += (size_t)segs[s].nq * dim; }
                    src = w->h_qb;
                }
                if (zero_copy_multi) {
                    const size_t oe = (size_t)nq * k;
                    d_q = w->h_qb; d_ids = w->h_out; d_dist = reinterpret_cast<float *>(w->h_out + oe); d_counts = w->h_out + 2 * oe;
                    stats_mirror = w->h_out + 2 * oe + nq;
                    stats_mirror[0] = stats_mirror[1] = stats_mirror[2] = stats_mirror[3] = 0;
                } else {
                    if (hipMemcpyAsync(w->d_q, src, (size_t)nq * dim * 4, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("H2D copy of queries failed"); rc = SHODH_ERR_DEVICE; break; }
                    d_q = w->d_q; d_ids = w->d_ids; d_dist = w->d_dist; d_counts = w->d_counts;
                }
            }
        }
        if (idx->cfg.kind == SHODH_INDEX_FLAT && idx->cfg.scan_mode == SHODH_SCAN_GRAPH && !force_exact) {
            // VamanaIndex::search without SHODH_VECTOR_EXACT (vamana.rs:764-808)
            if (idx->g_nodes != idx->n) { set_error("Vamana graph not built. Call build() first or add more vectors."); rc = SHODH_ERR_STATE; break; }
            const uint64_t dc = idx->n_deleted;
            const uint64_t search_k = dc > 0 ? (uint64_t)k + (dc < 2ull * k ? dc : 2ull * k) : k;
            if (search_k > 1000) { set_error("graph search: k + tombstone over-fetch = %llu exceeds 1000", (unsigned long long)search_k); rc = SHODH_ERR_UNSUPPORTED; break; }
            const uint32_t vis_words = (uint32_t)(idx->n / 32 + 1);
            if ((rc = w->reserve((size_t)nq * vis_words * 4 + 256)) != SHODH_OK) break;
            if (sync_host) hipEventRecord(w->ev[0], st);
            VgSearchArgs a{graph_of(idx), (uint32_t)idx->n, idx->g_medoid, d_q, nq, k, (uint32_t)search_k, idx->n_deleted ? idx->deleted : nullptr,
                           (uint32_t)idx->cfg.id_base, (uint32_t *)w->buf, vis_words, d_ids, d_dist, d_counts, idx->g_overflow};
            rc = vg_launch_search(a, st);
            if (sync_host) { hipEventRecord(w->ev[1], st); hipEventRecord(w->ev[2], st); hipEventRecord(w->ev[3], st); }
        } else if (idx->cfg.kind == SHODH_INDEX_FLAT) rc = enqueue_flat(idx, w, d_q, nq, k, d_ids, d_dist, d_counts, st, &fc, sync_host, sync_host ? w->d_stats : nullptr, stats_mirror);
        else {
            if ((rc = w->reserve(ivfpq_scratch_bytes(idx->ivfpq, idx->cfg, nq, k) + 256)) != SHODH_OK) break;
            if (sync_host) hipEventRecord(w->ev[0], st);
            rc = ivfpq_search(idx->ivfpq, idx->cfg, d_q, nq, k, d_ids, d_dist, d_counts, w->buf, st);
            if (sync_host) { hipEventRecord(w->ev[1], st); hipEventRecord(w->ev[2], st); hipEventRecord(w->ev[3], st); }
        }
        if (rc != SHODH_OK) break;
        if (sync_host) {
            const size_t oe = (size_t)nq * k;
            const size_t out_bytes = (2 * oe + nq + 4) * 4;          // (the four statistics words ride along)
            const bool no_copy = zero_copy || zero_copy_multi;
            if (!no_copy && hipMemcpyAsync(w->h_out, w->d_out, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("D2H copy of results failed"); rc = SHODH_ERR_DEVICE; break; }
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { set_error("search failed on device: %s", hipGetErrorString(e)); rc = SHODH_ERR_DEVICE; break; }
            if (fc.deferred && w->h_out[2 * oe + nq + 2] != 0) {
                // some query could not be settled by the pre-scan (unquantisable values, thousands of near-duplicates, an unusable threshold): exact scan now
                if ((rc = enqueue_flat_fallback(idx, w, fc, d_q, nq, k, d_ids, d_dist, d_counts, st)) != SHODH_OK) break;
                if (!no_copy && hipMemcpyAsync(w->h_out, w->d_out, (2 * oe + nq) * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("D2H copy of results failed"); rc = SHODH_ERR_DEVICE; break; }
                e = hipStreamSynchronize(st);
                if (e != hipSuccess) { set_error("search failed on device: %s", hipGetErrorString(e)); rc = SHODH_ERR_DEVICE; break; }
            }
            const bool graph_call = idx->cfg.scan_mode == SHODH_SCAN_GRAPH && idx->cfg.kind == SHODH_INDEX_FLAT && !force_exact;
            if (n_segs == 1 && segs[0].k == k) {
                memcpy(ids, w->h_out, oe * 4); memcpy(dist, w->h_out + oe, oe * 4); memcpy(counts, w->h_out + 2 * oe, (size_t)nq * 4);
            } else {
                // fan the rows out to their callers: row r of the pass is row r - first of its member, the first k_member entries of it
                const uint32_t *o_ids = w->h_out, *o_cnt = w->h_out + 2 * oe;
                const float *o_dist = reinterpret_cast<const float *>(w->h_out + oe);
                size_t r = 0;
                for (size_t s = 0; s < n_segs; ++s) {
                    const SearchSeg &g = segs[s];
                    for (uint32_t j = 0; j < g.nq; ++j, ++r) {
                        memcpy(g.ids + (size_t)j * g.k, o_ids + r * k, (size_t)g.k * 4);
                        memcpy(g.dist + (size_t)j * g.k, o_dist + r * k, (size_t)g.k * 4);
                        g.counts[j] = o_cnt[r] < g.k ? o_cnt[r] : g.k;
                    }
                }
            }
            if (graph_call) {
                // a walk whose frontier overflowed says so in the top bit of its count (vg_search_kernel)
                bool ovf = false;
                for (uint32_t i = 0; i < nq; ++i) { ovf = ovf || (counts[i] & 0x80000000u); counts[i] &= 0x7FFFFFFFu; }
                if (ovf) { set_error("graph walk: the frontier overflowed (thousands of equidistant rows); the answer may differ from the reference's"); rc = SHODH_ERR_UNSUPPORTED; break; }
            }
            collect_timings(idx, w, idx->cfg.kind == SHODH_INDEX_FLAT && !graph_call ? &fc : nullptr);
            {
                const uint32_t *st4 = w->h_out + 2 * oe + nq;
                std::lock_guard<std::mutex> g(idx->stat_mu);
                if (fc.used_mfma) {
                    idx->last_stats[0] = fc.sampled_rows; idx->last_stats[1] = st4[0]; idx->last_stats[2] = st4[1]; idx->last_stats[3] = st4[2];
                    idx->last_stats[4] = st4[3];
                } else idx->last_stats[0] = idx->last_stats[1] = idx->last_stats[2] = idx->last_stats[3] = idx->last_stats[4] = 0;
            }
        }
    } while (0);
    // a single-query call that failed half way may have left its list counter non-zero (the final stage hands it back zeroed)
    if (rc != SHODH_OK && fc.solo) (void)hipMemsetAsync(w->solo_cnt, 0, 4, st);
    // ... and the arrival counters of the exact fallback's in-scan merge, should an enqueue have failed between its launch and its end (they are zero between complete launches)
    if (rc != SHODH_OK && fc.used_mfma) (void)hipMemsetAsync(w->solo_cnt + 64, 0, 4096, st);
    w->last_stream = st;
    w->pending = !(sync_host && rc == SHODH_OK);       // successful host-pointer calls end with a stream synchronisation
    ws_release(idx, w);
    return rc;
}

// ---- coalescing front (combiner.h) -------------------------------------------------------------------------------------------------------
// `recall` asks one query per call from many threads at once (recall.rs:512-513, retrieval.rs:912-918). Host-pointer searches of a few queries
// that arrive while a pass is in flight share the NEXT pass; a caller that is alone runs exactly the single call it always ran.
constexpr uint32_t CO_MAX_CALL_NQ = 32;      // calls with more queries than this already amortise the pass: they run on their own
constexpr uint32_t CO_MAX_PASS_NQ = 256;     // one MFMA pass (MF_BPAD)
static bool coalescable(const shodh_index *idx, const float *q, uint32_t nq, uint32_t k, const uint32_t *ids, const float *dist, const uint32_t *counts) {
    if (!idx || !idx->coalesce || !q || !ids || !dist || !counts || nq == 0 || nq > CO_MAX_CALL_NQ || k == 0 || k > 2048) return false;
    if (idx->cfg.kind == SHODH_INDEX_FLAT && idx->cfg.scan_mode == SHODH_SCAN_GRAPH) return false;       // a walk per query: nothing to share
    const size_t n = (size_t)nq * idx->cfg.dim;
    for (size_t i = 0; i < n; ++i) if (!(fabsf(q[i]) <= 3.0e38f)) return false;      // (the direct call reports it; one member's bad input must not fail the others)
    return true;
}
static int coalesced_search(shodh_index *idx, const float *q, uint32_t nq, uint32_t k, uint32_t *ids, float *dist, uint32_t *counts) {
    SearchSeg mine{q, nq, k, ids, dist, counts};
    std::string err;
    const int rc = idx->co.submit(&mine, nq, CO_MAX_PASS_NQ,
        [idx](const std::vector<void *> &reqs) {
            if (reqs.size() == 1) return search_common(idx, static_cast<const SearchSeg *>(reqs[0]), 1, nullptr, true);
            std::vector<SearchSeg> segs;
            segs.reserve(reqs.size());
            for (void *r : reqs) segs.push_back(*static_cast<const SearchSeg *>(r));
            return search_common(idx, segs.data(), segs.size(), nullptr, true);
        },
        []() { return std::string(last_error_of_this_thread()); }, &err);
    if (rc != SHODH_OK) set_error("%s", err.c_str());
    return rc;
}

int shodh_index_search(shodh_index *idx, const float *q, uint32_t nq, uint32_t k, uint32_t *ids, float *dist, uint32_t *counts) {
    if (coalescable(idx, q, nq, k, ids, dist, counts)) return coalesced_search(idx, q, nq, k, ids, dist, counts);
    const SearchSeg s{q, nq, k, ids, dist, counts};
    return search_common(idx, &s, 1, nullptr, true);
}
int shodh_index_brute_force_search(shodh_index *idx, const float *q, uint32_t nq, uint32_t k, uint32_t *ids, float *dist, uint32_t *counts) {
    if (idx && idx->cfg.kind != SHODH_INDEX_FLAT) { set_error("brute_force_search is for FLAT indexes"); return SHODH_ERR_STATE; }
    const SearchSeg s{q, nq, k, ids, dist, counts};
    return search_common(idx, &s, 1, nullptr, true, true);
}
int shodh_index_search_device(shodh_index *idx, const float *d_q, uint32_t nq, uint32_t k, uint32_t *d_ids, float *d_dist,
                              uint32_t *d_counts, void *stream) {
    const SearchSeg s{d_q, nq, k, d_ids, d_dist, d_counts};
    return search_common(idx, &s, 1, (hipStream_t)stream, false);
}

int shodh_index_set_coalesce(shodh_index *idx, int enabled, uint32_t linger_us) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    idx->coalesce = enabled != 0;
    idx->co.linger_us = linger_us;
    return SHODH_OK;
}
int shodh_index_coalesce_stats(shodh_index *idx, uint64_t *stats6, int reset) {
    if (!idx || !stats6) { set_error("null argument"); return SHODH_ERR_INVALID; }
    const CombinerStats c = idx->co.stats();
    stats6[0] = c.batches; stats6[1] = c.members; stats6[2] = c.max_members; stats6[3] = c.lingered; stats6[4] = c.exec_ns / 1000; stats6[5] = c.linger_ns / 1000;
    if (reset) idx->co.reset_stats();
    return SHODH_OK;
}

int shodh_index_mark_deleted(shodh_index *idx, uint32_t id, int *was_valid) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    const uint64_t local = (uint64_t)id - idx->cfg.id_base;
    if (id < idx->cfg.id_base || local >= idx->n) { if (was_valid) *was_valid = 0; return SHODH_OK; }   // vamana.rs:814-819 returns false
    if (was_valid) *was_valid = 1;
    uint32_t &word = idx->deleted_host[local >> 5];
    const uint32_t bit = 1u << (local & 31);
    if (word & bit) return SHODH_OK;
    word |= bit;
    idx->n_deleted++;
    SHODH_HIP_TRY(hipDeviceSynchronize());
    SHODH_HIP_TRY(hipMemcpy(idx->deleted + (local >> 5), &word, 4, hipMemcpyHostToDevice));
    if (idx->shadow) { SHODH_TRY(launch_shadow_set_row(idx->rows, idx->rows_h, local, idx->cfg.dim, 1, nullptr)); SHODH_HIP_TRY(hipDeviceSynchronize()); }
    return SHODH_OK;
}

int shodh_index_mark_deleted_batch(shodh_index *idx, const uint32_t *ids, uint64_t n, uint64_t *n_marked_out) {
    if (!idx || (!ids && n)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    std::vector<uint32_t> fresh;                 // local rows that become tombstones with this call
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t local = (uint64_t)ids[i] - idx->cfg.id_base;
        if (ids[i] < idx->cfg.id_base || local >= idx->n) continue;          // not ours / not there: mark_deleted returns false (vamana.rs:814-819)
        uint32_t &word = idx->deleted_host[local >> 5];
        const uint32_t bit = 1u << (local & 31);
        if (word & bit) continue;
        word |= bit;
        fresh.push_back((uint32_t)local);
    }
    if (n_marked_out) *n_marked_out = fresh.size();
    if (fresh.empty()) return SHODH_OK;
    idx->n_deleted += fresh.size();
    SHODH_HIP_TRY(hipDeviceSynchronize());
    SHODH_HIP_TRY(hipMemcpy(idx->deleted, idx->deleted_host.data(), (idx->cap_rows / 32 + 1) * 4, hipMemcpyHostToDevice));
    if (idx->shadow) {
        uint32_t *d_list = nullptr;
        SHODH_HIP_TRY(dev_alloc((void **)&d_list, fresh.size() * 4));
        hipError_t e = hipMemcpy(d_list, fresh.data(), fresh.size() * 4, hipMemcpyHostToDevice);
        int rc = e == hipSuccess ? launch_shadow_zero_rows(idx->rows_h, d_list, fresh.size(), idx->cfg.dim, nullptr) : SHODH_ERR_DEVICE;
        if (hipDeviceSynchronize() != hipSuccess) rc = SHODH_ERR_DEVICE;
        dev_free(d_list);
        if (rc != SHODH_OK) { set_error("tombstoning the shadow rows failed"); return rc; }
    }
    return SHODH_OK;
}

int shodh_index_is_deleted(const shodh_index *idx, uint32_t id) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    const uint64_t local = (uint64_t)id - idx->cfg.id_base;
    if (id < idx->cfg.id_base || local >= idx->n) return 0;
    return (idx->deleted_host[local >> 5] >> (local & 31)) & 1u;
}

uint64_t shodh_index_len(const shodh_index *idx) {
    if (!idx) return 0;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    return idx->cfg.kind == SHODH_INDEX_IVFPQ ? ivfpq_len(idx->ivfpq) : idx->n;
}
uint64_t shodh_index_deleted_count(const shodh_index *idx) {
    if (!idx) return 0;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    return idx->n_deleted;
}
float shodh_index_deletion_ratio(const shodh_index *idx) {
    if (!idx) return 0.0f;
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    if (idx->n == 0) return 0.0f;
    return (float)idx->n_deleted / (float)idx->n;    // vamana.rs:834-840
}
int shodh_index_needs_compaction(const shodh_index *idx) { return shodh_index_deletion_ratio(idx) >= 0.30f; }   // DELETION_RATIO_THRESHOLD

int shodh_index_clear_deleted(shodh_index *idx) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    if (idx->n_deleted == 0) return SHODH_OK;
    SHODH_HIP_TRY(hipDeviceSynchronize());
    if (idx->shadow) SHODH_TRY(launch_shadow_restore_deleted(idx->rows, idx->rows_h, idx->deleted, idx->n, idx->cfg.dim, nullptr));
    SHODH_HIP_TRY(hipDeviceSynchronize());
    SHODH_HIP_TRY(hipMemset(idx->deleted, 0, (idx->cap_rows / 32 + 1) * 4));
    SHODH_HIP_TRY(hipDeviceSynchronize());       // (hipMemset does not wait; the next search runs on a non-blocking stream)
    std::fill(idx->deleted_host.begin(), idx->deleted_host.end(), 0u);
    idx->n_deleted = 0;
    return SHODH_OK;
}

int shodh_index_extract_rows(const shodh_index *idx, uint64_t first, uint64_t n, float *out_rows) {
    if (!idx || (!out_rows && n)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    if (first + n > idx->n) { set_error("rows [%llu,%llu) out of range (len %llu)", (unsigned long long)first, (unsigned long long)(first + n), (unsigned long long)idx->n); return SHODH_ERR_INVALID; }
    if (n) SHODH_HIP_TRY(hipMemcpy(out_rows, idx->rows + first * idx->cfg.dim, n * idx->cfg.dim * 4, hipMemcpyDeviceToHost));
    return SHODH_OK;
}

int shodh_index_extract_live_rows(const shodh_index *idx, float *out_rows, uint32_t *ids_out, uint64_t cap, uint64_t *n_out) {
    if (!idx || !n_out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    const uint64_t live = idx->n - idx->n_deleted;
    *n_out = live;
    if (!out_rows) return SHODH_OK;
    if (cap < live) { set_error("buffer holds %llu rows, %llu live", (unsigned long long)cap, (unsigned long long)live); return SHODH_ERR_INVALID; }
    const uint32_t dim = idx->cfg.dim;
    uint64_t o = 0, run_start = 0;
    bool in_run = false;
    auto flush = [&](uint64_t end) -> int {
        if (!in_run) return SHODH_OK;
        SHODH_HIP_TRY(hipMemcpy(out_rows + o * dim, idx->rows + run_start * dim, (end - run_start) * dim * 4, hipMemcpyDeviceToHost));
        if (ids_out) for (uint64_t r = run_start; r < end; ++r) ids_out[o + (r - run_start)] = (uint32_t)(idx->cfg.id_base + r);
        o += end - run_start;
        in_run = false;
        return SHODH_OK;
    };
    for (uint64_t r = 0; r < idx->n; ++r) {
        const bool del = (idx->deleted_host[r >> 5] >> (r & 31)) & 1u;
        if (!del && !in_run) { in_run = true; run_start = r; }
        if (del && in_run) SHODH_TRY(flush(r));
    }
    SHODH_TRY(flush(idx->n));
    return SHODH_OK;
}

uint32_t shodh_index_dim(const shodh_index *idx) { return idx ? idx->cfg.dim : 0; }

int shodh_index_set_graph(shodh_index *idx, const uint32_t *deg, const uint32_t *nbr, uint32_t stride, uint32_t medoid) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (idx->cfg.scan_mode != SHODH_SCAN_GRAPH) { set_error("not a SHODH_SCAN_GRAPH index"); return SHODH_ERR_STATE; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    const uint64_t n = idx->n;
    if (n && (!deg || !nbr)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (n && medoid >= n) { set_error("medoid %u out of range", medoid); return SHODH_ERR_INVALID; }
    std::vector<uint32_t> hn((size_t)n * idx->g_stride, 0u);
    for (uint64_t i = 0; i < n; ++i) {
        if (deg[i] > idx->cfg.max_degree + 1 || deg[i] > stride) { set_error("node %llu has %u neighbours (max_degree %u)", (unsigned long long)i, deg[i], idx->cfg.max_degree); return SHODH_ERR_INVALID; }
        for (uint32_t j = 0; j < deg[i]; ++j) {
            const uint32_t v = nbr[(size_t)i * stride + j];
            if (v >= n) { set_error("node %llu: neighbour %u out of range", (unsigned long long)i, v); return SHODH_ERR_INVALID; }
            hn[(size_t)i * idx->g_stride + j] = v;
        }
    }
    SHODH_HIP_TRY(hipDeviceSynchronize());
    if (n) {
        SHODH_HIP_TRY(hipMemcpy(idx->g_deg, deg, n * 4, hipMemcpyHostToDevice));
        SHODH_HIP_TRY(hipMemcpy(idx->g_nbr, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    }
    idx->g_medoid = medoid;
    idx->g_nodes = n;
    return SHODH_OK;
}

int shodh_index_get_graph(const shodh_index *idx, uint32_t *deg, uint32_t *nbr, uint32_t stride, uint32_t *medoid) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (idx->cfg.scan_mode != SHODH_SCAN_GRAPH) { set_error("not a SHODH_SCAN_GRAPH index"); return SHODH_ERR_STATE; }
    std::shared_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    if (idx->g_nodes != idx->n) { set_error("the graph does not cover the rows"); return SHODH_ERR_STATE; }
    const uint64_t n = idx->n;
    if (medoid) *medoid = idx->g_medoid;
    SHODH_HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> hd(n);
    if (n) SHODH_HIP_TRY(hipMemcpy(hd.data(), idx->g_deg, n * 4, hipMemcpyDeviceToHost));
    if (deg && n) memcpy(deg, hd.data(), n * 4);
    if (nbr && n) {
        if (stride < idx->g_stride) { set_error("stride %u < max_degree + 1", stride); return SHODH_ERR_INVALID; }
        std::vector<uint32_t> hn((size_t)n * idx->g_stride);
        SHODH_HIP_TRY(hipMemcpy(hn.data(), idx->g_nbr, hn.size() * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i)                             // entries past a node's degree read 0, whatever the slots held before
            for (uint32_t j = 0; j < stride; ++j) nbr[(size_t)i * stride + j] = j < hd[i] && j < idx->g_stride ? hn[(size_t)i * idx->g_stride + j] : 0u;
    }
    return SHODH_OK;
}

int shodh_index_build_with_graph(shodh_index *idx, const float *rows, uint64_t n, const uint32_t *deg, const uint32_t *nbr, uint32_t stride, uint32_t medoid) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (idx->cfg.scan_mode != SHODH_SCAN_GRAPH) { set_error("not a SHODH_SCAN_GRAPH index"); return SHODH_ERR_STATE; }
    SHODH_TRY(build_impl(idx, rows, n, hipMemcpyHostToDevice, false));
    return shodh_index_set_graph(idx, deg, nbr, stride, medoid);
}

int shodh_index_incremental_repair(shodh_index *idx, uint32_t first_node, uint32_t count, uint32_t *repaired_out) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (idx->cfg.scan_mode != SHODH_SCAN_GRAPH) { set_error("not a SHODH_SCAN_GRAPH index"); return SHODH_ERR_STATE; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);
    SHODH_TRY(set_device(idx));
    if (repaired_out) *repaired_out = 0;
    if (idx->g_nodes != idx->n) { set_error("the graph does not cover the rows: build it first"); return SHODH_ERR_STATE; }
    if (count == 0 || first_node >= idx->n) return SHODH_OK;
    uint32_t total = 0;
    for (uint64_t at = first_node; at < (uint64_t)first_node + count && at < idx->n; at += 1024) {       // bounded launches
        const uint32_t m = (uint32_t)std::min<uint64_t>(1024, std::min<uint64_t>((uint64_t)first_node + count, idx->n) - at);
        VgRepairArgs a{graph_of(idx), (uint32_t)idx->n, (uint32_t)at, m, idx->cfg.max_degree, idx->cfg.search_list_size, idx->g_medoid, idx->cfg.alpha,
                       idx->g_visited, idx->g_overflow, idx->g_overflow + 8};
        SHODH_TRY(vg_launch_repair(a, nullptr));
        uint32_t r = 0;
        const hipError_t e = hipMemcpy(&r, idx->g_overflow + 8, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("graph repair failed on device: %s", hipGetErrorString(e)); return SHODH_ERR_DEVICE; }
        total += r;
    }
    SHODH_TRY(check_graph_overflow(idx));
    if (repaired_out) *repaired_out = total;
    return SHODH_OK;
}

int shodh_index_vamana_build(shodh_index *idx, uint64_t seed, const uint32_t *init_deg, const uint32_t *init_nbr, uint32_t init_stride) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (idx->cfg.scan_mode != SHODH_SCAN_GRAPH) { set_error("not a SHODH_SCAN_GRAPH index"); return SHODH_ERR_STATE; }
    std::unique_lock<std::shared_mutex> lk(idx->mu);      // a construction is a write from start to end (searches wait, like behind the reference's &mut self)
    const uint64_t n = idx->n;
    if (n == 0) { idx->g_nodes = 0; return SHODH_OK; }
    const uint32_t R = idx->cfg.max_degree;
    std::vector<uint32_t> deg(n), nbr((size_t)n * (R + 1), 0u);
    if (init_deg && init_nbr) {
        for (uint64_t i = 0; i < n; ++i) {
            if (init_deg[i] > R || init_deg[i] > init_stride) { set_error("initial graph: node %llu has %u neighbours (max_degree %u)", (unsigned long long)i, init_deg[i], R); return SHODH_ERR_INVALID; }
            deg[i] = init_deg[i];
            for (uint32_t j = 0; j < deg[i]; ++j) {
                const uint32_t v = init_nbr[(size_t)i * init_stride + j];
                if (v >= n) { set_error("initial graph: node %llu lists neighbour %u (only %llu rows)", (unsigned long long)i, v, (unsigned long long)n); return SHODH_ERR_INVALID; }
                nbr[(size_t)i * (R + 1) + j] = v;
            }
        }
    } else {
        // initialize_graph (vamana.rs:287-312): `degree = min(R, n - 1)` distinct random neighbours per node, never the node itself
        uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ull;
        auto next = [&]() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
        const uint32_t d = (uint32_t)std::min<uint64_t>(R, n - 1);
        for (uint64_t i = 0; i < n; ++i) {
            deg[i] = d;
            uint32_t *row = nbr.data() + (size_t)i * (R + 1);
            for (uint32_t j = 0; j < d;) {
                uint64_t v = next() % (n - 1);
                if (v >= i) ++v;
                bool dup = false;
                for (uint32_t t = 0; t < j; ++t) dup |= row[t] == (uint32_t)v;
                if (!dup) row[j++] = (uint32_t)v;
            }
        }
    }
    // find_medoid (vamana.rs:407-441): mean vector, then the closest row with the first minimum winning == an exact search with k = 1
    uint32_t medoid = 0;
    {
        float *d_c = nullptr; uint32_t *d_i = nullptr; float *d_d = nullptr; uint32_t *d_n = nullptr;
        SHODH_TRY(set_device(idx));
        SHODH_HIP_TRY(dev_alloc((void **)&d_c, (size_t)idx->cfg.dim * 4 + 64));
        d_i = (uint32_t *)(d_c + idx->cfg.dim); d_d = (float *)(d_i + 1); d_n = (uint32_t *)(d_d + 1);
        int rc = vg_launch_centroid(idx->rows, (uint32_t)n, idx->cfg.dim, d_c, nullptr);
        if (rc == SHODH_OK) {
            const uint32_t gx = exact_grid_x(n, 1, 1, idx->cus);
            unsigned char *part = nullptr;
            if (dev_alloc((void **)&part, exact_partial_bytes(1, idx->cfg.dim, 1, gx) + 256) != hipSuccess) rc = SHODH_ERR_OOM;
            else {
                rc = launch_flat_exact(idx->rows, n, idx->cfg.dim, nullptr, d_c, 1, 1, idx->cfg.order, 0, (uint64_t *)part, gx, d_i, d_d, d_n, nullptr, nullptr, nullptr);
                if (rc == SHODH_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&medoid, d_i, 4, hipMemcpyDeviceToHost) != hipSuccess)) { set_error("medoid search failed"); rc = SHODH_ERR_DEVICE; }
                dev_free(part);
            }
        }
        dev_free(d_c);
        if (rc != SHODH_OK) return rc;
    }
    {
        SHODH_HIP_TRY(hipMemcpy(idx->g_deg, deg.data(), n * 4, hipMemcpyHostToDevice));
        SHODH_HIP_TRY(hipMemcpy(idx->g_nbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
        idx->g_medoid = medoid;
        // vamana.rs:246-283: pass after pass over all nodes, at most two, stopping when a pass changed nothing. Each launch takes
        // VG_BUILD_CHUNK nodes (a node costs a few milliseconds: one launch stays in the range of seconds); the pass state lives here.
        uint32_t *d_upd = nullptr;
        SHODH_HIP_TRY(dev_alloc((void **)&d_upd, 4));
        int rc = SHODH_OK;
        for (int pass = 1; pass <= 2 && rc == SHODH_OK; ++pass) {
            if (hipMemset(d_upd, 0, 4) != hipSuccess) { rc = SHODH_ERR_DEVICE; break; }
            for (uint64_t first = 0; first < n && rc == SHODH_OK; first += VG_BUILD_CHUNK) {
                VgBuildArgs a{graph_of(idx), (uint32_t)n, R, idx->cfg.search_list_size, medoid, idx->cfg.alpha, idx->g_visited, idx->g_overflow,
                              (uint32_t)first, (uint32_t)std::min<uint64_t>(VG_BUILD_CHUNK, n - first), d_upd};
                rc = vg_launch_build(a, nullptr);
                if (rc == SHODH_OK) { const hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { set_error("graph build failed on device: %s", hipGetErrorString(e)); rc = SHODH_ERR_DEVICE; } }
            }
            uint32_t upd = 0;
            if (rc == SHODH_OK && hipMemcpy(&upd, d_upd, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SHODH_ERR_DEVICE;
            if (upd == 0) break;
        }
        dev_free(d_upd);
        if (rc != SHODH_OK) return rc;
        idx->g_nodes = n;
        SHODH_TRY(check_graph_overflow(idx));
    }
    return SHODH_OK;
}

int shodh_topk_merge_device(const uint32_t *d_in_ids, const float *d_in_dist, uint32_t n_lists, uint32_t nq, uint32_t k,
                            uint32_t *d_ids, float *d_dist, uint32_t *d_counts, void *stream) {
    if (nq && (!d_in_ids || !d_in_dist || !d_ids || !d_dist || !d_counts)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    return shodh_topk_merge_strided_device(d_in_ids, d_in_dist, (uint64_t)nq * k, n_lists, nq, k, d_ids, d_dist, d_counts, stream);
}

int shodh_topk_merge_strided_device(const uint32_t *d_in_ids, const float *d_in_dist, uint64_t list_stride, uint32_t n_lists, uint32_t nq, uint32_t k,
                                    uint32_t *d_ids, float *d_dist, uint32_t *d_counts, void *stream) {
    if (nq && (!d_in_ids || !d_in_dist || !d_ids || !d_dist || !d_counts)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (k > 8192) { set_error("k too large"); return SHODH_ERR_UNSUPPORTED; }
    return launch_merge_lists(d_in_ids, d_in_dist, list_stride, n_lists, nq, k, d_ids, d_dist, d_counts, (hipStream_t)stream);
}

int shodh_index_stage_timings(const shodh_index *idx, float *us4) {
    if (!idx || !us4) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::lock_guard<std::mutex> g(idx->stat_mu);
    memcpy(us4, idx->last_us, sizeof(idx->last_us));
    return SHODH_OK;
}
int shodh_index_kernel_timing(shodh_index *idx, int reset, float *mean_us, float *min_us, uint32_t *count) {
    if (!idx) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::lock_guard<std::mutex> g(idx->ws_mu);
    double sum = 0; float mn = 1e30f; uint32_t n = 0;
    for (Workspace *w : idx->ws_free) {
        const uint32_t used = w->ring_pos < Workspace::RING ? w->ring_pos : Workspace::RING;
        for (uint32_t i = 0; i < used; ++i) {
            float ms = 0;
            if (w->ring[i][0] && w->ring[i][1] && hipEventElapsedTime(&ms, w->ring[i][0], w->ring[i][1]) == hipSuccess && ms > 0) { sum += ms; if (ms < mn) mn = ms; ++n; }
        }
        if (reset) w->ring_pos = 0;
    }
    (void)hipGetLastError();
    if (mean_us) *mean_us = n ? (float)(sum / n * 1000.0) : 0.0f;
    if (min_us) *min_us = n ? mn * 1000.0f : 0.0f;
    if (count) *count = n;
    return SHODH_OK;
}

int shodh_index_scan_stats(const shodh_index *idx, uint64_t *stats8) {
    if (!idx || !stats8) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::lock_guard<std::mutex> g(idx->stat_mu);
    memset(stats8, 0, 8 * sizeof(uint64_t));
    memcpy(stats8, idx->last_stats, sizeof(idx->last_stats));
    return SHODH_OK;
}

}  // extern "C"

// accessors for ivfpq.hip
namespace shodh {
IvfpqState *&index_ivfpq_slot(shodh_index *idx) { return idx->ivfpq; }
const shodh_index_cfg &index_cfg(const shodh_index *idx) { return idx->cfg; }
std::shared_mutex &index_mutex(shodh_index *idx) { return idx->mu; }
}  // namespace shodh

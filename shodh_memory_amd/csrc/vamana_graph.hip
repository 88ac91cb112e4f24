// vamana_graph.hip -- the Vamana GRAPH side of VamanaIndex on the device: the reference's default (non-exact) search path.
//   greedy_search   src/vector_db/vamana.rs:576-657      search (ANN)  :764-808
//   add_vector      :853-974 (incremental insert)        robust_prune  :665-746      find_medoid :407-441      build :200-284
//   incremental_repair :1033-1115
// Why it exists although the exact scan is faster up to ~10^7 rows per GPU: without SHODH_VECTOR_EXACT the reference answers from
// the graph, and an index that only ever grows through add_vector (what `remember` does) has a fully DETERMINISTIC graph (no RNG
// on that path, medoid 0). SHODH_SCAN_GRAPH reproduces that graph and that walk bit for bit, so a host that keeps the reference's
// default mode gets the reference's answers -- including its misses.
//
// SearchCandidate's order is total, (distance total_cmp, id) (:1664-1673), and ids are unique, so the two BinaryHeaps of
// greedy_search never show their internal order: here both are sorted arrays of 64-bit keys in LDS.
//
// One WORKGROUP of four waves runs one walk: wave 0 walks (the control flow is serial by nature), all four waves score each batch of
// neighbour rows (waves 1-3 wait in a request loop). State in LDS: the query, `w` (the k best so far, ascending), `cand` (the frontier,
// ascending, consumed from the head), the group sums of a distance batch, the visited set (an 8192-slot hash set; one bit per row in
// memory only when a walk outgrows it). A hop = pop the closest frontier node; its degree and adjacency row arrive in one round trip;
// its unvisited neighbours are found by the 64 lanes in parallel; their rows are read up to 64 at a time (every 16-byte piece of a batch
// in flight before the first is used); every distance is computed in the reference's accumulation order (distance_inline.rs:67-173);
// neighbours that cannot beat the worst of a full `w` are dropped in parallel, the rest are offered to `w` / `cand` ONE BY ONE in list
// order, exactly like the reference's loop: whether a neighbour is accepted depends on the worst entry of `w` at that moment.
// Inserts, repairs and the build run the same walk node after node: every step reads the graph the previous one wrote. The back-edge lists
// of one insert are independent of each other and are re-ranked four at a time, one per wave.
#include <cmath>

#include "common.h"

#pragma clang fp contract(off)

namespace shodh {

#ifdef SHODH_PROF
#define VGP_DECL long long vp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vq_ = wall_clock64();
#define VGP(i) { const long long t_ = wall_clock64(); vp_[i] += t_ - vq_; vq_ = t_; }
#else
#define VGP_DECL
#define VGP(i)
#endif
#ifdef SHODH_PROF
#define VGP_N vp_[7] += 1;
#else
#define VGP_N
#endif

constexpr int VG_W_CAP = 1024;     // beam: k (+ over-fetch for tombstones), search_list_size, max_degree
constexpr int VG_C_CAP = 2048;     // frontier entries kept (only those not worse than the worst of `w` can ever be expanded)
constexpr int VG_ROWS = 64;        // neighbour rows scored at a time (scalar-4 order, dim <= 512: a back-edge list of R + 1 = 33 rows is ONE batch; 32 in the
                                   // 8-chain order, 8 above 512 dimensions: the staging area has to fit the LDS)
constexpr int VG_HASH = 8192;     // entries of the visited set kept in LDS (a walk that outgrows 60 % of it moves to the one-bit-per-row map in memory)
constexpr int VG_U = 96;           // float4 loads per lane in flight in one round of vg_distances
constexpr int VG_MAXDEG = 128;     // neighbours per node the kernels handle (max_degree + 1 <= this)

struct VgGraph {
    const float *rows;      // [n][dim] f32 master rows
    uint32_t dim;
    uint32_t *deg;          // [cap_rows]
    uint32_t *nbr;          // [cap_rows][stride]
    uint32_t stride;        // >= max_degree + 1
    uint32_t order;         // SHODH_ORDER_SCALAR4 / AVX2
};

struct VgLds {
    float *q;               // [dim]
    uint64_t *w;            // [VG_W_CAP]
    uint64_t *cand;         // [VG_C_CAP]
    float *stage;           // [rows per batch][dim + 4]  raw rows (8-chain order only)
    float *tsum;            // [rows per batch][dim / 4 + 1] group sums (scalar-4) / [..][8] chain sums (AVX2)
    uint32_t rpb;           // rows per batch
    uint32_t *hset;         // [VG_HASH] visited set: open addressing, 0xFFFFFFFF = empty
    uint32_t *req;          // the walk's wave -> the three helper waves: rows to score (0xFFFFFFFF = done)
    uint32_t *newid;        // [VG_MAXDEG] unvisited neighbours of the current node, in list order
    float *newd;            // [VG_MAXDEG] their distances
    uint32_t *pr, *pr2;     // [VG_MAXDEG] pruned lists (build)
    float *dne, *dne2;      // [VG_MAXDEG] node -> kept neighbour distances (build)
};
__host__ __device__ inline uint32_t vg_rows_per_batch(uint32_t dim, uint32_t order) { return dim <= 512 ? (order == SHODH_ORDER_AVX2 ? 32u : (uint32_t)VG_ROWS) : 8u; }
__host__ __device__ inline size_t vg_lds_bytes(uint32_t dim, uint32_t order) {
    const size_t rpb = vg_rows_per_batch(dim, order), stage = order == SHODH_ORDER_AVX2 ? rpb * (dim + 4) * 4 : 0;      // scalar-4 sums straight from registers
    const size_t tcols = dim / 4 + 1 > 8 ? dim / 4 + 1 : 8;       // group sums per row (scalar-4) or the eight chain sums (8-chain order)
    return (size_t)dim * 4 + VG_W_CAP * 8 + VG_C_CAP * 8 + VG_HASH * 4 + stage + rpb * tcols * 4 + VG_MAXDEG * 8 + VG_MAXDEG * 16 + 64;
}
__device__ inline VgLds vg_carve(unsigned char *smem, uint32_t dim, uint32_t order) {
    VgLds l;
    l.w = reinterpret_cast<uint64_t *>(smem);
    l.cand = l.w + VG_W_CAP;
    l.hset = reinterpret_cast<uint32_t *>(l.cand + VG_C_CAP);
    l.q = reinterpret_cast<float *>(l.hset + VG_HASH);
    l.rpb = vg_rows_per_batch(dim, order);
    l.stage = l.q + dim;
    l.tsum = l.stage + (order == SHODH_ORDER_AVX2 ? (size_t)l.rpb * (dim + 4) : 0);
    l.newid = reinterpret_cast<uint32_t *>(l.tsum + (size_t)l.rpb * (dim / 4 + 1 > 8 ? dim / 4 + 1 : 8));
    l.newd = reinterpret_cast<float *>(l.newid + VG_MAXDEG);
    l.pr = reinterpret_cast<uint32_t *>(l.newd + VG_MAXDEG);
    l.pr2 = l.pr + VG_MAXDEG;
    l.dne = reinterpret_cast<float *>(l.pr2 + VG_MAXDEG);
    l.dne2 = l.dne + VG_MAXDEG;
    l.req = reinterpret_cast<uint32_t *>(l.dne2 + VG_MAXDEG);
    return l;
}

__device__ __forceinline__ float vg_key_dist(uint64_t k) { return order_key_inv((uint32_t)(k >> 32)); }

// distances -dot(q, row) of m rows (ids in l.newid) in the reference's order -> l.newd. NT cooperating threads (the walk's wave plus three
// helper waves: the batch is instruction-issue-bound for a single wave, ~1800 VALU / LDS instructions); dim % 8 == 0.
// Up to 32 rows per batch, and every 16-byte piece of a batch is requested before the first is used (VG_U per lane and round: a hop of
// the walk is a chain of dependent round trips, the rows are the longest of them and used to be fetched eight rows at a time).
constexpr int VG_NT = 256;          // threads of a walk's workgroup: wave 0 walks, all four waves score rows
template <int NT> __device__ __forceinline__ void vg_sync() {
    if (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}
// NT = 256: the whole workgroup on the walk's own buffers; NT = 64: ONE wave on private buffers (the back-edge lists of an insert are
// re-ranked four at a time, one per wave). VU = float4 loads per thread in flight per round.
struct VgDistBuf { const float *q; float *tsum; float *stage; const uint32_t *ids; float *out; uint32_t rpb; };
template <int NT, int VU>
__device__ void vg_dist_core(const VgGraph &g, const VgDistBuf &l, uint32_t m, int lane /* 0 .. NT - 1 */) {
    const uint32_t dim = g.dim, d4 = dim >> 2, sp = dim + 4, tp = d4 + 1, rpb = l.rpb;
    const bool avx = g.order == SHODH_ORDER_AVX2;
    for (uint32_t b0 = 0; b0 < m; b0 += rpb) {
        const uint32_t nb = m - b0 < rpb ? m - b0 : rpb, total = nb * d4;
        // element e = e0 + 64 u + lane is float4 group gq = e % d4 of batch row r = e / d4. d4 is a run-time value: the quotient comes from a float
        // multiply (exact for e < 2^13, d4 <= 256: (e + 0.5) / d4 stays 1 / (2 d4) away from every integer), an integer division per element
        // cost more than the loads themselves. Rounds of eight loads; a round past the end of the batch is skipped (a walk's first call scores
        // ONE row).
        const float inv_d4 = 1.0f / (float)d4;
        auto row_of = [&](uint32_t e) -> uint32_t { return (uint32_t)(((float)e + 0.5f) * inv_d4); };
        for (uint32_t e0 = 0; e0 < total; e0 += NT * VU) {
            float4 v[VU];
            const uint32_t e_first = e0 + lane;
#pragma unroll
            for (int u0 = 0; u0 < VU; u0 += 4) {
                if (e0 + u0 * NT < total) {                       // uniform
#pragma unroll
                    for (int u = u0; u < u0 + 4; ++u) {
                        const uint32_t e = e_first + u * NT, ec = e < total ? e : total - 1;
                        const uint32_t r = row_of(ec), gq = ec - r * d4;
                        v[u] = *reinterpret_cast<const float4 *>(g.rows + (size_t)l.ids[b0 + r] * dim + gq * 4);
                    }
                }
            }
#pragma unroll
            for (int u0 = 0; u0 < VU; u0 += 4) {
                if (e0 + u0 * NT < total) {
#pragma unroll
                    for (int u = u0; u < u0 + 4; ++u) {
                        const uint32_t e = e_first + u * NT;
                        if (e < total) {
                            const uint32_t r = row_of(e), gq = e - r * d4;
                            if (avx) {
                                *reinterpret_cast<float4 *>(l.stage + r * sp + gq * 4) = v[u];
                            } else {
                                // dot_product_scalar_inline (distance_inline.rs:157-173): t_g = ((a0 b0 + a1 b1) + a2 b2) + a3 b3 per group of
                                // four, straight from the registers the row arrived in
                                const float4 qa = *reinterpret_cast<const float4 *>(l.q + gq * 4);
                                float t = qa.x * v[u].x;
                                t = t + qa.y * v[u].y; t = t + qa.z * v[u].z; t = t + qa.w * v[u].w;
                                l.tsum[r * tp + gq] = t;
                            }
                        }
                    }
                }
            }
        }
        vg_sync<NT>();
        if (avx) {
            // dot_product_avx2_inline (distance_inline.rs:67-111): 8 FMA chains over i = c, c + 8, ...; lanes 0..7 summed in order
            for (uint32_t rc = lane; rc < nb * 8; rc += NT) {
                const uint32_t r = rc >> 3, c = rc & 7;
                const float *row = l.stage + r * sp;
                float acc = 0.0f;
                uint32_t i = 0;
                for (; i + 64 <= dim; i += 64) {                // eight steps of the chain per round of LDS reads
                    float qa[8], ra[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { qa[u] = l.q[i + 8 * u + c]; ra[u] = row[i + 8 * u + c]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(qa[u], ra[u], acc);
                }
                for (; i < dim; i += 8) acc = __builtin_fmaf(l.q[i + c], row[i + c], acc);
                l.tsum[r * 8 + c] = acc;
            }
            vg_sync<NT>();
            if ((uint32_t)lane < nb) {
                const float *p8 = l.tsum + lane * 8;
                float s = p8[0] + p8[1];
                s = s + p8[2]; s = s + p8[3]; s = s + p8[4]; s = s + p8[5]; s = s + p8[6]; s = s + p8[7];
                l.out[b0 + lane] = -s;
            }
        } else {
            if ((uint32_t)lane < nb) {            // sum += t_g in order
                const float *tr = l.tsum + lane * tp;
                float s = 0.0f;
                uint32_t gq = 0;
                for (; gq + 8 <= d4; gq += 8) {                 // eight LDS reads in flight, then the eight additions in order
                    float t8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t8[u] = tr[gq + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = s + t8[u];
                }
                for (; gq < d4; ++gq) s = s + tr[gq];
                l.out[b0 + lane] = -s;
            }
        }
        vg_sync<NT>();
    }
}

__device__ void vg_distances(const VgGraph &g, const VgLds &l, uint32_t m, int lane /* 0 .. VG_NT - 1 */) {
    const VgDistBuf b{l.q, l.tsum, l.stage, l.newid, l.newd, l.rpb};
    vg_dist_core<VG_NT, VG_U * 64 / VG_NT>(g, b, m, lane);
}

// The walk itself runs on wave 0 of the workgroup; the other three waves wait for row batches to score.
//   wave 0:   vg_dist_master(m)  =  request m rows -> barrier -> all four waves score -> barrier
//   waves 1-3: vg_helper_loop()  =  barrier -> read the request (0xFFFFFFFF: the walk is over) -> score -> barrier
__device__ void vg_dist_master(const VgGraph &g, const VgLds &l, uint32_t m, int lane) {
    if (lane == 0) *l.req = m;
    __syncthreads();
    vg_distances(g, l, m, lane);
    __syncthreads();
}
// The back edges of one insert (vamana.rs:925-957): every neighbour nb of the new node `id` gets `id` appended; a list that outgrows R keeps
// its R closest entries by (distance from nb, id). The lists belong to different nodes, so they are independent: wave w takes neighbours
// w, w + 4, ... with private buffers carved from the regions a finished walk no longer needs (frontier + visited set: three slices of
// 16 KiB; the group-sum area for the fourth wave). Scalar-4 order only (the 8-chain order would need 50 KiB of raw rows per wave).
constexpr uint32_t VG_REQ_EXIT = 0xFFFFFFFFu, VG_REQ_BACKEDGE = 0xFFFFFFFEu;
__host__ __device__ inline bool vg_parallel_backedges(uint32_t dim, uint32_t order, uint32_t R) {
    // per wave: q[dim] + group sums [(R + 1)][dim / 4 + 1] + 64 ids + 64 distances. Waves 0-2 get 16 KiB slices of the frontier + visited set; wave 3
    // gets the group-sum area of the walk, rows-per-batch x max(dim / 4 + 1, 8) floats, plus the walk's newid / newd behind it (2 x VG_MAXDEG
    // words, free once the walk is over) -- what follows is `pr`, the neighbour list all four waves are reading. (Round 2 checked only the
    // 16 KiB slices: max_degree 62 / 63 with dim 136 .. 232 put wave 3's distances on top of `pr`)
    const size_t need = (size_t)dim + (size_t)(R + 1) * (dim / 4 + 1) + 128;
    const size_t tcols = dim / 4 + 1 > 8 ? dim / 4 + 1 : 8;
    const size_t wave3 = (size_t)vg_rows_per_batch(dim, order) * tcols + 2 * VG_MAXDEG;
    return order != SHODH_ORDER_AVX2 && dim <= 512 && R + 1 <= 64 && need * 4 <= 16384 && need <= wave3;
}
__device__ void vg_backedge_task(const VgGraph &g, const VgLds &l, int wave, int lane) {
    const uint32_t id = l.req[2], take = l.req[3], R = l.req[4];
    const uint32_t dim = g.dim, tp = dim / 4 + 1;
    unsigned char *base = wave < 3 ? reinterpret_cast<unsigned char *>(l.cand) + (size_t)wave * 16384 : reinterpret_cast<unsigned char *>(l.tsum);
    float *qw = reinterpret_cast<float *>(base);
    float *tsw = qw + dim;
    uint32_t *idw = reinterpret_cast<uint32_t *>(tsw + (size_t)(R + 1) * tp);
    float *dw = reinterpret_cast<float *>(idw + 64);
    for (uint32_t i = wave; i < take; i += 4) {
        const uint32_t nb = l.pr[i];
        const uint32_t dg = g.deg[nb];
        if (lane == 0) { g.nbr[(size_t)nb * g.stride + dg] = id; g.deg[nb] = dg + 1; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        if (dg + 1 > R) {
            const uint32_t cnt = dg + 1;
            for (uint32_t t = lane; t < dim; t += 64) qw[t] = g.rows[(size_t)nb * dim + t];
            if ((uint32_t)lane < cnt) idw[lane] = g.nbr[(size_t)nb * g.stride + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const VgDistBuf b{qw, tsw, nullptr, idw, dw, 64u};
            vg_dist_core<64, 32>(g, b, cnt, lane);
            if ((uint32_t)lane < cnt) {
                const uint64_t key = make_key(dw[lane], idw[lane]);
                uint32_t r = 0;
                for (uint32_t u = 0; u < cnt; ++u) r += make_key(dw[u], idw[u]) < key;
                if (r < R) g.nbr[(size_t)nb * g.stride + r] = idw[lane];
            }
            if (lane == 0) g.deg[nb] = R;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        }
    }
}
__device__ void vg_helper_loop(const VgGraph &g, const VgLds &l, int tid) {
    for (;;) {
        __syncthreads();
        const uint32_t m = *l.req;
        if (m == VG_REQ_EXIT) break;
        if (m == VG_REQ_BACKEDGE) vg_backedge_task(g, l, tid >> 6, tid & 63);
        else vg_distances(g, l, m, tid);
        __syncthreads();
    }
}
__device__ void vg_walk_done(const VgLds &l, int lane) {
    if (lane == 0) *l.req = 0xFFFFFFFFu;
    __syncthreads();
}

// inserts key into the ascending array a[lo, n) (n < cap); returns the new n. One wave.
__device__ uint32_t vg_sorted_insert(uint64_t *a, uint32_t lo, uint32_t n, uint64_t key, int lane) {
    if (n - lo < 64u) {
        // the usual case (a beam of k <= 63, a frontier of a few dozen entries): one pass in registers -- lane i holds entry lo + i and its
        // left neighbour, the ballot gives the position, every lane writes its new value
        const uint32_t cnt = n - lo;
        const uint64_t v = (uint32_t)lane < cnt ? a[lo + lane] : ~0ull;
        const uint64_t left = (lane > 0 && (uint32_t)lane <= cnt) ? a[lo + lane - 1] : 0ull;
        const uint32_t pos = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64((uint32_t)lane < cnt && v < key));
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane <= cnt) a[lo + lane] = (uint32_t)lane < pos ? v : ((uint32_t)lane == pos ? key : left);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        return n + 1;
    }
    // position = lo + #keys in [lo, n) smaller than key
    uint32_t pos = lo;
    for (uint32_t i0 = lo; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool less = i < n && a[i] < key;
        pos += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(less));
    }
    // shift [pos, n) up by one, highest chunk first (a chunk is read completely before it is written)
    for (uint32_t hi = n; hi > pos;) {
        const uint32_t cl = hi - pos < 64 ? hi - pos : 64, base = hi - cl;
        const uint32_t i = base + lane;
        uint64_t v = 0;
        if ((uint32_t)lane < cl) v = a[i];
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < cl) a[i + 1] = v;
        __builtin_amdgcn_wave_barrier();
        hi = base;
    }
    if (lane == 0) a[pos] = key;
    __builtin_amdgcn_wave_barrier();
    return n + 1;
}

// greedy_search (vamana.rs:576-657) for the query in l.q over nodes [0, n): the k best end up in l.w[0, return value), ascending.
// visited: n bits, private to this wave. *overflow is set if the frontier array was ever full (cannot happen unless thousands of
// candidates tie with the worst of `w`).
__device__ uint32_t vg_greedy(const VgGraph &g, const VgLds &l, uint32_t n, uint32_t k, uint32_t entry, uint32_t *visited, uint32_t *overflow, int lane) {
    // visited set: a hash set in LDS (no trip to memory per hop, nothing to clear but 32 KiB of LDS); `visited` (one bit per row in
    // memory, test-and-set with atomicOr) takes over only if a walk outgrows it
    for (uint32_t i = lane * 4; i < (uint32_t)VG_HASH; i += 256) *reinterpret_cast<uint4 *>(l.hset + i) = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t hcount = 1;                 // wave-uniform
    bool in_memory = false;              // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { l.newid[0] = entry; l.hset[(entry * 2654435761u) >> 19] = entry; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // true if id was not in the set before (and now is)
    auto hash_insert = [&](uint32_t id) -> bool {
        uint32_t h = (id * 2654435761u) >> 19;       // 13 bits
        for (;;) {
            const uint32_t old = atomicCAS(l.hset + h, 0xFFFFFFFFu, id);
            if (old == 0xFFFFFFFFu) return true;
            if (old == id) return false;
            h = (h + 1) & (uint32_t)(VG_HASH - 1);
        }
    };
    VGP_DECL
    vg_dist_master(g, l, 1, lane);
    VGP(0)
    uint32_t wn = 1, ch = 0, cn = 1;             // |w|, frontier head, frontier end  (wave-uniform)
    if (lane == 0) { const uint64_t key = make_key(l.newd[0], entry); l.w[0] = key; l.cand[0] = key; }
    __builtin_amdgcn_wave_barrier();
    while (ch < cn) {
        const uint64_t cur = l.cand[ch++];
        if (vg_key_dist(cur) > vg_key_dist(l.w[wn - 1])) break;          // current.distance > worst of w
        const uint32_t cid = (uint32_t)cur;
        if (cid >= n) continue;
        // the degree and the whole adjacency row in one round trip (the row is allocated to its full stride)
        uint32_t nbv[VG_MAXDEG / 64];
#pragma unroll
        for (int t = 0; t < VG_MAXDEG / 64; ++t) { const uint32_t j = t * 64 + lane; nbv[t] = j < g.stride ? g.nbr[(size_t)cid * g.stride + j] : 0xFFFFFFFFu; }
        const uint32_t dg = g.deg[cid];
#ifdef SHODH_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        VGP(1)
        // unvisited neighbours, in list order
        uint32_t m = 0;
#pragma unroll
        for (int t = 0; t < VG_MAXDEG / 64; ++t) {
            const uint32_t j = t * 64 + lane;
            const uint32_t nb = nbv[t];
            bool fresh = false;
            if (j < dg && nb < n) fresh = in_memory ? (atomicOr(visited + (nb >> 5), 1u << (nb & 31)) & (1u << (nb & 31))) == 0 : hash_insert(nb);
            const uint64_t bal = __builtin_amdgcn_ballot_w64(fresh);
            if (fresh) l.newid[m + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = nb;
            m += (uint32_t)__builtin_popcountll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        VGP(2)
        hcount += m;
        if (!in_memory && hcount > (uint32_t)(VG_HASH * 6 / 10)) {
            // the set is filling up: everything seen so far goes to the bit map in memory, which serves the rest of the walk
            for (uint32_t i = lane; i < (n + 31) / 32; i += 64) visited[i] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            for (uint32_t i = lane; i < (uint32_t)VG_HASH; i += 64) { const uint32_t id = l.hset[i]; if (id != 0xFFFFFFFFu) atomicOr(visited + (id >> 5), 1u << (id & 31)); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            in_memory = true;
        }
        if (m == 0) continue;
        vg_dist_master(g, l, m, lane);
        VGP(3)
        VGP_N
        // The neighbours are offered to `w` / the frontier one by one in list order, like the reference's loop. Once `w` is full its worst
        // entry only ever improves, so a neighbour that does not beat the worst entry NOW never will: those are dropped up front, in
        // parallel (late in a walk that is nearly all of them), and only the rest takes the sequential path.
        if (wn == k) {
            const float worst0 = vg_key_dist(l.w[wn - 1]);
            uint32_t kept = 0;
            for (uint32_t i0 = 0; i0 < m; i0 += 64) {
                const uint32_t i = i0 + lane;
                const uint32_t idv = i < m ? l.newid[i] : 0u;
                const float dv = i < m ? l.newd[i] : 0.0f;
                const bool keep = i < m && dv < worst0;
                const uint64_t bal = __builtin_amdgcn_ballot_w64(keep);
                __builtin_amdgcn_wave_barrier();
                if (keep) {
                    const uint32_t pos = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    l.newid[pos] = idv; l.newd[pos] = dv;          // pos <= i: entries of later chunks are not overwritten before they are read
                }
                kept += (uint32_t)__builtin_popcountll(bal);
                __builtin_amdgcn_wave_barrier();
            }
            m = kept;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        for (uint32_t i = 0; i < m; ++i) {                                // one by one, in list order, like the reference
            const float d = l.newd[i];
            const bool add = wn < k || d < vg_key_dist(l.w[wn - 1]);
            if (!add) continue;
            const uint64_t key = make_key(d, l.newid[i]);
            if (cn == (uint32_t)VG_C_CAP) {
                if (ch > 0) {                                             // reclaim the consumed head
                    for (uint32_t t0 = 0; t0 < cn - ch; t0 += 64) {
                        uint64_t v = 0;
                        if (t0 + lane < cn - ch) v = l.cand[ch + t0 + lane];
                        __builtin_amdgcn_wave_barrier();
                        if (t0 + lane < cn - ch) l.cand[t0 + lane] = v;
                        __builtin_amdgcn_wave_barrier();
                    }
                    cn -= ch; ch = 0;
                }
                if (cn == (uint32_t)VG_C_CAP) { cn = VG_C_CAP - 1; if (lane == 0) *overflow = 1; }     // drop the worst frontier entry
            }
            cn = vg_sorted_insert(l.cand, ch, cn, key, lane);
            wn = vg_sorted_insert(l.w, 0, wn, key, lane);
            if (wn > k) wn = k;                                           // w.pop(): the largest key goes
        }
        VGP(4)
    }
#ifdef SHODH_PROF
    if (blockIdx.x == 0 && lane == 0) printf("walk: entry %lld | adjacency %lld | visited %lld | distances %lld | offers %lld (10 ns ticks), hops %lld\n", vp_[0], vp_[1], vp_[2], vp_[3], vp_[4], vp_[7]);
#endif
    return wn;
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------------
struct VgSearchArgs {
    VgGraph g;
    uint32_t n, medoid;
    const float *q;           // [nq][dim]
    uint32_t nq, k, search_k; // search_k = k + min(deleted, 2k)
    const uint32_t *deleted;  // bitmask or null
    uint32_t id_base;
    uint32_t *visited;        // [nq][vis_words]
    uint32_t vis_words;
    uint32_t *ids; float *dist; uint32_t *counts;   // [nq][k]
    uint32_t *overflow;
};
__global__ __launch_bounds__(256) void vg_search_kernel(VgSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgLds l = vg_carve(smem, a.g.dim, a.g.order);
    const int lane = threadIdx.x;
    if (threadIdx.x >= 64) { vg_helper_loop(a.g, l, (int)threadIdx.x); return; }      // waves 1-3 only score row batches for wave 0
    const uint32_t qi = blockIdx.x;
    for (uint32_t i = lane; i < a.g.dim; i += 64) l.q[i] = a.q[(size_t)qi * a.g.dim + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint32_t *ovf = l.req + 1;                 // this walk's own overflow flag (LDS): reported in the top bit of counts[qi]
    if (lane == 0) *ovf = 0;
    const uint32_t wn = vg_greedy(a.g, l, a.n, a.search_k, a.medoid, a.visited + (size_t)qi * a.vis_words, ovf, lane);
    vg_walk_done(l, lane);
    // filter tombstones, take k (vamana.rs:797-804)
    uint32_t o = 0;
    for (uint32_t i0 = 0; i0 < wn && o < a.k; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool live = false;
        uint64_t key = 0;
        if (i < wn) { key = l.w[i]; const uint32_t id = (uint32_t)key; live = !(a.deleted && ((a.deleted[id >> 5] >> (id & 31)) & 1u)); }
        const uint64_t bal = __builtin_amdgcn_ballot_w64(live);
        const uint32_t slot = o + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (live && slot < a.k) { a.ids[(size_t)qi * a.k + slot] = a.id_base + (uint32_t)key; a.dist[(size_t)qi * a.k + slot] = vg_key_dist(key); }
        o += (uint32_t)__builtin_popcountll(bal);
    }
    if (o > a.k) o = a.k;
    for (uint32_t i = o + lane; i < a.k; i += 64) { a.ids[(size_t)qi * a.k + i] = 0xFFFFFFFFu; a.dist[(size_t)qi * a.k + i] = __builtin_inff(); }
    if (lane == 0) a.counts[qi] = o | (*ovf ? 0x80000000u : 0u);
}

// add_vector for rows [first, first + count), one after the other (vamana.rs:853-974). One wave.
struct VgInsertArgs {
    VgGraph g;
    uint32_t first, count, R, medoid;
    uint32_t *visited;        // [vis_words]
    uint32_t *overflow;
};
__global__ __launch_bounds__(256) void vg_insert_kernel(VgInsertArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgLds l = vg_carve(smem, a.g.dim, a.g.order);
    const int lane = threadIdx.x;
    if (threadIdx.x >= 64) { vg_helper_loop(a.g, l, (int)threadIdx.x); return; }      // waves 1-3 only score row batches for wave 0
    const uint32_t dim = a.g.dim;
    for (uint32_t id = a.first; id < a.first + a.count; ++id) {
        if (id == 0) { if (lane == 0) a.g.deg[0] = 0; continue; }          // the first vector: a node without neighbours
        for (uint32_t i = lane; i < dim; i += 64) l.q[i] = a.g.rows[(size_t)id * dim + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t wn = vg_greedy(a.g, l, id, a.R, a.medoid, a.visited, a.overflow, lane);   // the graph holds nodes [0, id)
        const uint32_t take = wn < a.R ? wn : a.R;
        for (uint32_t i = lane; i < take; i += 64) a.g.nbr[(size_t)id * a.g.stride + i] = (uint32_t)l.w[i];
        if (lane == 0) a.g.deg[id] = take;
        // back edges; a list that outgrows R keeps its R closest by (distance, id) (:925-957)
        if (vg_parallel_backedges(dim, a.g.order, a.R)) {
            for (uint32_t i = lane; i < take; i += 64) l.pr[i] = (uint32_t)l.w[i];
            if (lane == 0) { l.req[2] = id; l.req[3] = take; l.req[4] = a.R; *l.req = VG_REQ_BACKEDGE; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");           // the new node's own list is out before anybody lists it
            __syncthreads();
            vg_backedge_task(a.g, l, 0, lane);
            __syncthreads();
        } else
        for (uint32_t i = 0; i < take; ++i) {
            const uint32_t nb = (uint32_t)l.w[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            const uint32_t dg = a.g.deg[nb];
            if (lane == 0) { a.g.nbr[(size_t)nb * a.g.stride + dg] = id; a.g.deg[nb] = dg + 1; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            if (dg + 1 > a.R) {
                // distances from nb to each of its dg + 1 neighbours: the "query" is nb's row (w stays where it is)
                const uint32_t cnt = dg + 1;
                for (uint32_t t = lane; t < dim; t += 64) l.q[t] = a.g.rows[(size_t)nb * dim + t];
                for (uint32_t t = lane; t < cnt; t += 64) l.newid[t] = a.g.nbr[(size_t)nb * a.g.stride + t];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                vg_dist_master(a.g, l, cnt, lane);
                // rank by (distance total_cmp, id); ranks are unique (ids are): entry t goes to position rank(t) if rank < R
                for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
                    const uint32_t t = t0 + lane;
                    if (t < cnt) {
                        const uint64_t key = make_key(l.newd[t], l.newid[t]);
                        uint32_t r = 0;
                        for (uint32_t u = 0; u < cnt; ++u) r += make_key(l.newd[u], l.newid[u]) < key;
                        if (r < a.R) a.g.nbr[(size_t)nb * a.g.stride + r] = l.newid[t];
                    }
                }
                if (lane == 0) a.g.deg[nb] = a.R;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    }
    vg_walk_done(l, lane);
}

// robust_prune of `node` over the candidates in l.w[0, nc) (keys: distance | id; sorted ascending == the reference's sort) -> pruned ids
// in l.newid[?]... the pruned list is written to out[0, return). One wave. (vamana.rs:665-746)
__device__ uint32_t vg_robust_prune(const VgGraph &g, const VgLds &l, uint32_t node, uint32_t nc, uint32_t R, float alpha, uint32_t *out /* LDS, R */,
                                    float *out_dne /* LDS, R */, int lane) {
    const uint32_t dim = g.dim;
    uint32_t np = 0;
    // node's row as the query for dist_nc
    for (uint32_t i = 0; i < nc && np < R; ++i) {
        const uint32_t cid = (uint32_t)l.w[i];
        if (cid == node) continue;
        // dist_nc = distance(node, c); dist_ce = distance(c, e_j) for the kept e_j: one batch of 1 + np rows against the query c... the
        // reference evaluates dist_ce lazily and stops at the first hit; the values are the same, only evaluated eagerly here.
        for (uint32_t t = lane; t < dim; t += 64) l.q[t] = g.rows[(size_t)cid * dim + t];
        if (lane == 0) l.newid[0] = node;
        for (uint32_t t = lane; t < np; t += 64) l.newid[1 + t] = out[t];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        vg_dist_master(g, l, 1 + np, lane);                                   // newd[0] = dist(c, node) == dist(node, c) term by term
        const float dist_nc = l.newd[0];
        bool hit = false;
        for (uint32_t j0 = 0; j0 < np; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool h = false;
            if (j < np) { const float dist_ce = l.newd[1 + j]; h = alpha * (dist_ce + 1.0f) <= (dist_nc + 1.0f) && dist_ce <= out_dne[j]; }
            hit = hit || __builtin_amdgcn_ballot_w64(h) != 0;
        }
        if (!hit) { if (lane == 0) { out[np] = cid; out_dne[np] = dist_nc; } ++np; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    return np;
}

// build (vamana.rs:200-284) over the graph already in deg / nbr (the initial random graph). One wave, node after node, at most two passes.
// A launch covers the nodes [first, first + count) of ONE pass and adds the lists it changed to *updates: the host walks the node ranges
// (VG_BUILD_CHUNK per launch) and decides about the second pass, so that no launch runs for minutes under the index's lock and a build can
// be bounded / reported on (round 2 ran both passes over all nodes in one launch).
constexpr uint32_t VG_BUILD_CHUNK = 512;
struct VgBuildArgs {
    VgGraph g;
    uint32_t n, R, L, medoid;
    float alpha;
    uint32_t *visited;
    uint32_t *overflow;
    uint32_t first, count;
    uint32_t *updates;        // device counter (atomicAdd)
};
__global__ __launch_bounds__(256) void vg_build_kernel(VgBuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgLds l = vg_carve(smem, a.g.dim, a.g.order);
    const int lane = threadIdx.x;
    if (threadIdx.x >= 64) { vg_helper_loop(a.g, l, (int)threadIdx.x); return; }      // waves 1-3 only score row batches for wave 0
    const uint32_t dim = a.g.dim;
    uint32_t *pr = l.pr, *pr2 = l.pr2;
    float *dne = l.dne, *dne2 = l.dne2;
    {
        uint32_t updates = 0;
        const uint32_t end = a.first + a.count < a.n ? a.first + a.count : a.n;
        for (uint32_t node = a.first; node < end; ++node) {
            for (uint32_t i = lane; i < dim; i += 64) l.q[i] = a.g.rows[(size_t)node * dim + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t m = vg_greedy(a.g, l, a.n, a.L, a.medoid, a.visited, a.overflow, lane);
            const uint32_t np = vg_robust_prune(a.g, l, node, m, a.R, a.alpha, pr, dne, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            const uint32_t dg = a.g.deg[node];
            bool same = np == dg;
            if (same) {
                bool diff = false;
                for (uint32_t j = lane; j < np; j += 64) diff = diff || a.g.nbr[(size_t)node * a.g.stride + j] != pr[j];
                same = __builtin_amdgcn_ballot_w64(diff) == 0;
            }
            if (same) continue;
            ++updates;
            for (uint32_t j = lane; j < np; j += 64) a.g.nbr[(size_t)node * a.g.stride + j] = pr[j];
            if (lane == 0) a.g.deg[node] = np;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            for (uint32_t j = 0; j < np; ++j) {
                const uint32_t nb = pr[j];
                if (nb >= a.n) continue;
                const uint32_t dn = a.g.deg[nb];
                bool has = false;
                for (uint32_t t = lane; t < dn; t += 64) has = has || a.g.nbr[(size_t)nb * a.g.stride + t] == node;
                if (__builtin_amdgcn_ballot_w64(has) != 0) continue;
                if (lane == 0) { a.g.nbr[(size_t)nb * a.g.stride + dn] = node; a.g.deg[nb] = dn + 1; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                if (dn + 1 > a.R) {
                    // robust_prune(nb, its neighbours with distance 0.0): sorted by (0.0, id) = by id
                    const uint32_t cnt = dn + 1;
                    for (uint32_t t = lane; t < cnt; t += 64) l.newid[t] = a.g.nbr[(size_t)nb * a.g.stride + t];
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
                        const uint32_t t = t0 + lane;
                        if (t < cnt) {
                            const uint32_t idv = l.newid[t];
                            uint32_t r = 0;
                            for (uint32_t u = 0; u < cnt; ++u) r += l.newid[u] < idv;
                            l.w[r] = make_key(0.0f, idv);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t q2 = vg_robust_prune(a.g, l, nb, cnt, a.R, a.alpha, pr2, dne2, lane);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                    for (uint32_t t = lane; t < q2; t += 64) a.g.nbr[(size_t)nb * a.g.stride + t] = pr2[t];
                    if (lane == 0) a.g.deg[nb] = q2;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                }
            }
        }
        if (lane == 0 && updates) atomicAdd(a.updates, updates);
    }
    vg_walk_done(l, lane);
}

// incremental_repair (vamana.rs:1033-1115) for nodes [first, first + count): walk with beam L from the medoid, robust_prune, and where the
// list changed: remove the stale back edges (retain), push the new ones and truncate to R. One workgroup, node after node.
struct VgRepairArgs {
    VgGraph g;
    uint32_t n, first, count, R, L, medoid;
    float alpha;
    uint32_t *visited;
    uint32_t *overflow;
    uint32_t *repaired;       // out: nodes whose list changed
};
__global__ __launch_bounds__(256) void vg_repair_kernel(VgRepairArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const VgLds l = vg_carve(smem, a.g.dim, a.g.order);
    const int lane = threadIdx.x;
    if (threadIdx.x >= 64) { vg_helper_loop(a.g, l, (int)threadIdx.x); return; }
    const uint32_t dim = a.g.dim;
    uint32_t *pr = l.pr, *old = l.pr2;
    float *dne = l.dne;
    uint32_t repaired = 0;
    for (uint32_t node = a.first; node < a.first + a.count && node < a.n; ++node) {
        for (uint32_t i = lane; i < dim; i += 64) l.q[i] = a.g.rows[(size_t)node * dim + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t m = vg_greedy(a.g, l, a.n, a.L, a.medoid, a.visited, a.overflow, lane);
        const uint32_t np = vg_robust_prune(a.g, l, node, m, a.R, a.alpha, pr, dne, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        const uint32_t od = a.g.deg[node];
        bool same = np == od;
        if (same) {
            bool diff = false;
            for (uint32_t j = lane; j < np; j += 64) diff = diff || a.g.nbr[(size_t)node * a.g.stride + j] != pr[j];
            same = __builtin_amdgcn_ballot_w64(diff) == 0;
        }
        if (same) continue;
        ++repaired;
        for (uint32_t j = lane; j < od; j += 64) old[j] = a.g.nbr[(size_t)node * a.g.stride + j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = lane; j < np; j += 64) a.g.nbr[(size_t)node * a.g.stride + j] = pr[j];
        if (lane == 0) a.g.deg[node] = np;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        for (uint32_t j = 0; j < od; ++j) {                      // back edges of neighbours that are gone: retain(|x| x != node)
            const uint32_t o = old[j];
            bool kept = false;
            for (uint32_t t = lane; t < np; t += 64) kept = kept || pr[t] == o;
            if (__builtin_amdgcn_ballot_w64(kept) != 0 || o >= a.n) continue;
            const uint32_t dg = a.g.deg[o];
            // lists hold at most VG_MAXDEG entries: two rounds of 64 lanes, the entries after the removed one move up by one
            uint32_t vals[VG_MAXDEG / 64];
            uint32_t pos = 0xFFFFFFFFu;
#pragma unroll
            for (int t = 0; t < VG_MAXDEG / 64; ++t) {
                const uint32_t i = t * 64 + lane;
                vals[t] = i < dg ? a.g.nbr[(size_t)o * a.g.stride + i] : 0xFFFFFFFFu;
                const uint64_t hit = __builtin_amdgcn_ballot_w64(i < dg && vals[t] == node);
                if (hit && pos == 0xFFFFFFFFu) pos = t * 64 + (uint32_t)__builtin_ctzll(hit);
            }
            if (pos == 0xFFFFFFFFu) continue;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < VG_MAXDEG / 64; ++t) {
                const uint32_t i = t * 64 + lane;
                if (i < dg && i > pos) a.g.nbr[(size_t)o * a.g.stride + i - 1] = vals[t];
            }
            if (lane == 0) a.g.deg[o] = dg - 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        }
        for (uint32_t j = 0; j < np; ++j) {                      // back edges to the new neighbours: push, truncate to R
            const uint32_t p = pr[j];
            if (p >= a.n) continue;
            const uint32_t dg = a.g.deg[p];
            bool has = false;
            for (uint32_t t = lane; t < dg; t += 64) has = has || a.g.nbr[(size_t)p * a.g.stride + t] == node;
            if (__builtin_amdgcn_ballot_w64(has) != 0) continue;
            if (lane == 0) {
                if (dg < a.R) { a.g.nbr[(size_t)p * a.g.stride + dg] = node; a.g.deg[p] = dg + 1; }
                else a.g.deg[p] = a.R;                            // the pushed entry (and anything beyond R) falls to the truncation
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        }
    }
    if (lane == 0) *a.repaired = repaired;
    vg_walk_done(l, lane);
}

// find_medoid (vamana.rs:407-441): the mean vector, coordinate sums in row order; then the closest row, first minimum wins.
__global__ __launch_bounds__(256) void vg_centroid_kernel(const float *rows, uint32_t n, uint32_t dim, float *centroid) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= dim) return;
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) s = s + rows[(size_t)i * dim + j];
    centroid[j] = s / (float)n;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
int vg_launch_search(const VgSearchArgs &a, hipStream_t st) {
    const size_t lds = vg_lds_bytes(a.g.dim, a.g.order);
    SHODH_TRY(ensure_dynamic_lds((const void *)vg_search_kernel, lds));
    hipLaunchKernelGGL(vg_search_kernel, dim3(a.nq), dim3(VG_NT), lds, st, a);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int vg_launch_insert(const VgInsertArgs &a, hipStream_t st) {
    const size_t lds = vg_lds_bytes(a.g.dim, a.g.order);
    SHODH_TRY(ensure_dynamic_lds((const void *)vg_insert_kernel, lds));
    hipLaunchKernelGGL(vg_insert_kernel, dim3(1), dim3(VG_NT), lds, st, a);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int vg_launch_build(const VgBuildArgs &a, hipStream_t st) {
    const size_t lds = vg_lds_bytes(a.g.dim, a.g.order);
    SHODH_TRY(ensure_dynamic_lds((const void *)vg_build_kernel, lds));
    hipLaunchKernelGGL(vg_build_kernel, dim3(1), dim3(VG_NT), lds, st, a);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int vg_launch_repair(const VgRepairArgs &a, hipStream_t st) {
    const size_t lds = vg_lds_bytes(a.g.dim, a.g.order);
    SHODH_TRY(ensure_dynamic_lds((const void *)vg_repair_kernel, lds));
    hipLaunchKernelGGL(vg_repair_kernel, dim3(1), dim3(VG_NT), lds, st, a);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int vg_launch_centroid(const float *rows, uint32_t n, uint32_t dim, float *centroid, hipStream_t st) {
    hipLaunchKernelGGL(vg_centroid_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, rows, n, dim, centroid);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

}  // namespace shodh

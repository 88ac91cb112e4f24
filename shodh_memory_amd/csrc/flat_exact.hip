// flat_exact.hip -- exact brute-force scan: every (query,row) score is computed in the
// reference's accumulation order, so scores are bit-identical to VamanaIndex::brute_force_search
// (src/vector_db/vamana.rs:1167-1188 with distance = -dot_product_inline, distance_inline.rs:479-481)
// and no re-rank is needed.
//
// Roofline: HBM. One pass over the f32 rows (dim*4 bytes per row) per group of QB queries.
// Layout: rows [N][dim] f32 row-major in HBM (the reference's Vec<Vec<f32>> / vamana_vectors.bin).
//
// Mapping: one LANE owns one ROW, so each lane runs the reference's sequential sum itself
// (scalar4: sum += ((a0b0+a1b1)+a2b2)+a3b3 without FMA; avx2: 8 FMA chains then a 0..7 lane sum).
// A wave stages 64 rows x 32 floats (64 full 128-B lines, coalesced f32x4 loads) through its
// own LDS slice and reads it back row-per-lane with conflict-free ds_read_b128 (row pitch 36
// floats = 9 sixteen-byte slots, odd => 16 distinct slots per lane group). Waves never barrier
// on each other inside the stream; the block only synchronises once per tile for the shared
// top-k buffer.
#include "common.h"
#include "topk.h"

#pragma clang fp contract(off)

namespace shodh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int EX_NT = 256;            // threads per block (4 waves)
constexpr uint32_t FLAT_ARRIVE_WORDS = 1024;   // capacity of the arrival counters of the in-scan merge (index.hip allocates them next to a workspace's single-query counter)
constexpr uint32_t EX_MAX_GROUPS = 32;  // query groups of one row range in flight at a time (1M rows: 1024 queries 42.7 ms with 128 groups in flight)
constexpr int EX_CHUNK = 32;          // floats of a row staged per step (one 128-B line)
constexpr int EX_PITCH = EX_CHUNK + 4;  // LDS row pitch in floats
// scoring op of the exact kernels: 0/1 = -dot in SHODH_ORDER_SCALAR4 / SHODH_ORDER_AVX2 order (flat index),
// 2/3 = SpannIndex::compute_distance (strictly sequential sums): 1 - dot, squared L2
constexpr int EX_OP_SEQ_ONE_MINUS_DOT = 2;
constexpr int EX_OP_SEQ_L2 = 3;

struct ExactArgs {
    const float *rows;
    uint64_t n_rows;
    uint32_t dim;
    const uint32_t *deleted;   // bitmask or nullptr
    const float *queries;      // [nq][dim]
    uint32_t nq;
    uint32_t k;
    uint32_t cap;              // top-k buffer capacity (pow2)
    uint64_t *partial;         // [query group][gridDim.x][QB][k]
    uint32_t id_base;
    const uint32_t *qlist;     // optional indirection: slot -> query index (device-side fallback list)
    const uint32_t *qcount;    // optional device-side number of slots (overrides nq)
    // Merge inside the scan (round 6, fallback mode of the pre-scan pipeline only): the workgroup that delivers the LAST partial list of a query group merges the
    // group's lists itself -- one launch instead of two for a step that nearly always finds its fallback list empty (two empty launches were 9.4 us of a 238 us step).
    uint32_t *arrive;          // [query groups] arrivals per group, zero between launches (the last arriver hands its counter back zeroed); nullptr = separate merge_topk_kernel
    uint32_t mcap;             // key buffer capacity of the merge (topk_capacity(k))
    uint32_t *ids;             // [nq][k] final results (arrive != nullptr)
    float *dist;
    uint32_t *counts;
};

// the merge of one query's partial lists (merge_topk_kernel's body; also run by the last arriver of flat_exact_kernel). LDS: keys[cap] | mins[2 EX_NT] | thr | cnt.
// Every thread of the block calls; ends with the results written (no trailing barrier).
__device__ __forceinline__ void merge_query_lists(const uint64_t *lists, uint32_t nlists, uint32_t qb, uint32_t k, uint32_t cap, uint32_t y, uint32_t qi, uint32_t q,
                                                  uint32_t *ids, float *dist, uint32_t *counts, unsigned char *lds) {
    uint64_t *keys = reinterpret_cast<uint64_t *>(lds);
    uint64_t *mins = keys + cap;
    uint64_t *thr = mins + 2 * EX_NT;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    const int tid = threadIdx.x;
    TopKBuf buf{keys, cnt, thr, cap, k};
    const uint64_t total = (uint64_t)nlists * k;
    auto key_at = [&](uint64_t e) -> uint64_t {
        const uint64_t l = e / k, i = e % k;
        return lists[((((uint64_t)y * nlists + l) * qb) + qi) * k + i];
    };
    const uint32_t m = block_select_topk<EX_NT>(key_at, total, buf, mins);
    for (uint32_t i = tid; i < k; i += EX_NT) {
        if (i < m) {
            const uint64_t key = buf.keys[i];
            ids[(size_t)q * k + i] = (uint32_t)key;
            dist[(size_t)q * k + i] = order_key_inv((uint32_t)(key >> 32));
        } else {
            ids[(size_t)q * k + i] = 0xFFFFFFFFu;
            dist[(size_t)q * k + i] = __builtin_inff();
        }
    }
    if (tid == 0) counts[q] = m;
}

// ---- fast kernel: dim % 32 == 0 -----------------------------------------------------------------
// One lane = one row (the reference's sums are serial chains per (query, row), so the parallelism is across rows and queries):
// a wave stages 64 rows x 32 floats through LDS (coalesced 128-B lines in, one row per lane out) and scores them against QB
// queries. The query values are wave-uniform, so they come through SCALAR loads straight from global memory into SGPRs and
// feed the VALU as scalar operands: no LDS traffic per query, which is what lets QB be 8 (an LDS-resident query block cost
// one broadcast read per query and 4-float group and capped QB at 2: 128 passes over the corpus for a 256-query batch, HBM-bound).
// Workgroups that share rows (same row range, different query group) are adjacent in dispatch order, so the re-reads of a
// row by the other query groups hit L2 / Infinity Cache.
template <int QB, int ORDER>
__global__ __launch_bounds__(EX_NT) void flat_exact_kernel(ExactArgs a, const float *__restrict__ qglob) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t dim = a.dim;
    float *stage = reinterpret_cast<float *>(smem);                       // [4 waves][64][EX_PITCH]
    uint64_t *keys = reinterpret_cast<uint64_t *>(stage + 4 * 64 * EX_PITCH);   // [QB][cap]
    uint64_t *thr = keys + (size_t)QB * a.cap;                            // [QB]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + QB);               // [QB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t nq_eff = a.qcount ? *a.qcount : a.nq;
    // dispatch order is x-fastest: consecutive workgroups take consecutive query groups of the SAME rows
    const uint32_t flat_id = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t bx = flat_id / gridDim.y, grp0 = flat_id % gridDim.y;

  for (uint32_t grp = grp0; grp * QB < nq_eff; grp += gridDim.y) {
    const uint32_t q0 = grp * QB;
    __syncthreads();
    if (tid < QB) { cnt[tid] = 0; thr[tid] = a.k ? KEY_NONE : 0; }
    __syncthreads();
    // wave-uniform base of every query of the group (a slot past the end re-reads the group's first query; its results are dropped)
    const float *qbase[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const uint32_t s_ = (q0 + q < nq_eff) ? q0 + q : q0;
        const uint32_t qi = a.qlist ? a.qlist[s_] : s_;
        qbase[q] = qglob + (size_t)qi * dim;
    }

    TopKBuf buf[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) { buf[q].keys = keys + (size_t)q * a.cap; buf[q].cnt = cnt + q; buf[q].thr = thr + q; buf[q].cap = a.cap; buf[q].k = a.k; }

    float *my = stage + wave * 64 * EX_PITCH;
    const uint64_t n_tiles = (a.n_rows + 63) / 64;
    const uint64_t wave_gid = (uint64_t)bx * 4 + wave;
    const uint64_t wave_stride = (uint64_t)gridDim.x * 4;
    const uint64_t n_iter = (n_tiles + wave_stride - 1) / wave_stride;
    const int nchunk = dim / EX_CHUNK;
    const int ld_row = lane >> 3;      // 0..7 : row within the 8-row group an instruction covers
    const int ld_c4 = lane & 7;        // f32x4 column within the 128-B line

    for (uint64_t it = 0; it < n_iter; ++it) {
        const uint64_t tile = wave_gid + it * wave_stride;
        const bool active = tile < n_tiles;
        if (active) {
            const uint64_t row0 = tile * 64;
            f32x4 pre[8];
            // prefetch chunk 0: instruction i covers rows 8i..8i+7 of the tile
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint64_t r = row0 + i * 8 + ld_row;
                if (r >= a.n_rows) r = a.n_rows - 1;
                pre[i] = *reinterpret_cast<const f32x4 *>(a.rows + r * dim + ld_c4 * 4);
            }
            float s[QB];
            float acc8[ORDER == SHODH_ORDER_AVX2 ? QB : 1][8];
#pragma unroll
            for (int q = 0; q < QB; ++q) s[q] = 0.0f;
            if (ORDER == SHODH_ORDER_AVX2) {
#pragma unroll
                for (int q = 0; q < QB; ++q)
#pragma unroll
                    for (int l = 0; l < 8; ++l) acc8[q][l] = 0.0f;
            }
            for (int c = 0; c < nchunk; ++c) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<f32x4 *>(my + (i * 8 + ld_row) * EX_PITCH + ld_c4 * 4) = pre[i];
                __builtin_amdgcn_wave_barrier();
                if (c + 1 < nchunk) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        uint64_t r = row0 + i * 8 + ld_row;
                        if (r >= a.n_rows) r = a.n_rows - 1;
                        pre[i] = *reinterpret_cast<const f32x4 *>(a.rows + r * dim + (c + 1) * EX_CHUNK + ld_c4 * 4);
                    }
                }
                f32x4 v[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) v[g] = *reinterpret_cast<const f32x4 *>(my + lane * EX_PITCH + g * 4);
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const float *qp = qbase[q] + c * EX_CHUNK;
                    if (ORDER == SHODH_ORDER_SCALAR4) {
                        // distance_inline.rs:165-168
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            f32x4 w = *reinterpret_cast<const f32x4 *>(qp + g * 4);
                            float t = w.x * v[g].x;
                            t = t + w.y * v[g].y;
                            t = t + w.z * v[g].z;
                            t = t + w.w * v[g].w;
                            s[q] = s[q] + t;
                        }
                    } else if (ORDER == EX_OP_SEQ_ONE_MINUS_DOT) {
                        // spann.rs:566-568: a.iter().zip(b).map(|(x, y)| x * y).sum()  (strictly sequential)
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            f32x4 w = *reinterpret_cast<const f32x4 *>(qp + g * 4);
                            s[q] = s[q] + w.x * v[g].x;
                            s[q] = s[q] + w.y * v[g].y;
                            s[q] = s[q] + w.z * v[g].z;
                            s[q] = s[q] + w.w * v[g].w;
                        }
                    } else if (ORDER == EX_OP_SEQ_L2) {
                        // spann.rs:564: map(|(x, y)| (x - y).powi(2)).sum()
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            f32x4 w = *reinterpret_cast<const f32x4 *>(qp + g * 4);
                            float d0 = w.x - v[g].x, d1 = w.y - v[g].y, d2 = w.z - v[g].z, d3 = w.w - v[g].w;
                            s[q] = s[q] + d0 * d0;
                            s[q] = s[q] + d1 * d1;
                            s[q] = s[q] + d2 * d2;
                            s[q] = s[q] + d3 * d3;
                        }
                    } else {
                        // distance_inline.rs:77-97: one 8-lane FMA accumulator
#pragma unroll
                        for (int g = 0; g < 8; g += 2) {
                            f32x4 w0 = *reinterpret_cast<const f32x4 *>(qp + g * 4);
                            f32x4 w1 = *reinterpret_cast<const f32x4 *>(qp + g * 4 + 4);
                            acc8[q][0] = __builtin_fmaf(w0.x, v[g].x, acc8[q][0]);
                            acc8[q][1] = __builtin_fmaf(w0.y, v[g].y, acc8[q][1]);
                            acc8[q][2] = __builtin_fmaf(w0.z, v[g].z, acc8[q][2]);
                            acc8[q][3] = __builtin_fmaf(w0.w, v[g].w, acc8[q][3]);
                            acc8[q][4] = __builtin_fmaf(w1.x, v[g + 1].x, acc8[q][4]);
                            acc8[q][5] = __builtin_fmaf(w1.y, v[g + 1].y, acc8[q][5]);
                            acc8[q][6] = __builtin_fmaf(w1.z, v[g + 1].z, acc8[q][6]);
                            acc8[q][7] = __builtin_fmaf(w1.w, v[g + 1].w, acc8[q][7]);
                        }
                    }
                }
            }
            const uint64_t row = row0 + lane;
            bool live = row < a.n_rows;
            if (live && a.deleted) live = ((a.deleted[row >> 5] >> (row & 31)) & 1u) == 0;
            if (live) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    float dist;
                    if (ORDER == SHODH_ORDER_SCALAR4) dist = -s[q];                      // distance_inline.rs:479-481
                    else if (ORDER == EX_OP_SEQ_ONE_MINUS_DOT) dist = 1.0f - s[q];     // spann.rs:568
                    else if (ORDER == EX_OP_SEQ_L2) dist = s[q];
                    else {
                        // distance_inline.rs:100-108: lanes summed 0 -> 7
                        float r = acc8[q][0] + acc8[q][1];
                        r = r + acc8[q][2]; r = r + acc8[q][3]; r = r + acc8[q][4];
                        r = r + acc8[q][5]; r = r + acc8[q][6]; r = r + acc8[q][7];
                        dist = -r;
                    }
                    if (q0 + q < nq_eff) topk_push(buf[q], make_key(dist, a.id_base + (uint32_t)row));
                }
            }
        }
        // at most 256 pushes per query per iteration: keep 256 free slots. The counters are read between two barriers so
        // that every thread sees the same values (see topk_compact_if_short).
        __syncthreads();
        uint32_t filled[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) filled[q] = *buf[q].cnt;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if (filled[q] + EX_NT > a.cap) topk_compact<EX_NT>(buf[q]);   // block-uniform condition
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        topk_compact<EX_NT>(buf[q]);
        uint64_t *out = a.partial + (((size_t)grp * gridDim.x + bx) * QB + q) * a.k;
        const uint32_t m = *buf[q].cnt;
        for (uint32_t i = tid; i < a.k; i += EX_NT) out[i] = (i < m) ? buf[q].keys[i] : KEY_NONE;
    }
    if (a.arrive) {
        // the group's lists of this row range are out: count the arrival; the last of the gridDim.x row ranges merges the group
        __threadfence();                       // (each thread: its part of the lists is visible device-wide before the arrival is)
        __syncthreads();
        if (tid == 0) {
            const uint32_t old = atomicAdd(&a.arrive[grp], 1u);
            const bool last = old + 1u == gridDim.x;
            if (last) a.arrive[grp] = 0u;      // handed back zeroed (nobody else touches this group's counter any more in this launch)
            cnt[0] = last ? 1u : 0u;
        }
        __syncthreads();
        const bool last = cnt[0] != 0u;        // block-uniform
        __syncthreads();
        if (last) {
            __threadfence();                   // (acquire: the other workgroups' lists)
#pragma unroll 1
            for (int q = 0; q < QB; ++q) {
                if (q0 + q >= nq_eff) break;
                const uint32_t s_ = q0 + q;
                const uint32_t qi = a.qlist ? a.qlist[s_] : s_;
                __syncthreads();
                merge_query_lists(a.partial, gridDim.x, QB, a.k, a.mcap, grp, (uint32_t)q, qi, a.ids, a.dist, a.counts, smem);      // (the scan's LDS is free: merge buffers from its base)
            }
            __syncthreads();
        }
    }
  }
}

// ---- generic kernel: any dim (tiny reference test fixtures, dim % 32 != 0). One row per thread,
// rows read straight from global memory. Same arithmetic order. ------------------------------------
template <int ORDER>
__global__ __launch_bounds__(EX_NT) void flat_exact_generic_kernel(ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t dim = a.dim;
    float *qs = reinterpret_cast<float *>(smem);                                   // [dim]
    uint64_t *keys = reinterpret_cast<uint64_t *>(qs + ((dim + 3) & ~3u));         // [cap]
    uint64_t *thr = keys + a.cap;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    const int tid = threadIdx.x;
    const uint32_t nq_eff = a.qcount ? *a.qcount : a.nq;
  for (uint32_t grp = blockIdx.y; grp < nq_eff; grp += gridDim.y) {
    const uint32_t q = a.qlist ? a.qlist[grp] : grp;
    __syncthreads();
    for (uint32_t i = tid; i < dim; i += EX_NT) qs[i] = a.queries[(size_t)q * dim + i];
    if (tid == 0) { *cnt = 0; *thr = a.k ? KEY_NONE : 0; }
    __syncthreads();
    TopKBuf buf{keys, cnt, thr, a.cap, a.k};
    const uint64_t n_iter = (a.n_rows + (uint64_t)gridDim.x * EX_NT - 1) / ((uint64_t)gridDim.x * EX_NT);
    for (uint64_t it = 0; it < n_iter; ++it) {
        const uint64_t row = (it * gridDim.x + blockIdx.x) * EX_NT + tid;
        bool live = row < a.n_rows;
        if (live && a.deleted) live = ((a.deleted[row >> 5] >> (row & 31)) & 1u) == 0;
        if (live) {
            const float *r = a.rows + row * dim;
            float dist;
            if (ORDER == SHODH_ORDER_SCALAR4) {
                const uint32_t un = dim & ~3u;
                float sum = 0.0f;
                for (uint32_t i = 0; i < un; i += 4) {
                    float t = qs[i] * r[i];
                    t = t + qs[i + 1] * r[i + 1];
                    t = t + qs[i + 2] * r[i + 2];
                    t = t + qs[i + 3] * r[i + 3];
                    sum = sum + t;
                }
                for (uint32_t j = un; j < dim; ++j) sum = sum + qs[j] * r[j];
                dist = -sum;
            } else if (ORDER == EX_OP_SEQ_ONE_MINUS_DOT) {
                float sum = 0.0f;
                for (uint32_t i = 0; i < dim; ++i) sum = sum + qs[i] * r[i];
                dist = 1.0f - sum;
            } else if (ORDER == EX_OP_SEQ_L2) {
                float sum = 0.0f;
                for (uint32_t i = 0; i < dim; ++i) { const float d = qs[i] - r[i]; sum = sum + d * d; }
                dist = sum;
            } else {
                const uint32_t sn = dim & ~7u;
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t i = 0; i < sn; i += 8)
#pragma unroll
                    for (int l = 0; l < 8; ++l) acc[l] = __builtin_fmaf(qs[i + l], r[i + l], acc[l]);
                float rr = acc[0] + acc[1];
                rr = rr + acc[2]; rr = rr + acc[3]; rr = rr + acc[4]; rr = rr + acc[5]; rr = rr + acc[6]; rr = rr + acc[7];
                for (uint32_t j = sn; j < dim; ++j) rr = rr + qs[j] * r[j];
                dist = -rr;
            }
            topk_push(buf, make_key(dist, a.id_base + (uint32_t)row));
        }
        topk_compact_if_short<EX_NT>(buf, EX_NT);
    }
    __syncthreads();
    topk_compact<EX_NT>(buf);
    uint64_t *out = a.partial + ((size_t)grp * gridDim.x + blockIdx.x) * a.k;
    const uint32_t m = *buf.cnt;
    for (uint32_t i = tid; i < a.k; i += EX_NT) out[i] = (i < m) ? buf.keys[i] : KEY_NONE;
  }
}

// ---- merge: one block per query folds `nlists` sorted k-lists into the final top-k ---------------
struct MergeArgs {
    const uint64_t *lists;   // query q, list l at lists[(q_group(q)...)] -- see index math below
    uint32_t nlists;         // lists per query
    uint32_t qb;             // queries per scan block (layout [y][x][qb][k])
    uint32_t k;
    uint32_t cap;
    uint32_t nq;
    uint32_t *ids;           // [nq][k]
    float *dist;             // [nq][k]
    uint32_t *counts;        // [nq]
    const uint32_t *qlist;   // optional slot -> query index
    const uint32_t *qcount;  // optional device-side slot count
};

__global__ __launch_bounds__(EX_NT) void merge_topk_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t slot = blockIdx.x;
    const uint32_t nq_eff = a.qcount ? *a.qcount : a.nq;
    if (slot >= nq_eff) return;
    const uint32_t q = a.qlist ? a.qlist[slot] : slot;
    merge_query_lists(a.lists, a.nlists, a.qb, a.k, a.cap, slot / a.qb, slot % a.qb, q, a.ids, a.dist, a.counts, smem);
}

// ---- merge of per-shard results (multi-GPU): lists of (id, dist) rows gathered from every rank ----
// in_ids/in_dist: [n_lists][nq][k] (entries with id 0xFFFFFFFF are padding). One block per query.
struct MergeListsArgs {
    const uint32_t *in_ids;
    const float *in_dist;
    uint32_t n_lists, nq, k, cap;
    uint64_t list_stride;     // elements between the blocks of consecutive lists (nq*k when they are dense)
    uint32_t *ids;
    float *dist;
    uint32_t *counts;
};
__global__ __launch_bounds__(EX_NT) void merge_lists_kernel(MergeListsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *mins = keys + a.cap;
    uint64_t *thr = mins + 2 * EX_NT;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    TopKBuf buf{keys, cnt, thr, a.cap, a.k};
    const uint64_t total = (uint64_t)a.n_lists * a.k;
    auto key_at = [&](uint64_t e) -> uint64_t {
        const uint64_t l = e / a.k, i = e % a.k;
        const size_t off = (size_t)l * a.list_stride + (size_t)q * a.k + i;
        const uint32_t id = a.in_ids[off];
        return id == 0xFFFFFFFFu ? KEY_NONE : make_key(a.in_dist[off], id);
    };
    const uint32_t m = block_select_topk<EX_NT>(key_at, total, buf, mins);
    for (uint32_t i = tid; i < a.k; i += EX_NT) {
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

// ---- host launchers -------------------------------------------------------------------------------
// per-query key buffer of the scan: the k kept keys plus one iteration's pushes (one per thread)
static uint32_t exact_scan_cap(uint32_t k) { return next_pow2(k + (uint32_t)EX_NT); }
static size_t exact_lds_bytes(int qb, uint32_t k) {
    return (size_t)4 * 64 * EX_PITCH * 4 + (size_t)qb * exact_scan_cap(k) * 8 + (size_t)qb * 8 + (size_t)qb * 4 + 16;
}

uint32_t topk_capacity(uint32_t k) {
    uint32_t need = k + 2 * EX_NT;
    if (need < 2 * k) need = 2 * k;
    if (need < 1024) need = 1024;
    return next_pow2(need);
}

// choose queries-per-block so that LDS stays <= 64 KiB (>= 2 blocks per CU)
int exact_pick_qb(uint32_t nq, uint32_t dim, uint32_t k) {
    (void)dim;
    int qb = 8;      // two workgroups per CU: <= 80 KiB each
    while (qb > 1 && (exact_lds_bytes(qb, k) > 80 * 1024 || (uint32_t)qb > next_pow2(nq))) qb >>= 1;
    return qb;
}

size_t exact_partial_bytes(uint32_t nq, uint32_t dim, uint32_t k, uint32_t grid_x) {
    const int qb = (dim % 32 == 0) ? exact_pick_qb(nq, dim, k) : 1;
    const uint32_t gy = (uint32_t)ceil_div(nq, qb);
    return (size_t)gy * grid_x * qb * (k ? k : 1) * 8;
}

uint32_t exact_grid_x(uint64_t n_rows, uint32_t nq, uint32_t k, int cus) {
    uint64_t blocks = ceil_div(n_rows, EX_NT);
    uint64_t cap_blocks = (uint64_t)cus * 4;
    // Batches: the grid is (row ranges) x (query groups). About 16 workgroups per CU in total is enough to balance; more row
    // ranges only add per-workgroup set-up, partial lists and merge work (1M rows, 256 queries: 1024 ranges 11.0 ms, 128 6.3 ms).
    uint64_t groups = ceil_div(nq, (uint32_t)exact_pick_qb(nq, 0, k));
    if (groups > EX_MAX_GROUPS) groups = EX_MAX_GROUPS;
    uint64_t want = ceil_div((uint64_t)cus * 16, groups ? groups : 1);
    if (want < 16) want = 16;
    if (cap_blocks > want) cap_blocks = want;
    // keep the partial-result buffer modest for big nq*k
    while (cap_blocks > 64 && cap_blocks * (uint64_t)nq * (k ? k : 1) * 8 > (64ull << 20)) cap_blocks >>= 1;
    if (blocks > cap_blocks) blocks = cap_blocks;
    if (blocks < 1) blocks = 1;
    return (uint32_t)blocks;
}

template <int QB, int OP>
static int launch_fast_op(const ExactArgs &a, dim3 grid, size_t lds, hipStream_t st) {
    SHODH_TRY(ensure_dynamic_lds((const void *)flat_exact_kernel<QB, OP>, lds));
    hipLaunchKernelGGL((flat_exact_kernel<QB, OP>), grid, dim3(EX_NT), lds, st, a, a.queries);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
template <int QB>
static int launch_fast(const ExactArgs &a, uint32_t op, dim3 grid, size_t lds, hipStream_t st) {
    switch (op) {
        case SHODH_ORDER_AVX2: return launch_fast_op<QB, SHODH_ORDER_AVX2>(a, grid, lds, st);
        case EX_OP_SEQ_ONE_MINUS_DOT: return launch_fast_op<QB, EX_OP_SEQ_ONE_MINUS_DOT>(a, grid, lds, st);
        case EX_OP_SEQ_L2: return launch_fast_op<QB, EX_OP_SEQ_L2>(a, grid, lds, st);
        default: return launch_fast_op<QB, SHODH_ORDER_SCALAR4>(a, grid, lds, st);
    }
}
template <int OP>
static int launch_generic_op(const ExactArgs &a, dim3 grid, size_t lds, hipStream_t st) {
    SHODH_TRY(ensure_dynamic_lds((const void *)flat_exact_generic_kernel<OP>, lds));
    hipLaunchKernelGGL((flat_exact_generic_kernel<OP>), grid, dim3(EX_NT), lds, st, a);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

// Scans all rows for nq queries and writes final top-k (ids, dist, counts) on `st`.
// `partial` must hold exact_partial_bytes(...) bytes. With qlist/qcount (device pointers) the
// scan runs over the device-side list of query slots instead (fallback of the MFMA path): the
// grid is sized for `nq` slots at most and blocks beyond *qcount exit at once.
// `arrive` (optional; FLAT_ARRIVE_WORDS zeroed words that are zero again when the launch is done): the scan merges its own partial lists (ExactArgs::arrive) and no
// merge kernel is launched -- taken for the fast kernel when the query groups fit the counter array, otherwise the two launches as before.
int launch_flat_exact_arrive(const float *rows, uint64_t n_rows, uint32_t dim, const uint32_t *deleted,
                             const float *d_queries, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                             uint64_t *partial, uint32_t grid_x, uint32_t *d_ids, float *d_dist, uint32_t *d_counts,
                             const uint32_t *qlist, const uint32_t *qcount, uint32_t *arrive, hipStream_t st);
int launch_flat_exact(const float *rows, uint64_t n_rows, uint32_t dim, const uint32_t *deleted,
                      const float *d_queries, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                      uint64_t *partial, uint32_t grid_x, uint32_t *d_ids, float *d_dist, uint32_t *d_counts,
                      const uint32_t *qlist, const uint32_t *qcount, hipStream_t st) {
    return launch_flat_exact_arrive(rows, n_rows, dim, deleted, d_queries, nq, k, order, id_base, partial, grid_x, d_ids, d_dist, d_counts, qlist, qcount, nullptr, st);
}
int launch_flat_exact_arrive(const float *rows, uint64_t n_rows, uint32_t dim, const uint32_t *deleted,
                             const float *d_queries, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                             uint64_t *partial, uint32_t grid_x, uint32_t *d_ids, float *d_dist, uint32_t *d_counts,
                             const uint32_t *qlist, const uint32_t *qcount, uint32_t *arrive, hipStream_t st) {
    if (nq == 0) return SHODH_OK;
    const uint32_t cap = topk_capacity(k);
    ExactArgs a{rows, n_rows, dim, deleted, d_queries, nq, k, cap, partial, id_base, qlist, qcount, nullptr, cap, d_ids, d_dist, d_counts};
    const size_t mlds = (size_t)cap * 8 + 2 * EX_NT * 8 + 8 + 4 + 16;
    int qb = 1;
    if (dim % 32 == 0) {
        qb = exact_pick_qb(nq, dim, k);
        uint32_t gy = (uint32_t)ceil_div(nq, qb);
        static const bool fuse_ok = !(getenv("SHODH_EXACT_FUSED_MERGE") && atoi(getenv("SHODH_EXACT_FUSED_MERGE")) == 0);
        if (arrive && fuse_ok && gy <= FLAT_ARRIVE_WORDS) a.arrive = arrive;
        if (qcount && gy > 4) gy = 4;          // fallback mode: few resident groups, they loop
        if (gy > EX_MAX_GROUPS) gy = EX_MAX_GROUPS;   // more groups loop: the rows a round of groups shares stay cache-resident
        dim3 grid(grid_x, gy);
        a.cap = exact_scan_cap(k);
        size_t lds = exact_lds_bytes(qb, k);
        if (a.arrive && mlds > lds) lds = mlds;      // (the last arriver's merge buffers alias the scan's)
        if (lds > 160 * 1024) { set_error("k=%u too large for the exact scan (LDS %zu B)", k, lds); return SHODH_ERR_UNSUPPORTED; }
        switch (qb) {
            case 8: SHODH_TRY(launch_fast<8>(a, order, grid, lds, st)); break;
            case 4: SHODH_TRY(launch_fast<4>(a, order, grid, lds, st)); break;
            case 2: SHODH_TRY(launch_fast<2>(a, order, grid, lds, st)); break;
            default: SHODH_TRY(launch_fast<1>(a, order, grid, lds, st)); break;
        }
    } else {
        uint32_t gy = nq;
        if (qcount && gy > 4) gy = 4;
        dim3 grid(grid_x, gy);
        const size_t lds = (size_t)((dim + 3) & ~3u) * 4 + (size_t)cap * 8 + 8 + 4 + 16;
        if (lds > 160 * 1024) { set_error("dim/k too large for the generic exact scan"); return SHODH_ERR_UNSUPPORTED; }
        switch (order) {
            case SHODH_ORDER_AVX2: SHODH_TRY(launch_generic_op<SHODH_ORDER_AVX2>(a, grid, lds, st)); break;
            case EX_OP_SEQ_ONE_MINUS_DOT: SHODH_TRY(launch_generic_op<EX_OP_SEQ_ONE_MINUS_DOT>(a, grid, lds, st)); break;
            case EX_OP_SEQ_L2: SHODH_TRY(launch_generic_op<EX_OP_SEQ_L2>(a, grid, lds, st)); break;
            default: SHODH_TRY(launch_generic_op<SHODH_ORDER_SCALAR4>(a, grid, lds, st)); break;
        }
    }
    if (a.arrive) return SHODH_OK;      // merged by the scan itself
    MergeArgs m{partial, grid_x, (uint32_t)qb, k, cap, nq, d_ids, d_dist, d_counts, qlist, qcount};
    SHODH_TRY(ensure_dynamic_lds((const void *)merge_topk_kernel, mlds));
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(EX_NT), mlds, st, m);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

int launch_merge_lists(const uint32_t *in_ids, const float *in_dist, uint64_t list_stride, uint32_t n_lists, uint32_t nq, uint32_t k,
                       uint32_t *ids, float *dist, uint32_t *counts, hipStream_t st) {
    if (nq == 0) return SHODH_OK;
    const uint32_t cap = topk_capacity(k);
    MergeListsArgs a{in_ids, in_dist, n_lists, nq, k, cap, list_stride, ids, dist, counts};
    const size_t lds = (size_t)cap * 8 + 2 * EX_NT * 8 + 8 + 4 + 16;
    SHODH_TRY(ensure_dynamic_lds((const void *)merge_lists_kernel, lds));
    hipLaunchKernelGGL(merge_lists_kernel, dim3(nq), dim3(EX_NT), lds, st, a);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

}  // namespace shodh

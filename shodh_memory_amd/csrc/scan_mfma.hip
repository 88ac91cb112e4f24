// scan_mfma.hip -- batched flat scan on the matrix cores with EXACT results.
//
// Goal: VamanaIndex::brute_force_search (src/vector_db/vamana.rs:1167-1188) for up to 256
// queries per pass while reading the corpus ONCE per pass, bit-exact ids and distances.
//
// The reference's score is an f32 sum in a fixed order without FMA (distance_inline.rs:157-173);
// no matrix instruction reproduces that rounding, and doing it on the VALU for 256 queries is
// ~13x off the HBM roofline. So the scan is split:
//   1. pre-scan  (this file, MFMA): approximate scores s~ = fp16(256 q) . fp16(256 c) / 2^16 with
//      f32 accumulation on v_mfma_f32_32x32x16_f16. |s~ - dot_ref| <= eps(q) is PROVEN
//      (eps derivation in DESIGN.md; it assumes the worst case for fp16 denormals and for the
//      accumulation order), so  {rows : s~ >= kth_best(s~) - 2 eps}  contains the exact top-k.
//   2. re-score  (final_stage_kernel): the few survivors are scored again from the f32 rows in
//      the reference's exact accumulation order and ordered by (dist total_cmp, id).
// The pre-scan reads the fp16 shadow copy of the corpus (dim*2 bytes per row).
//
// Pre-scan kernel: persistent, one 512-thread workgroup per CU, wave w owns queries
// [32w, 32w+32) as 32x32x16 MFMA B-fragments held in registers for the whole launch (dim/16
// fragments of 4 VGPRs); corpus tiles of 64 rows stream HBM -> registers -> LDS (double
// buffered, 16-B chunks XOR-swizzled by row&15 so the 16-lane ds_read_b128 groups hit 16
// distinct slots) and every wave multiplies the whole tile against its 32 queries. MFMA C layout
// puts one query per lane (col = lane&31) and 16 corpus rows per lane in registers, so the
// top-k filter is a register compare against a per-lane threshold; survivors (rare) are appended
// to per-query candidate lists with one atomic each.
//
// Thresholds come from a 1/32 strided sample of the tiles (same kernel, BLOCKMAX epilogue): the
// k-th largest of the per-16-row maxima is a valid lower bound of the k-th best score.
#include <cstdlib>

#include "common.h"
#include "topk.h"

#pragma clang fp contract(off)

namespace shodh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_EQ_CAP = 2048;   // LDS emit-queue entries per workgroup
constexpr int MF_TR = 64;        // corpus rows per tile
constexpr int MF_BPAD = 256;     // queries per pass
constexpr float MF_SCALE = 256.0f;                 // rows and queries are stored as fp16(256 x)
constexpr float MF_INV_SCALE2 = 1.0f / 65536.0f;   // acc -> score

enum { MF_MODE_BLOCKMAX = 0, MF_MODE_EMIT = 1 };

struct MfmaArgs {
    const _Float16 *rows_h;   // [n_rows][dim] fp16(256*x); tombstoned rows are zero
    uint64_t n_rows;
    uint32_t dim;
    const _Float16 *q_h;      // [passes][256][dim]
    const float *thr;         // [passes][256] emit threshold on the true score scale
    const uint32_t *deleted;  // bitmask or nullptr
    uint64_t *cand;           // [passes][256][cand_cap]  key = (order_key(-s~) << 32) | local row
    uint32_t *cand_cnt;       // [passes][256]
    uint32_t cand_cap;
    float *blockmax;          // [passes][J][256], J = n_sel_tiles * 4
    uint32_t tile_stride;     // tile index = sel * tile_stride
    uint32_t n_sel_tiles;
    uint32_t ablate;          // diagnostics only (SHODH_ABLATE): 1 = skip the MFMA phase, 2 = skip the HBM loads
};

// QB = 32-query blocks per wave: QB=1 -> 8 waves (2 per SIMD, 256 registers each), QB=2 -> 4 waves
// (1 per SIMD, 512 registers each; every A fragment read from LDS feeds two MFMAs).
template <int MODE, int KSTEPS, int QB>
__global__ __launch_bounds__(512 / QB, 2 / QB) void mfma_scan_kernel(MfmaArgs a) {
    constexpr int NT = 512 / QB;
    constexpr int DIM = KSTEPS * 16;
    constexpr int CPR = KSTEPS * 2;          // 16-B chunks per row
    constexpr int PITCH = DIM * 2;           // bytes per row in LDS
    constexpr int CPT = MF_TR * CPR / NT;    // chunks per thread per tile
    constexpr int TILE_BYTES = MF_TR * PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: [2 tiles][emit queue: MF_EQ_CAP x (key u64, query u32)][queue counter]
    uint64_t *eq_key = reinterpret_cast<uint64_t *>(smem + 2 * TILE_BYTES);
    uint32_t *eq_q = reinterpret_cast<uint32_t *>(eq_key + MF_EQ_CAP);
    uint32_t *eq_cnt = eq_q + MF_EQ_CAP;
    uint32_t *eq_flag = eq_cnt + 1;          // [2] double-buffered "flush now" decision (block-uniform)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const uint32_t pass = blockIdx.y;
    const uint32_t q_base = wave * 32 * QB + l31;       // this lane's queries within the pass: q_base + 32*qb
    if (tid == 0) *eq_cnt = 0;

    // resident B fragments: query q_base + 32*qb, k = ks*16 + hi*8 .. +8
    half8 bq[QB][KSTEPS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const _Float16 *qp = a.q_h + ((size_t)pass * MF_BPAD + q_base + 32 * qb) * DIM + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) bq[qb][ks] = *reinterpret_cast<const half8 *>(qp + ks * 16);
    }
    float thr_l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
        thr_l[qb] = (MODE == MF_MODE_EMIT) ? a.thr[(size_t)pass * MF_BPAD + q_base + 32 * qb] * (MF_SCALE * MF_SCALE) : 0.0f;

    u32x4 pre[CPT];
    auto prefetch = [&](uint32_t sel) {
        const uint64_t row0 = (uint64_t)sel * a.tile_stride * MF_TR;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int S = i * NT + tid;
            const int row = S / CPR, c = S % CPR;
            uint64_t r = row0 + row;
            if (r >= a.n_rows) r = a.n_rows - 1;
            pre[i] = *reinterpret_cast<const u32x4 *>(a.rows_h + r * DIM + c * 8);
        }
    };
    auto stage = [&](unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int S = i * NT + tid;
            const int row = S / CPR, c = S % CPR;
            *reinterpret_cast<u32x4 *>(buf + row * PITCH + ((c ^ (row & 15)) << 4)) = pre[i];
        }
    };
    // drains the LDS emit queue into the per-query candidate lists (global atomics). Called by the
    // whole block at points where no prefetch is in flight, so its vmcnt waits cost only themselves.
    bool eq_overflowed = false;              // block-uniform: entries were dropped at some point
    auto flush = [&]() {                     // caller: __syncthreads() before, and after before the next push
        const uint32_t raw = *eq_cnt;
        eq_overflowed |= raw > (uint32_t)MF_EQ_CAP;
        const uint32_t n = raw < (uint32_t)MF_EQ_CAP ? raw : (uint32_t)MF_EQ_CAP;
        for (uint32_t i = tid; i < n; i += NT) {
            const uint64_t key = eq_key[i];
            const uint32_t row = (uint32_t)key;
            if (a.deleted && ((a.deleted[row >> 5] >> (row & 31)) & 1u)) continue;   // tombstoned (vamana.rs:1175-1177)
            const size_t qi = (size_t)pass * MF_BPAD + eq_q[i];
            const uint32_t slot = atomicAdd(a.cand_cnt + qi, 1u);
            if (slot < a.cand_cap) a.cand[qi * a.cand_cap + slot] = key;
        }
        __syncthreads();
        if (tid == 0) *eq_cnt = 0;
    };

    uint32_t sel = blockIdx.x;
    if (sel < a.n_sel_tiles) prefetch(sel);
    if (sel < a.n_sel_tiles) stage(smem);
    // Everything issued so far (query fragments, thresholds, first tile) must have landed before
    // the stream starts: otherwise hipcc re-waits for the fragment loads INSIDE the loop with
    // counted vmcnt(N), which in steady state drains the next tile's prefetch half-way through
    // the MFMA chain and exposes the HBM latency on every tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    // LDS byte offsets of this lane's A fragments. chunk c = 2*ks + hi; swizzled chunk = c ^ (row&15)
    // = (c & ~15) | ((c & 15) ^ sw): only 8 distinct low parts per lane, the rest is an immediate.
    const int sw = l31 & 15;
    int aoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aoff[j] = l31 * PITCH + (((2 * j + hi) ^ sw) << 4);
    int cur = 0;
    for (; sel < a.n_sel_tiles; sel += gridDim.x) {
        const uint32_t nxt = sel + gridDim.x;
        const bool has_next = nxt < a.n_sel_tiles;
        if (has_next && !(a.ablate & 2)) prefetch(nxt);
        const unsigned char *buf = smem + cur * TILE_BYTES;
        const uint64_t tile_row0 = (uint64_t)sel * a.tile_stride * MF_TR;
        floatx16 acc[2][QB];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][qb][r] = 0.0f;
        if (!(a.ablate & 1)) {
            // A fragments are read from LDS one group (GS k-steps, 2*GS reads) ahead of the MFMAs that
            // consume them, in two register sets. Left alone hipcc issues each read right before its
            // MFMA and the chain runs at LDS latency.
            constexpr int GS = 2;
            constexpr int NG = KSTEPS / GS;
            half8 fa[2][GS][2];
            auto load_group = [&](int slot, int g) {
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    const int ks = g * GS + j;
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        fa[slot][j][rb] = *reinterpret_cast<const half8 *>(buf + rb * 32 * PITCH + aoff[ks & 7] + (ks >> 3) * 256);
                }
            };
            load_group(0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group((g + 1) & 1, g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < GS; ++j)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            acc[rb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][j][rb], bq[qb][g * GS + j], acc[rb][qb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // C layout (32x32): col = lane&31 (query), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const floatx16 &c = acc[rb][qb];
                const uint32_t q_local = q_base + 32 * qb;
                if (MODE == MF_MODE_EMIT) {
                    float m = c[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, c[r]);
                    if (m >= thr_l[qb]) {
                        // rare: survivors go to the workgroup's LDS queue (LDS atomic, no HBM round trip)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (c[r] >= thr_l[qb]) {
                                const uint64_t grow = tile_row0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                                if (grow < a.n_rows) {
                                    const uint32_t slot = atomicAdd(eq_cnt, 1u);
                                    if (slot < (uint32_t)MF_EQ_CAP) {
                                        eq_key[slot] = make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)grow);
                                        eq_q[slot] = q_local;
                                    }
                                }
                            }
                        }
                    }
                } else {
                    float m;
                    if (tile_row0 + MF_TR <= a.n_rows) {
                        m = c[0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = fmaxf(m, c[r]);
                    } else {
                        m = -__builtin_inff();
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const uint64_t grow = tile_row0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            if (grow < a.n_rows) m = fmaxf(m, c[r]);
                        }
                    }
                    const size_t j = ((size_t)sel * 2 + rb) * 2 + hi;
                    a.blockmax[((size_t)pass * a.n_sel_tiles * 4 + j) * MF_BPAD + q_local] = m * MF_INV_SCALE2;
                }
            }
        }
        if (has_next) stage(smem + (cur ^ 1) * TILE_BYTES);
        // Flush the queue early when it is half full. The decision must be block-uniform: thread 0
        // snapshots it into a double-buffered flag BEFORE the barrier, everyone reads that slot after it
        // (the slot is rewritten two barriers later). A stale (low) count only delays the flush;
        // dropped entries are detected by the counter itself (eq_overflowed).
        if (MODE == MF_MODE_EMIT && tid == 0) eq_flag[cur] = (*eq_cnt > (uint32_t)MF_EQ_CAP / 2) ? 1u : 0u;
        __syncthreads();
        if (MODE == MF_MODE_EMIT) {
            if (eq_flag[cur]) { flush(); __syncthreads(); }
        }
        cur ^= 1;
    }
    if (MODE == MF_MODE_EMIT) {
        __syncthreads();
        flush();
        // entries were dropped somewhere: poison every list of this pass so that the final stage sends
        // those queries to the exact scan (adversarial inputs only, e.g. thousands of identical rows)
        if (eq_overflowed && tid < MF_BPAD) atomicAdd(a.cand_cnt + (size_t)pass * MF_BPAD + tid, a.cand_cap + 1u);
    }
}

// ---- conversions ------------------------------------------------------------------------------------
// rows f32 -> fp16(256 x) shadow; also folds max row norm^2, max |x| and a non-finite flag into stats.
// stats[0] = max norm^2 (float bits, atomicMax on uint works for non-negative floats)
// stats[1] = max |x| (float bits), stats[2] = non-finite count
__global__ __launch_bounds__(256) void convert_rows_kernel(const float *rows, uint64_t first, uint64_t n, uint32_t dim,
                                                           _Float16 *rows_h, uint32_t *stats) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint64_t wave_cnt = ((uint64_t)gridDim.x * 256) >> 6;
    float mx_n = 0.0f, mx_a = 0.0f;
    uint32_t bad = 0;
    for (uint64_t r = first + wave_gid; r < first + n; r += wave_cnt) {
        const float *src = rows + r * dim;
        _Float16 *dst = rows_h + r * dim;
        float ss = 0.0f;
        for (uint32_t i = lane * 4; i < dim; i += 256) {   // dim % 4 == 0 on this path
            const float4 v = *reinterpret_cast<const float4 *>(src + i);
            const float e[4] = {v.x, v.y, v.z, v.w};
            _Float16 h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = e[j];
                if (!(__builtin_fabsf(x) <= 3.0e38f)) bad++;   // NaN or Inf
                mx_a = fmaxf(mx_a, __builtin_fabsf(x));
                ss = __builtin_fmaf(x, x, ss);
                h[j] = (_Float16)(x * MF_SCALE);
            }
            *reinterpret_cast<uint2 *>(dst + i) = *reinterpret_cast<const uint2 *>(h);
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        mx_n = fmaxf(mx_n, ss);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx_a = fmaxf(mx_a, __shfl_xor(mx_a, off));
        bad += __shfl_xor(bad, off);
    }
    if (lane == 0) {
        atomicMax(stats + 0, __float_as_uint(mx_n));
        atomicMax(stats + 1, __float_as_uint(mx_a));
        if (bad) atomicAdd(stats + 2, bad);
    }
}

// zero (tombstone) or restore one row of the shadow copy
__global__ void shadow_set_row_kernel(const float *rows, _Float16 *rows_h, uint64_t row, uint32_t dim, int zero) {
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x)
        rows_h[row * dim + i] = zero ? (_Float16)0.0f : (_Float16)(rows[row * dim + i] * MF_SCALE);
}
// restore every row whose tombstone bit is set (clear_deleted)
__global__ void shadow_restore_deleted_kernel(const float *rows, _Float16 *rows_h, const uint32_t *deleted, uint64_t n, uint32_t dim) {
    const uint64_t row = blockIdx.x;
    if (row >= n) return;
    if (((deleted[row >> 5] >> (row & 31)) & 1u) == 0) return;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) rows_h[row * dim + i] = (_Float16)(rows[row * dim + i] * MF_SCALE);
}

// queries f32 [nq][dim] -> fp16 [passes*256][dim] (zero padded); per-query norm and flags; zeroes
// the per-query counters. One wave per query slot.
struct QueryPrep {
    const float *q;          // [nq][dim]
    uint32_t nq, dim, n_slots;
    _Float16 *q_h;           // [n_slots][dim]
    float *qnorm;            // [n_slots]
    uint32_t *cand_cnt;      // [n_slots]
    uint32_t *fallback;      // [n_slots] 1 = must go through the exact scan
    uint32_t *fb_count;      // single counter, zeroed here
    uint32_t *stats;         // [4] zeroed here: emitted, rescored, overflowed, -
};
__global__ __launch_bounds__(256) void convert_queries_kernel(QueryPrep p) {
    const int lane = threadIdx.x & 63;
    const uint32_t slot = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (blockIdx.x == 0 && threadIdx.x < 4) p.stats[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 4) *p.fb_count = 0;
    if (slot >= p.n_slots) return;
    float ss = 0.0f, mx = 0.0f;
    uint32_t bad = 0;
    for (uint32_t i = lane; i < p.dim; i += 64) {
        float x = 0.0f;
        if (slot < p.nq) x = p.q[(size_t)slot * p.dim + i];
        if (!(__builtin_fabsf(x) <= 3.0e38f)) bad++;
        mx = fmaxf(mx, __builtin_fabsf(x));
        ss = __builtin_fmaf(x, x, ss);
        p.q_h[(size_t)slot * p.dim + i] = (_Float16)(x * MF_SCALE);
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
        bad += __shfl_xor(bad, off);
    }
    if (lane == 0) {
        p.qnorm[slot] = __builtin_sqrtf(ss) * 1.00001f;
        p.cand_cnt[slot] = 0;
        // unquantisable query (fp16 range) or non-finite: exact path decides
        p.fallback[slot] = (slot < p.nq && (bad || mx * MF_SCALE > 60000.0f)) ? 1u : 0u;
    }
}

// ---- threshold: k-th largest sampled block maximum per query -------------------------------------------
struct ThrArgs {
    const float *blockmax;   // [passes][J][256]
    uint32_t J, k, cap, nq;
    const float *qnorm;      // [n_slots]
    const uint32_t *fallback;
    float eps_rel_maxnorm;   // eps_rel * maxnorm
    float eps_abs_a;         // coefficient of (|q| + maxnorm)
    float maxnorm;
    float *thr;              // [n_slots] emit threshold
    float *eps;              // [n_slots]
};
__global__ __launch_bounds__(256) void threshold_kernel(ThrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *mins = keys + a.cap;
    uint64_t *thr = mins + 256;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + 1);
    const int tid = threadIdx.x;
    const uint32_t slot = blockIdx.x;            // pass*256 + q
    const uint32_t pass = slot / MF_BPAD, ql = slot % MF_BPAD;
    const float qn = a.qnorm[slot];
    const float eps = a.eps_rel_maxnorm * qn + a.eps_abs_a * (qn + a.maxnorm) + 1e-9f;
    if (slot >= a.nq || a.fallback[slot]) {
        if (tid == 0) { a.thr[slot] = __builtin_inff(); a.eps[slot] = eps; }   // never emits
        return;
    }
    TopKBuf buf{keys, cnt, thr, a.cap, a.k};
    auto key_at = [&](uint64_t j) -> uint64_t {
        return make_key(-a.blockmax[((size_t)pass * a.J + j) * MF_BPAD + ql], (uint32_t)j);
    };
    const uint32_t m = block_select_topk<256>(key_at, a.J, buf, mins);
    if (tid == 0) {
        float t = -__builtin_inff();
        if (m == a.k && a.k > 0) {
            const float kth = -order_key_inv((uint32_t)(buf.keys[a.k - 1] >> 32));
            // a positive k-th block max is backed by k LIVE rows (tombstoned rows score exactly 0)
            if (kth > 0.0f) t = kth - (2.001f * eps + 1e-7f * __builtin_fabsf(kth));
        }
        a.thr[slot] = t;   // -inf => emit everything => list overflow => exact fallback
        a.eps[slot] = eps;
    }
}

// ---- final stage: k-th best approximate score, 2-eps window, exact re-score, sort, write -----------------
struct FinalArgs {
    const float *rows;        // f32 master rows
    uint32_t dim;
    const float *q;           // [nq][dim] f32 queries
    uint32_t nq, k, cap;      // cap: TopKBuf capacity
    const uint64_t *cand;     // [n_slots][cand_cap]
    const uint32_t *cand_cnt;
    uint32_t cand_cap;
    const float *eps;         // [n_slots]
    uint32_t fcap;            // capacity of the re-score list
    uint32_t order;
    uint32_t id_base;
    uint32_t *fallback;       // [n_slots] in/out
    uint32_t *fb_list;        // [n_slots] compacted list of fallback queries
    uint32_t *fb_count;
    uint32_t *ids;            // [nq][k]
    float *dist;
    uint32_t *counts;
    uint32_t *stats;          // [0] emitted, [1] rescored, [2] overflowed queries
};

template <int ORDER>
__device__ __forceinline__ float exact_dot_row(const float *__restrict__ q_lds, const float *__restrict__ row, uint32_t dim) {
    if (ORDER == SHODH_ORDER_SCALAR4) {
        float sum = 0.0f;
        const uint32_t un = dim & ~3u;
        for (uint32_t i = 0; i < un; i += 4) {
            const float4 r = *reinterpret_cast<const float4 *>(row + i);
            const float4 w = *reinterpret_cast<const float4 *>(q_lds + i);
            float t = w.x * r.x;
            t = t + w.y * r.y;
            t = t + w.z * r.z;
            t = t + w.w * r.w;
            sum = sum + t;
        }
        for (uint32_t j = un; j < dim; ++j) sum = sum + q_lds[j] * row[j];
        return sum;
    } else {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t sn = dim & ~7u;
        for (uint32_t i = 0; i < sn; i += 8) {
            const float4 r0 = *reinterpret_cast<const float4 *>(row + i), r1 = *reinterpret_cast<const float4 *>(row + i + 4);
            const float4 w0 = *reinterpret_cast<const float4 *>(q_lds + i), w1 = *reinterpret_cast<const float4 *>(q_lds + i + 4);
            acc[0] = __builtin_fmaf(w0.x, r0.x, acc[0]); acc[1] = __builtin_fmaf(w0.y, r0.y, acc[1]);
            acc[2] = __builtin_fmaf(w0.z, r0.z, acc[2]); acc[3] = __builtin_fmaf(w0.w, r0.w, acc[3]);
            acc[4] = __builtin_fmaf(w1.x, r1.x, acc[4]); acc[5] = __builtin_fmaf(w1.y, r1.y, acc[5]);
            acc[6] = __builtin_fmaf(w1.z, r1.z, acc[6]); acc[7] = __builtin_fmaf(w1.w, r1.w, acc[7]);
        }
        float r = acc[0] + acc[1];
        r = r + acc[2]; r = r + acc[3]; r = r + acc[4]; r = r + acc[5]; r = r + acc[6]; r = r + acc[7];
        for (uint32_t j = sn; j < dim; ++j) r = r + q_lds[j] * row[j];
        return r;
    }
}

template <int ORDER>
__global__ __launch_bounds__(256) void final_stage_kernel(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = reinterpret_cast<float *>(smem);                          // [dim]
    uint64_t *keys = reinterpret_cast<uint64_t *>(qs + a.dim);            // [cap]
    uint64_t *mins = keys + a.cap;                                        // [256]
    uint64_t *ekeys = mins + 256;                                         // [fcap] exact keys of the window
    uint64_t *thr = ekeys + a.fcap;
    uint32_t *flist = reinterpret_cast<uint32_t *>(thr + 1);              // [fcap] rows of the window
    uint32_t *cnt = flist + a.fcap;
    uint32_t *fcnt = cnt + 1;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= a.nq) return;
    const uint32_t n = a.cand_cnt[q];
    bool bad = a.fallback[q] != 0 || n > a.cand_cap;
    if (!bad) {
        for (uint32_t i = tid; i < a.dim; i += 256) qs[i] = a.q[(size_t)q * a.dim + i];
        if (tid == 0) *fcnt = 0;
        TopKBuf buf{keys, cnt, thr, a.cap, a.k};
        const uint64_t *list = a.cand + (size_t)q * a.cand_cap;
        // pass A: k-th best approximate score
        auto key_a = [&](uint64_t i) -> uint64_t { return list[i]; };
        const uint32_t ma = block_select_topk<256>(key_a, n, buf, mins);
        // window: every candidate with s~ >= kth - 2 eps (all of them if fewer than k exist)
        float lo = -__builtin_inff();
        if (ma == a.k && a.k > 0) {
            const float kth = -order_key_inv((uint32_t)(buf.keys[a.k - 1] >> 32));
            lo = kth - (2.001f * a.eps[q] + 1e-7f * __builtin_fabsf(kth));
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) {
            const uint64_t key = list[i];
            const float s = -order_key_inv((uint32_t)(key >> 32));
            if (s >= lo) {
                const uint32_t slot = atomicAdd(fcnt, 1u);
                if (slot < a.fcap) flist[slot] = (uint32_t)key;
            }
        }
        __syncthreads();
        const uint32_t nf = *fcnt;
        if (nf > a.fcap) bad = true;       // block-uniform
        if (!bad) {
            // pass B: exact reference-order scores of the window, then top-k by (dist, id)
            for (uint32_t i = tid; i < nf; i += 256) {
                const uint32_t row = flist[i];
                const float dot = exact_dot_row<ORDER>(qs, a.rows + (size_t)row * a.dim, a.dim);
                ekeys[i] = make_key(-dot, a.id_base + row);
            }
            __syncthreads();
            auto key_b = [&](uint64_t i) -> uint64_t { return ekeys[i]; };
            const uint32_t m = block_select_topk<256>(key_b, nf, buf, mins);
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
            if (tid == 0) {
                a.counts[q] = m;
                atomicAdd(a.stats + 0, n);
                atomicAdd(a.stats + 1, nf);
            }
        }
    }
    if (bad && tid == 0) {
        a.fallback[q] = 1;
        const uint32_t s = atomicAdd(a.fb_count, 1u);
        a.fb_list[s] = q;
        atomicAdd(a.stats + 2, 1u);
    }
}

// ---- host side -----------------------------------------------------------------------------------------
struct MfmaPlan {
    uint32_t passes, n_slots, ksteps;
    uint32_t tile_stride, n_sel_tiles, J;
    uint64_t n_tiles;
    uint32_t cand_cap, fcap, topk_cap;
    int grid_x;
};

uint32_t topk_capacity(uint32_t k);   // flat_exact.hip

bool mfma_supported(uint32_t dim) { return dim == 128 || dim == 256 || dim == 384 || dim == 512; }

MfmaPlan mfma_plan(uint64_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int cus) {
    MfmaPlan p{};
    p.passes = (uint32_t)ceil_div(nq, MF_BPAD);
    p.n_slots = p.passes * MF_BPAD;
    p.ksteps = dim / 16;
    p.n_tiles = ceil_div(n_rows, MF_TR);
    // sample: every S-th tile (S = 16 unless SHODH_SAMPLE_STRIDE; measured: 32 -> 463, 16 -> 265, 8 -> 150 candidates/query at 1M), but at least 2k block maxima (4 per tile) and 64 tiles
    static const uint32_t sample_stride = getenv("SHODH_SAMPLE_STRIDE") ? (uint32_t)atoi(getenv("SHODH_SAMPLE_STRIDE")) : 16u;
    uint64_t want = p.n_tiles / (sample_stride ? sample_stride : 16u);
    const uint64_t min_tiles = (uint64_t)k / 2 + 64;
    if (want < min_tiles) want = min_tiles;
    if (want > p.n_tiles) want = p.n_tiles;
    p.tile_stride = (uint32_t)(p.n_tiles / want);
    if (p.tile_stride < 1) p.tile_stride = 1;
    p.n_sel_tiles = (uint32_t)ceil_div(p.n_tiles, p.tile_stride);
    p.J = p.n_sel_tiles * 4;
    uint32_t cc = 192u * (k ? k : 1);
    if (cc < 4096) cc = 4096;
    p.cand_cap = next_pow2(cc);
    uint32_t fc = 4u * (k ? k : 1);
    if (fc < 2048) fc = 2048;
    p.fcap = next_pow2(fc);
    p.topk_cap = topk_capacity(k);
    p.grid_x = cus;
    return p;
}

struct MfmaWorkspace {
    _Float16 *q_h; float *qnorm; float *thr; float *eps; uint32_t *cand_cnt; uint32_t *fallback; uint32_t *fb_list;
    uint32_t *fb_count; uint32_t *stats; float *blockmax; uint64_t *cand;
};

size_t mfma_workspace_bytes(const MfmaPlan &p, uint32_t dim, size_t *offs /*[11]*/) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    offs[0] = take((size_t)p.n_slots * dim * 2);         // q_h
    offs[1] = take((size_t)p.n_slots * 4);               // qnorm
    offs[2] = take((size_t)p.n_slots * 4);               // thr
    offs[3] = take((size_t)p.n_slots * 4);               // eps
    offs[4] = take((size_t)p.n_slots * 4);               // cand_cnt
    offs[5] = take((size_t)p.n_slots * 4);               // fallback
    offs[6] = take((size_t)p.n_slots * 4);               // fb_list
    offs[7] = take(256);                                 // fb_count
    offs[8] = take(256);                                 // stats
    offs[9] = take((size_t)p.passes * p.J * MF_BPAD * 4);  // blockmax
    offs[10] = take((size_t)p.n_slots * p.cand_cap * 8);   // cand
    return o;
}

template <int MODE>
static int launch_scan(const MfmaArgs &a, const MfmaPlan &p, uint32_t n_sel, hipStream_t st) {
    const size_t lds = 2ull * MF_TR * a.dim * 2 + (size_t)MF_EQ_CAP * 12 + 32;
    dim3 grid((uint32_t)p.grid_x, p.passes);
    if ((uint32_t)p.grid_x > n_sel) grid.x = n_sel ? n_sel : 1;
    static const int qb_env = getenv("SHODH_MFMA_QB") ? atoi(getenv("SHODH_MFMA_QB")) : 1;
#define SHODH_LAUNCH_KS(KS)                                                                                        \
    case KS:                                                                                                       \
        if (qb_env == 2) {                                                                                         \
            SHODH_TRY(ensure_dynamic_lds((const void *)mfma_scan_kernel<MODE, KS, 2>, lds));                        \
            hipLaunchKernelGGL((mfma_scan_kernel<MODE, KS, 2>), grid, dim3(256), lds, st, a);                       \
        } else {                                                                                                   \
            SHODH_TRY(ensure_dynamic_lds((const void *)mfma_scan_kernel<MODE, KS, 1>, lds));                        \
            hipLaunchKernelGGL((mfma_scan_kernel<MODE, KS, 1>), grid, dim3(512), lds, st, a);                       \
        }                                                                                                          \
        break;
    switch (p.ksteps) {
        SHODH_LAUNCH_KS(8)
        SHODH_LAUNCH_KS(16)
        SHODH_LAUNCH_KS(24)
        SHODH_LAUNCH_KS(32)
        default: set_error("MFMA scan: unsupported dim %u", a.dim); return SHODH_ERR_UNSUPPORTED;
    }
#undef SHODH_LAUNCH_KS
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

// Enqueues the whole pre-scan + re-score pipeline. Queries that could not be resolved here
// (list overflow, unusable threshold, unquantisable query) are left in ws.fb_list / ws.fb_count
// for the exact scan, which the caller enqueues right after (flat_exact.hip, device-side count).
int launch_mfma_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim,
                         const uint32_t *deleted, const float *d_q, uint32_t nq, uint32_t k, uint32_t order,
                         uint32_t id_base, float maxnorm, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs,
                         uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                         hipEvent_t ev_scan_done, hipEvent_t ev_select_done, hipEvent_t ev_emit0, hipEvent_t ev_emit1) {
    MfmaWorkspace w;
    w.q_h = (_Float16 *)(ws_base + offs[0]); w.qnorm = (float *)(ws_base + offs[1]); w.thr = (float *)(ws_base + offs[2]);
    w.eps = (float *)(ws_base + offs[3]); w.cand_cnt = (uint32_t *)(ws_base + offs[4]); w.fallback = (uint32_t *)(ws_base + offs[5]);
    w.fb_list = (uint32_t *)(ws_base + offs[6]); w.fb_count = (uint32_t *)(ws_base + offs[7]); w.stats = (uint32_t *)(ws_base + offs[8]);
    w.blockmax = (float *)(ws_base + offs[9]); w.cand = (uint64_t *)(ws_base + offs[10]);

    QueryPrep qp{d_q, nq, dim, p.n_slots, w.q_h, w.qnorm, w.cand_cnt, w.fallback, w.fb_count, w.stats};
    hipLaunchKernelGGL(convert_queries_kernel, dim3((p.n_slots * 64 + 255) / 256), dim3(256), 0, st, qp);
    SHODH_HIP_TRY(hipGetLastError());

    static const uint32_t ablate = getenv("SHODH_ABLATE") ? (uint32_t)atoi(getenv("SHODH_ABLATE")) : 0u;
    MfmaArgs a{rows_h, n_rows, dim, w.q_h, w.thr, deleted, w.cand, w.cand_cnt, p.cand_cap, w.blockmax, p.tile_stride, p.n_sel_tiles, 0u};
    SHODH_TRY(launch_scan<MF_MODE_BLOCKMAX>(a, p, p.n_sel_tiles, st));

    // eps (DESIGN.md "error bound"): fp16 rounding of both operands 2^-10 (1+2^-11), f32 accumulation
    // dim * 2^-23, reference rounding ~1e-5; absolute term for flushed/denormal fp16 after the 2^8 scale
    const float eps_rel = 9.7704e-4f + (float)dim * 1.1921e-7f * 1.01f + 1.0e-5f;
    const float eps_abs_a = 2.3842e-7f * __builtin_sqrtf((float)dim) * 1.01f;
    ThrArgs t{w.blockmax, p.J, k, p.topk_cap, nq, w.qnorm, w.fallback, eps_rel * maxnorm, eps_abs_a, maxnorm, w.thr, w.eps};
    const size_t tlds = (size_t)p.topk_cap * 8 + 256 * 8 + 8 + 4 + 16;
    SHODH_TRY(ensure_dynamic_lds((const void *)threshold_kernel, tlds));
    hipLaunchKernelGGL(threshold_kernel, dim3(p.n_slots), dim3(256), tlds, st, t);
    SHODH_HIP_TRY(hipGetLastError());

    a.tile_stride = 1;
    a.n_sel_tiles = (uint32_t)p.n_tiles;
    a.ablate = ablate;
    if (ev_emit0) SHODH_HIP_TRY(hipEventRecord(ev_emit0, st));
    SHODH_TRY(launch_scan<MF_MODE_EMIT>(a, p, (uint32_t)p.n_tiles, st));
    if (ev_emit1) SHODH_HIP_TRY(hipEventRecord(ev_emit1, st));
    if (ev_scan_done) SHODH_HIP_TRY(hipEventRecord(ev_scan_done, st));

    FinalArgs f{rows, dim, d_q, nq, k, p.topk_cap, w.cand, w.cand_cnt, p.cand_cap, w.eps, p.fcap, order, id_base,
                w.fallback, w.fb_list, w.fb_count, d_ids, d_dist, d_counts, w.stats};
    const size_t flds = (size_t)dim * 4 + (size_t)p.topk_cap * 8 + 256 * 8 + (size_t)p.fcap * 8 + 8 + (size_t)p.fcap * 4 + 8 + 16;
    if (order == SHODH_ORDER_AVX2) {
        SHODH_TRY(ensure_dynamic_lds((const void *)final_stage_kernel<SHODH_ORDER_AVX2>, flds));
        hipLaunchKernelGGL((final_stage_kernel<SHODH_ORDER_AVX2>), dim3(nq), dim3(256), flds, st, f);
    } else {
        SHODH_TRY(ensure_dynamic_lds((const void *)final_stage_kernel<SHODH_ORDER_SCALAR4>, flds));
        hipLaunchKernelGGL((final_stage_kernel<SHODH_ORDER_SCALAR4>), dim3(nq), dim3(256), flds, st, f);
    }
    SHODH_HIP_TRY(hipGetLastError());
    if (ev_select_done) SHODH_HIP_TRY(hipEventRecord(ev_select_done, st));
    return SHODH_OK;
}

int launch_convert_rows(const float *rows, uint64_t first, uint64_t n, uint32_t dim, _Float16 *rows_h, uint32_t *stats, hipStream_t st) {
    if (n == 0) return SHODH_OK;
    uint64_t blocks = ceil_div(n, 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(convert_rows_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, rows, first, n, dim, rows_h, stats);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int launch_shadow_set_row(const float *rows, _Float16 *rows_h, uint64_t row, uint32_t dim, int zero, hipStream_t st) {
    hipLaunchKernelGGL(shadow_set_row_kernel, dim3(1), dim3(128), 0, st, rows, rows_h, row, dim, zero);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int launch_shadow_restore_deleted(const float *rows, _Float16 *rows_h, const uint32_t *deleted, uint64_t n, uint32_t dim, hipStream_t st) {
    if (n == 0) return SHODH_OK;
    hipLaunchKernelGGL(shadow_restore_deleted_kernel, dim3((uint32_t)n), dim3(128), 0, st, rows, rows_h, deleted, n, dim);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

}  // namespace shodh

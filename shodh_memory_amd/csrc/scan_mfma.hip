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
// Thresholds come from a 1/16 strided sample of the tiles (same kernel, BLOCKMAX epilogue): the
// k-th largest of the per-tile maxima is a valid lower bound of the k-th best score.
#include <cstdlib>

#include <hip/hip_ext.h>

#include "common.h"
#include "topk.h"
#include "glds.h"

#pragma clang fp contract(off)

namespace shodh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

constexpr int MF_WQ_CAP = 120;    // LDS emit-queue entries per WAVE (8 private queues per workgroup); 128 until the dynamic threshold needed 3 KiB of LDS
constexpr int MF_EQ_CAP = 8 * MF_WQ_CAP;
#ifndef SHODH_MF_SLOTS
#define SHODH_MF_SLOTS 4
#endif
constexpr int MF_SLOTS = SHODH_MF_SLOTS;       // private candidate slots per (query, workgroup); the rest goes to the shared overflow list
constexpr int MF_TR = 64;        // corpus rows per tile
constexpr int MF_BPAD = 256;     // queries per pass
#ifndef SHODH_DYN_PERIOD
#define SHODH_DYN_PERIOD 4
#endif

// The tile hand-over of mfma_scan_kernel. 0: one s_barrier per tile (rounds 1 - 5). Round 6: per-wave progress words in LDS instead -- the barrier made the eight
// waves meet once per tile, and the older wave of every SIMD, which the arbiter favours, sat out ~2 200 of a tile's ~5 800 cycles waiting for its partner; a survivor block
// in ANY wave delayed all eight (k = 120: most of what its survivors cost). What the barrier guaranteed is two facts, and each wave now checks only the one it needs,
// where it needs it: (a) this tile's data has landed -- every wave, at the top of the tile, against the DMA waves' "landed" words; (b) the buffer the tile's DMA refills
// is no longer read -- only the DMA waves, and with 2 only in the MIDDLE of the tile (the DMA pieces are issued during the second MFMA chain), against everybody's
// "finished" words. Waves drift up to a tile apart; a value is only ever read after the fact it stands for (the writer's own counted wait / lgkmcnt(0) precedes the write).
// Same results (every parity test), emit kernel 211 -> 203 us at k = 10, 231 -> 216 at k = 120 (same box; 1 without the late DMA: 209 / 234; 3, a first look at the
// words a tile early: no change).
// 4 (the product): with three buffers a DMA wave posts "tile t + 1 has landed" in the MIDDLE of tile t, where it waits for the others anyway: everything it has in
// flight there is at least a tile old, so vmcnt(0) costs nothing, and nobody finds the words short at the top of the next tile (per tile and wave ~350 cycles of
// polling instead of ~800). Together with the two knobs below 203 -> 194 us at k = 10, 224 -> 213 at k = 120 (same box, tools/r6_flat_variants.sh).
// 5 (measured, not kept: 203 / 227): the first A fragments of tile t + 1 requested during the last steps of tile t -- the look at the landed words moves into the
// second chain and the waves wait there instead.
#ifndef SHODH_MF_POLL
#define SHODH_MF_POLL 4
#endif
constexpr bool MF_POLL = SHODH_MF_POLL != 0;
// Who issues the corpus DMA, and who wins the SIMD. Per-tile section timers (tools/r6_emit_phases.sh) show the younger wave of every SIMD as the pole: the arbiter
// serves the older one first, so the younger one's chains stretch (1 000 + 1 860 cycles against 800 + 800 + the DMA's 780), and the older one then sits ~1 200 cycles
// in the middle of its tile waiting for it. Measured, emit kernel us at k = 10 / 120 on one box: DMA by the older waves 196.8 / 214.8 (+ younger at priority 1: 199 /
// 221.8, at 3: 198.6 / 222.6); DMA by the YOUNGER waves 204.3 / 228; DMA by the younger waves AND those at priority 1: 193.8 / 213.4 -- the waves that carry the
// issue stalls of the DMA are the ones that may take the matrix pipe when they are ready. Small effects, each reproduced twice; the structure as a whole sits at
// ~53 % of the matrix pipe (MI355X_MICROARCH.md "Two waves per SIMD": moving work between the two waves of a SIMD is zero-sum).
#ifndef SHODH_MF_YPRIO
#define SHODH_MF_YPRIO 1       // issue priority (s_setprio) of the younger wave of every SIMD, waves 4 - 7; the older wave, which the arbiter favours at equal priority, stays at 0
#endif
#ifndef SHODH_MF_DMAY
#define SHODH_MF_DMAY 1        // 1: the corpus DMA is issued by the younger waves (4 - 7) instead of the older ones
#endif
constexpr bool MF_POLL_XPREF = SHODH_MF_POLL >= 5;     // 5: ... and a wave requests the first A fragments of tile t + 1 during the last steps of tile t (no cold start at the top of a tile)
constexpr bool MF_POLL_EARLY = SHODH_MF_POLL >= 4;     // 4: with three buffers a DMA wave posts "tile t + 1 has landed" in the MIDDLE of tile t (its pieces were issued a tile ago; nothing younger is in flight there), not at its end
constexpr bool MF_POLL_LATE = SHODH_MF_POLL >= 2;      // 2: the DMA of a tile is issued in its second half only, and the DMA waves look at the others' progress there, not at the top
constexpr int MF_DYN_NB = 16;    // levels of the threshold that tightens DURING the emit scan (see "dynamic threshold" below)
constexpr int MF_DYN_REP = 4;    // ... and replicas of every query's level counters (workgroup b adds to replica b % 4; the owner sums them): a query's counters are one hot line otherwise
constexpr float MF_SCALE = 256.0f;                 // rows and queries are stored as fp16(256 x)
constexpr float MF_INV_SCALE2 = 1.0f / 65536.0f;   // acc -> score

#ifdef SHODH_PROF
__device__ unsigned long long g_scores_end_ticks = 0;      // when the last workgroup of the score pre-scan finished (s_memrealtime)
#endif
enum { MF_MODE_BLOCKMAX = 0, MF_MODE_EMIT = 1, MF_MODE_SCORES = 2 };      // SCORES: every score of the pass is written out (small tables: IVF probe selection, see probe_select_kernel)

// build with SHODH_EXTRA_FLAGS=-DSHODH_PROF to print per-section wave cycles (s_memtime) of the scan kernel
#ifdef SHODH_PROF
#define PROF_DECL long long pt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq_ = clock64();
#define PROF_T(i) { const long long t_ = clock64(); pt_[i] += t_ - tq_; tq_ = t_; }
#define PROF_N(i) pt_[i] += 1;
#else
#define PROF_DECL
#define PROF_T(i)
#define PROF_N(i)
#endif

struct MfmaArgs {
    const _Float16 *rows_h;   // [n_rows][dim] fp16(256*x); tombstoned rows are zero
    uint64_t n_rows;
    uint32_t dim;
    const _Float16 *q_h;      // [passes][8 waves][dim/16 k-steps][64 lanes][8] fragment-major fp16(256*q), zero padded
    const float *thr;         // [passes][256] emit threshold on the true score scale
    const uint32_t *deleted;  // bitmask or nullptr
    uint64_t *slots;          // [passes][256][gridDim.x][MF_SLOTS] private (query, workgroup) candidate slots, KEY_NONE = empty
    uint64_t *cand;           // [passes][256][cand_cap]  shared overflow list, key = (order_key(-s~) << 32) | local row
    uint32_t *cand_cnt;       // [passes][256]
    uint32_t cand_cap;
    float *blockmax;          // [passes][J][256], J = n_sel_tiles (one maximum per sampled 64-row tile and query)
    uint32_t tile_stride;     // tile index = sel * tile_stride
    uint32_t n_sel_tiles;
    uint32_t ablate;          // diagnostics only (SHODH_ABLATE=8: thresholds forced to +inf, nothing is emitted; results invalid)
    uint32_t nq;              // real queries of the call (the last pass may be partly padding)
    // dynamic threshold (emit scan, dim <= 384; dcnt == nullptr: off). The sampled bound is the k-th best of a 1/S sample: its rank among all rows is ~ S k, so
    // ~ S k rows per query survive it (1 255 at recall's k = 120 -- and the emit scan's time follows the survivor count: 168 us with none, 203 at 265 per
    // query, 240 at 1 255). Every survivor is a ROW WITH A KNOWN SCORE, though: once k live rows with approximate score >= E have been seen -- by all
    // workgroups together -- the k-th best is >= E and the bound may rise to E for the rest of the scan.
    //   levels: equidistant on the order-key scale, level j <=> key <= K_B - j * step (dpar[slot] = {K_B, floor(2^32 / step)}, K_B = key of the sampled bound);
    //   dcnt[slot][j-1] = rows seen whose HIGHEST level is j (one device-scope atomic per survivor that reaches a level its query has not passed yet);
    //   dthr[slot][j-1] = the emit threshold level j allows (same 2 eps margin as the sampled one; +inf = level not offered);
    //   dpub[slot]      = the published bound of the query: (threshold bits & ~31) | level, 0 = none yet.
    // Traffic is what this has to avoid (a first form in which every wave polled its 32 queries' counters every tile -- 65 k scattered coherent loads per tile
    // time over all workgroups -- doubled the kernel's time): query ql of a pass is OWNED by workgroup ql % gridDim.x, whose wave 6 reads that ONE query's 16
    // counters per tile (one 64-byte line) and publishes the highest level that has reached k; wave 7 of every workgroup reads the 1 KiB of published words
    // of the pass per tile (8 lines) into LDS, where every wave picks up its queries' bounds. Both loads are issued at the top of a tile and consumed at its
    // end (waves 6 and 7 issue no DMA). Exactness: a count is only ever a count of distinct live rows that were really seen, so "count >= k" is a fact
    // whoever reads it and whenever; a late or missed update only leaves the bound lower.
    uint32_t *dcnt;           // [passes*256][MF_DYN_REP][MF_DYN_NB]
    const float *dthr;        // [passes*256][MF_DYN_NB] scaled by 2^16 like the accumulators
    uint32_t *dpub;           // [passes*256]
    const uint32_t *dpar;     // [passes*256] x {K_B, step}
    uint32_t k;
    // MF_MODE_SCORES: scores[(pass * 256 + query) * scores_ld + row] = s~ (true score scale), rows of whole tiles (scores_ld = tiles * 64; rows >= n_rows: whatever the
    // slab's padding holds, the reader ignores them)
    float *scores;
    uint32_t scores_ld;
};

template <int KSTEPS>
__host__ __device__ constexpr int mfma_scan_nbuf() { return (3 * MF_TR * KSTEPS * 32 + MF_EQ_CAP * 12 + MF_BPAD * 4 + MF_BPAD * 12 + 256 + 64 + 64 <= 160 * 1024) ? 3 : 2; }

// 8 waves per workgroup (2 per SIMD), wave w owns queries [32w, 32w+32) as resident B fragments (dim/16 x 4 VGPRs).
// Every wave runs ONE software-pipelined instruction stream per 64-row tile: 2*KSTEPS MFMAs (row block 0 into acc0 for all
// k-steps, then row block 1 into acc1), and in their shadows (the matrix pipe is busy 32 cycles per MFMA; the wave
// is free to issue other work meanwhile)
//   - one ds_read_b128 per MFMA: the A fragment D steps ahead;
//   - the running maximum of the accumulator that is NOT being written (acc1 of the previous tile during the acc0
//     chain, acc0 during the acc1 chain), one v_max per MFMA; only when a lane's maximum reaches its threshold --
//     rare -- are the 16 values walked;
//   - the LDS-DMA (global_load_lds_dwordx4) of this thread's share of the tile NBUF-1 ahead: no staging registers,
//     no ds_write pass, and two tile periods for the data to land (NBUF = 3 where the LDS allows it).
// The LDS image of a tile is row-major with the 16-B chunks of a row XOR-swizzled by row&15 (conflict-free
// ds_read_b128 for the 32x32x16 A fragment). LDS-DMA writes lane-linear, so the swizzle is applied to the SOURCE
// address: the lane that owns LDS chunk (row, p) fetches global chunk (row, p ^ (row&15)) -- a permutation inside one
// 256-B segment, still whole cache lines per wave.
// Rounds 1 - 5: one s_barrier per tile handed the buffers over (raw barrier + lgkmcnt(0): a __syncthreads() would not wait for
// the DMA anyway, the counted vmcnt before it did). Round 6: progress words in LDS instead (SHODH_MF_POLL above).
// History, cycles per tile at 1M x 384, 256 queries (s_memtime; matrix-pipe floor 2 waves x 48 x 32 = 3072):
//   register staging, all phases in sequence, waves in lock-step: 5190; two wave groups half a tile out of phase
//   (one group multiplies while the other stages): 5390 including the profiling reads -- the two groups' MFMA phases
//   ended up serialised by the two barriers per tile.
template <int MODE, int KSTEPS>
__global__ __launch_bounds__(512, 2) void mfma_scan_kernel(MfmaArgs a) {
    constexpr int NT = 512;
    constexpr int DIM = KSTEPS * 16;
    constexpr int CPR = KSTEPS * 2;          // 16-B chunks per row
    constexpr int PITCH = DIM * 2;           // bytes per row in LDS
    constexpr int CPT = MF_TR * CPR / NT;    // chunks per thread per tile
    constexpr int TILE_BYTES = MF_TR * PITCH;
    constexpr int NS = 2 * KSTEPS;           // MFMAs per wave and tile (step s: row block s / KSTEPS, k-step s % KSTEPS)
    constexpr int D = 6;                     // A fragments in flight (10 measured the same)
    constexpr int RING = 8;
    constexpr int NBUF = mfma_scan_nbuf<KSTEPS>();
    constexpr int PF = NBUF - 1;             // tiles the DMA runs ahead
    static_assert(4 * 2 * CPT <= NS, "DMA issue slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: [NBUF tiles][8 wave-private emit queues: MF_WQ_CAP x (key u64, query u32)][per-query slot counters]
    uint64_t *eq_key = reinterpret_cast<uint64_t *>(smem + NBUF * TILE_BYTES);
    uint32_t *eq_q = reinterpret_cast<uint32_t *>(eq_key + MF_EQ_CAP);
    uint32_t *qcount = eq_q + MF_EQ_CAP;     // [MF_BPAD] entries this workgroup has written per query
    constexpr bool DYN = MODE == MF_MODE_EMIT && KSTEPS <= 24;      // (dim 512 has no registers to spare: 256 with spills as it is)
    constexpr uint32_t DYN_P = SHODH_DYN_PERIOD;                    // tiles per round of the dynamic threshold (a power of two)
    uint32_t *dpar_l = qcount + MF_BPAD;     // [MF_BPAD][2] dynamic threshold: {K_B, step} of every query of the pass (kept for diagnostics; the lanes hold their own)
    uint32_t *pub_l = dpar_l + 2 * MF_BPAD;  // [MF_BPAD] ... and its published bound (MfmaArgs::dpub), refreshed by wave 7 once per tile (LDS-DMA)
    uint32_t *own_l = pub_l + MF_BPAD;       // [64] the counters of the queries this workgroup owns, refreshed by wave 6 once per tile (LDS-DMA)
    uint32_t *sync_l = own_l + 64;           // [16] MF_POLL: [0 .. 3] tiles whose data waves 0-3 have seen land, [8 .. 15] tiles waves 0-7 have finished reading
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool dma_wave = SHODH_MF_DMAY ? wave >= 4 : wave < 4;      // the four waves that issue the corpus DMA (wave & 3: which quarter of every piece)
    if (SHODH_MF_YPRIO && wave >= 4) __builtin_amdgcn_s_setprio(SHODH_MF_YPRIO);
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const uint32_t pass = blockIdx.y;
    const uint32_t q_local = wave * 32 + l31;      // this lane's query within the pass
    // a wave whose 32 queries are all padding (small batches: a single query leaves seven of the eight) multiplies nothing:
    // it only keeps its share of the DMA and the barriers, so a small batch costs the bytes, not the MFMA work of a full one
#ifdef SHODH_DIAG
    const bool active = (uint32_t)(pass * MF_BPAD + wave * 32) < a.nq && !(a.ablate & 32u);      // (diagnostics, bit 32: the stream alone -- DMA, waits, barriers -- no multiplication)
#else
    const bool active = (uint32_t)(pass * MF_BPAD + wave * 32) < a.nq;      // wave-uniform
#endif
    if (MODE == MF_MODE_EMIT && tid < MF_BPAD) {
        qcount[tid] = 0;
        // this workgroup's private candidate slots of query tid start out empty (nobody else writes them)
        uint64_t *sl = a.slots + (((size_t)blockIdx.y * MF_BPAD + tid) * gridDim.x + blockIdx.x) * MF_SLOTS;
#pragma unroll
        for (int j = 0; j < MF_SLOTS; ++j) sl[j] = KEY_NONE;
    }
    const bool dyn_on = DYN && a.dcnt != nullptr;        // uniform
    if (DYN && tid < MF_BPAD) {
        dpar_l[2 * tid] = dyn_on ? a.dpar[((size_t)pass * MF_BPAD + tid) * 2] : 0u;        // K_B = 0: no key is <= it, nothing is ever counted
        dpar_l[2 * tid + 1] = dyn_on ? a.dpar[((size_t)pass * MF_BPAD + tid) * 2 + 1] : 0u;
        pub_l[tid] = 0u;
        if (tid < 64) own_l[tid] = 0u;
    }
#ifdef SHODH_PROF
    const long long wc0_ = wall_clock64();
#endif

    // source byte offsets inside a tile for this thread's LDS chunks p = i*NT + tid (see the swizzle note above).
    // No row clamp: the shadow slab is allocated in multiples of 64 rows, rows >= n_rows are never reported.
    // The DMA is issued by waves 0-3 only (2*CPT pieces of 256 x 16 B each per tile): the SIMD arbiter favours the
    // older wave of a SIMD, so waves 0-3 reach the end of a tile long before their partners 4-7 and have the slack to
    // absorb the ~100-150 cycles an LDS-DMA instruction stalls its wave at issue; on waves 4-7 the same stalls were
    // fully exposed (measured per tile: waves 0-3 2800 cycles busy + 2000 waiting at the barrier, waves 4-7 4600 busy).
    constexpr int NPC = 2 * CPT;             // DMA pieces per issuing thread and tile
    uint32_t srcoff[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int p = i * 256 + (tid & 255);
        const int row = p / CPR, slot = p % CPR;
        const int c = (slot & ~15) | ((slot & 15) ^ (row & 15));
        srcoff[i] = (uint32_t)(row * PITCH + c * 16);
    }
    const size_t tile_bytes_g = (size_t)a.tile_stride * MF_TR * DIM * 2;
    const unsigned char *rows_b = reinterpret_cast<const unsigned char *>(a.rows_h);
    const uint32_t wave_lds = smem_lds + (uint32_t)(wave & 3) * 1024u;     // this wave's 64 x 16 B window inside each 256-chunk slab

    uint32_t sel = blockIdx.x;
    const uint32_t step = gridDim.x;
    // prologue DMA: tiles 0 .. PF-1 of this workgroup (clamped to a valid tile: a harmless extra read, no branch)
#pragma unroll
    for (int b = 0; b < PF; ++b) {
        const uint32_t tsel = sel + b * step < a.n_sel_tiles ? sel + b * step : (sel < a.n_sel_tiles ? sel : 0u);
        const unsigned char *src = uniform_ptr(rows_b + (size_t)tsel * tile_bytes_g);
        if (dma_wave) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) glds16(src, srcoff[i], wave_lds + b * TILE_BYTES + i * 4096);
        }
    }

    // resident B fragments: query q_local, k = ks*16 + hi*8 .. +8
    half8 bq[KSTEPS];
    {
        const half8 *qp = reinterpret_cast<const half8 *>(a.q_h) + ((size_t)pass * 8 + wave) * KSTEPS * 64 + lane;   // fragment-major (convert_queries_kernel)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) bq[ks] = qp[ks * 64];
    }
    float thr_l = (MODE == MF_MODE_EMIT) ? a.thr[(size_t)pass * MF_BPAD + q_local] * (MF_SCALE * MF_SCALE) : 0.0f;
    if (a.ablate & 8u) thr_l = __builtin_inff();   // diagnostics: nothing is emitted
    // dynamic threshold, the owner's side (wave 6): lane group g = lane / 16 looks after owned query blockIdx.x + g * gridDim.x, lane j = lane % 16 after its
    // level j + 1: that level's threshold (read once) and, per tile, its counter. The readers' side (wave 7): four published words per lane.
    const uint32_t own_q = blockIdx.x;                                                     // query of the pass this workgroup owns (queries >= gridDim.x have no owner: small corpora)
    const bool own_ok = dyn_on && own_q < (uint32_t)MF_BPAD;
    const uint32_t own_off = (uint32_t)((((size_t)pass * MF_BPAD + (own_ok ? own_q : 0u)) * MF_DYN_REP * MF_DYN_NB + lane) * 4);      // lane = replica * 16 + (level - 1)
    float own_thr = __builtin_inff();
    if (own_ok) own_thr = a.dthr[((size_t)pass * MF_BPAD + own_q) * MF_DYN_NB + (lane & 15)];
    const unsigned char *dcnt_base = uniform_ptr(reinterpret_cast<const unsigned char *>(a.dcnt));
    const unsigned char *dpub_base = uniform_ptr(reinterpret_cast<const unsigned char *>(a.dpub + (size_t)pass * MF_BPAD));
    const uint32_t own_lds = smem_lds + (uint32_t)((unsigned char *)own_l - smem), pub_lds = smem_lds + (uint32_t)((unsigned char *)pub_l - smem);

    // LDS byte offsets of this lane's A fragments. chunk c = 2*ks + hi; swizzled chunk = c ^ (row&15)
    // = (c & ~15) | ((c & 15) ^ sw): only 8 distinct low parts per lane, the rest is an immediate.
    const int sw = l31 & 15;
    int aoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aoff[j] = l31 * PITCH + (((2 * j + hi) ^ sw) << 4);

    // ---- survivors -------------------------------------------------------------------------------------
    // Each wave appends to its OWN LDS queue; the fill count is a wave-uniform register, slots come from a
    // ballot + mbcnt (no atomic, nothing to wait for). The queue is drained by the wave itself (no workgroup
    // barrier): entry -> slot s = qcount[q]++ (LDS atomic) of this workgroup's private MF_SLOTS slots of query q,
    // a plain store nobody else writes to; only s >= MF_SLOTS takes the shared overflow list and its global atomic.
    // (With one shared list per query all 256 workgroups drained at the end of the launch into the same 256
    // counters: 14 us of serialised device-scope atomics.)
    uint64_t *wq_key = eq_key + wave * MF_WQ_CAP;
    uint32_t *wq_q = eq_q + wave * MF_WQ_CAP;
    uint32_t wq_n = 0;                       // wave-uniform
    uint64_t *my_slots = a.slots + ((size_t)pass * MF_BPAD * gridDim.x + blockIdx.x) * MF_SLOTS;   // + q * gridDim.x * MF_SLOTS
#ifdef SHODH_PROF
    long long ep_[6] = {0, 0, 0, 0, 0, 0};      // emit_block: entries, cycles, survivors; drain: calls, cycles, entries
#endif
    auto drain = [&]() {
#ifdef SHODH_PROF
        const long long d0_ = clock64(); ep_[3] += 1; ep_[5] += wq_n;
#endif
        const uint32_t n = wq_n < (uint32_t)MF_WQ_CAP ? wq_n : (uint32_t)MF_WQ_CAP;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint64_t key = wq_key[i];
            const uint32_t row = (uint32_t)key, k32 = (uint32_t)(key >> 32);
            // tombstoned (vamana.rs:1175-1177). A tombstoned row is all zeros in the shadow copy, so its score is exactly +0: only a survivor whose score is
            // not positive (key >= order_key(-0.0)) can be one, and only for it is the bit fetched (a load the wave would have to wait for on every drain)
            if (a.deleted && k32 >= 0x7FFFFFFFu && ((a.deleted[row >> 5] >> (row & 31)) & 1u)) continue;
            const uint32_t ql = wq_q[i];
            if (DYN && dyn_on) {
                // dynamic threshold: a survivor that reaches a level its query has not passed yet is counted, ONCE, in the counter of the highest level it
                // reaches (the counters are a histogram, the owner sums it from the top). Here and not where the survivor is found: a memory instruction
                // issued from the scan loop stalls its wave for ~150 cycles behind the corpus DMA, and at the tile barrier all eight waves pay for it --
                // counting each survivor as it was found cost more than the survivors it saved. One atomic INSTRUCTION per drain covers up to 64 entries.
                const uint32_t kb = dpar_l[2 * ql];
                if (k32 <= kb) {
                    // levels this score reaches: (K_B - k32) / step through the multiplier floor(2^32 / step) -- the quotient may come out one short (a row
                    // a hair above an edge counts one level lower: short is safe), never long
                    const uint32_t lvl = pub_l[ql] & 31u;
                    uint32_t j = __umulhi(kb - k32, dpar_l[2 * ql + 1]);
                    if (j > (uint32_t)MF_DYN_NB) j = MF_DYN_NB;
                    if (j > lvl)
                        __hip_atomic_fetch_add(a.dcnt + (((size_t)pass * MF_BPAD + ql) * MF_DYN_REP + (blockIdx.x & (MF_DYN_REP - 1))) * MF_DYN_NB + (j - 1u), 1u,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const uint32_t s_ = atomicAdd(qcount + ql, 1u);
            if (s_ < (uint32_t)MF_SLOTS) {
                my_slots[(size_t)ql * gridDim.x * MF_SLOTS + s_] = key;
            } else {
                const size_t qi = (size_t)pass * MF_BPAD + ql;
                const uint32_t slot = atomicAdd(a.cand_cnt + qi, 1u);
                if (slot < a.cand_cap) a.cand[qi * a.cand_cap + slot] = key;
            }
        }
        wq_n = 0;
#ifdef SHODH_PROF
        ep_[4] += clock64() - d0_;
#endif
    };
    // the survivors of one 32-row block (entered BY THE WHOLE WAVE when some lane's maximum reached its threshold: the
    // queue bookkeeping below is wave-uniform). C layout (32x32):
    // col = lane&31 (query), value r of a lane is row (r&3) + 8*(r>>2) + 4*(lane>>5) of the block.
    // one survivor straight to its candidate slot (the body of drain() for the calling lane)
    auto emit_direct = [&](uint64_t key) {
        const uint32_t row = (uint32_t)key;
        if (a.deleted && ((a.deleted[row >> 5] >> (row & 31)) & 1u)) return;
        const uint32_t s_ = atomicAdd(qcount + q_local, 1u);
        if (s_ < (uint32_t)MF_SLOTS) {
            my_slots[(size_t)q_local * gridDim.x * MF_SLOTS + s_] = key;
        } else {
            const size_t qi = (size_t)pass * MF_BPAD + q_local;
            const uint32_t slot = atomicAdd(a.cand_cnt + qi, 1u);
            if (slot < a.cand_cap) a.cand[qi * a.cand_cap + slot] = key;
        }
    };
    auto emit_block = [&](const floatx16 &c, uint64_t brow0 /* first row of the block */) {
        const uint64_t left = a.n_rows > brow0 + 4 * hi ? a.n_rows - (brow0 + 4 * hi) : 0;
        const uint32_t lim = left < 64 ? (uint32_t)left : 64u;      // this lane's values with row offset < lim exist
        const uint32_t wq_n0 = wq_n;
#ifdef SHODH_PROF
        const long long e0_ = clock64(); ep_[0] += 1;
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t roff = (r & 3) + 8 * (r >> 2);
            const bool hit = c[r] >= thr_l && roff < lim;
            const uint64_t b = __builtin_amdgcn_ballot_w64(hit);
            if (__builtin_expect(b != 0, 0)) {
                const uint32_t slot = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (hit && slot < (uint32_t)MF_WQ_CAP) {
                    const uint64_t key_ = make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)(brow0 + 4 * hi + roff));
                    wq_key[slot] = key_;
                    wq_q[slot] = q_local;
                }
                wq_n = __builtin_amdgcn_readfirstlane(wq_n + (uint32_t)__builtin_popcountll(b));
            }
        }
        if (__builtin_expect(wq_n > (uint32_t)MF_WQ_CAP, 0)) {
            // A dense block (a corpus inside a narrow cone: most lanes hit on most values) holds more survivors than the queue
            // has room for. Its queue entries are discarded and every hit of the block goes straight to its candidate slot
            // (round 1 dropped them and sent every query of the pass to the exact scan of the corpus).
            wq_n = wq_n0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t roff = (r & 3) + 8 * (r >> 2);
                if (c[r] >= thr_l && roff < lim) emit_direct(make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)(brow0 + 4 * hi + roff)));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // these stores / atomics share the VM counter with the DMA (see the loop)
        }
#ifdef SHODH_PROF
        ep_[1] += clock64() - e0_; ep_[2] += wq_n - wq_n0;
#endif
    };

    if (MF_POLL && tid < 16) sync_l[tid] = tid < 8 ? (uint32_t)PF : 0u;      // the prologue's PF tiles have landed (waited for below); nobody has finished a tile
    // the pipeline is primed: everything issued so far (DMA, query fragments, thresholds) lands before the stream starts
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): hipcc's own loads, so that it does not re-wait for them inside the loop
    __syncthreads();

    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    const floatx16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool have_prev = false;                 // acc1 of the previous tile waits for its epilogue
    uint32_t prev_sel = 0;
    uint32_t cur = 0;                       // LDS buffer of the current tile
    PROF_DECL
#ifdef SHODH_PROF
    const long long wc1_ = wall_clock64();
#endif
    uint32_t tile_no = 0;                   // tiles this workgroup has done (uniform)
    constexpr bool XP = MF_POLL_XPREF && PF == 2 && KSTEPS >= 8;      // the first D fragments of a tile are requested during the tile before it
    half8 ring[RING];
    auto rd = [&](const unsigned char *buf_, int st) {
#ifdef SHODH_HALF_LDS      // (diagnostic builds, results invalid: every second A fragment is not read -- is the LDS port what the chains wait for?)
        if (st & 1) return;
#endif
        const int rb = st / KSTEPS, ks = st % KSTEPS;
        ring[st % RING] = *reinterpret_cast<const half8 *>(buf_ + rb * 32 * PITCH + aoff[ks & 7] + (ks >> 3) * 256);
    };
    auto wait_landed = [&](uint32_t upto) {      // every DMA wave has seen its pieces of the tiles < upto land
        const uint32_t need = (lane < 4) ? upto : 0u;
        const uint32_t paddr = smem_lds + (uint32_t)((unsigned char *)sync_l - smem) + (uint32_t)(lane & 15) * 4u;
        for (;;) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(paddr) : "memory");
            if (__builtin_amdgcn_ballot_w64(v >= need) == ~0ull) break;
            __builtin_amdgcn_s_sleep(2);
        }
    };
    if (XP && active) {
#pragma unroll
        for (int st = 0; st < D; ++st) rd(smem, st);
    }
    for (; sel < a.n_sel_tiles; sel += step, ++tile_no) {
        const unsigned char *buf = smem + cur * TILE_BYTES;
        const unsigned char *nbuf = smem + (cur + 1 == NBUF ? 0 : cur + 1) * TILE_BYTES;
        const uint32_t pfb = cur + PF >= NBUF ? cur + PF - NBUF : cur + PF;      // buffer the DMA fills during this tile
        const uint32_t psel = sel + PF * step < a.n_sel_tiles ? sel + PF * step : sel;
        const unsigned char *psrc = uniform_ptr(rows_b + (size_t)psel * tile_bytes_g);
        const uint32_t pdst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + pfb * TILE_BYTES));

      if (MF_POLL && !XP) {
        // Hand-over without a workgroup barrier (see the end of the tile): this tile's data must have landed (every DMA wave has seen its pieces of it land:
        // sync_l[0 .. 3] > tile_no), and a DMA wave may refill the buffer of the tile before this one only when all eight waves are through with it
        // (sync_l[8 .. 15] >= tile_no). One read per poll: lane l looks at word l. (XP: looked at during the tile before, where its first fragments are requested.)
        const uint32_t need = (lane < 4) ? tile_no + 1u : ((!MF_POLL_LATE && lane >= 8 && lane < 16 && dma_wave) ? tile_no : 0u);
        const uint32_t paddr = smem_lds + (uint32_t)((unsigned char *)sync_l - smem) + (uint32_t)(lane & 15) * 4u;
        for (;;) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(paddr) : "memory");
            if (__builtin_amdgcn_ballot_w64(v >= need) == ~0ull) break;
            __builtin_amdgcn_s_sleep(2);
        }
        PROF_T(8)
      }
      auto wait_others = [&]() {      // (MF_POLL_LATE, DMA waves) every wave has finished the tile before this one
        const uint32_t paddr2 = smem_lds + (uint32_t)((unsigned char *)sync_l - smem) + (8u + (uint32_t)(lane & 7)) * 4u;
        for (;;) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(paddr2) : "memory");
            if (__builtin_amdgcn_ballot_w64(v >= tile_no) == ~0ull) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (MF_POLL_EARLY && PF == 2) {
            // everything this wave has in flight here is older than the pieces it is about to issue: the pieces of tile t + 1 (issued during tile t - 1) and the
            // stores of its last drain. Once they are in, tiles <= t + 1 have landed as far as this wave's share goes: nobody waits at the top of the next tile.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const uint32_t ld_ = tile_no + 2u;
                asm volatile("ds_write_b32 %0, %1" ::"v"(smem_lds + (uint32_t)((unsigned char *)sync_l - smem) + (uint32_t)(wave & 3) * 4u), "v"(ld_) : "memory");
            }
        }
      };
      if (DYN && dyn_on) {
        // dynamic threshold: the owner's counters (wave 6) and the published bounds of the pass (wave 7) by device-coherent LDS-DMA -- no destination
        // registers for the compiler to move around while the data is in flight -- issued now, waited for at the end of the tile (neither wave issues the
        // corpus DMA: a plain vmcnt(0) there waits for nothing else)
        // Once per DYN_P tiles each (the counters only move when the queues are drained, every DYN_P tiles too; a memory instruction issued from this loop
        // stalls its wave for a few hundred cycles): drain at the end of phase DYN_P - 1, owner's look at phase 1, readers' at phase 2.
        if (wave == 6) { if ((tile_no & (DYN_P - 1u)) == (1u & (DYN_P - 1u))) glds4_coherent(dcnt_base, own_off, own_lds); }
        else if (wave == 7) { if ((tile_no & (DYN_P - 1u)) == (2u & (DYN_P - 1u))) glds16_coherent(dpub_base, (uint32_t)lane * 16u, pub_lds); }
      }
      if (active) {
        if (DYN && dyn_on) {       // this query's published bound, if any (pub_l: landed during the previous tile, before its barrier)
            const uint32_t w_ = pub_l[q_local];
            if (w_) thr_l = fmaxf(thr_l, __uint_as_float(w_ & ~31u));
        }
        if (!XP) {
#pragma unroll
            for (int st = 0; st < D; ++st) rd(buf, st);
        }
        PROF_T(0)
        constexpr int PER1 = (16 + KSTEPS - 1) / KSTEPS;          // accumulator values folded into the maximum per step
        constexpr int PER0 = (16 + KSTEPS - 3) / (KSTEPS - 2);
        const uint64_t row0 = (uint64_t)sel * a.tile_stride * MF_TR;
        const uint64_t prow1 = (uint64_t)prev_sel * a.tile_stride * MF_TR + 32;
        float m0 = -__builtin_inff(), m1 = -__builtin_inff();
        // row block 0 -> acc0; in the shadows: maximum of the previous tile's acc1, DMA issue
#pragma unroll
        for (int st = 0; st < KSTEPS; ++st) {
            rd(buf, st + D);
#ifdef SHODH_HALF_LDS
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[(st & ~1) % RING], bq[st], st == 0 ? zero16 : acc0, 0, 0, 0);
#elif defined(SHODH_HALF_MFMA)      // (diagnostic builds, results invalid: every second multiplication left out, its fragment still read)
            if (st & 1) asm volatile("" ::"v"(ring[st % RING]));
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq[st], st == 0 ? zero16 : acc0, 0, 0, 0);
#else
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq[st], st == 0 ? zero16 : acc0, 0, 0, 0);
#endif
            if (MODE == MF_MODE_EMIT) {
#pragma unroll
                for (int u = 0; u < PER1; ++u) if (st * PER1 + u < 16) m1 = fmaxf(m1, acc1[st * PER1 + u]);
            }
            if (!MF_POLL_LATE && (st & 3) == 2 && (st >> 2) < NPC) {
                if (dma_wave) glds16(psrc, srcoff[st >> 2], pdst + (st >> 2) * 4096);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PROF_T(1)
        if (MODE == MF_MODE_EMIT) {
            if (have_prev && __builtin_amdgcn_ballot_w64(m1 >= thr_l) != 0) emit_block(acc1, prow1);   // wave-uniform entry
        }
        PROF_T(2)
        if (MF_POLL_LATE && dma_wave) { wait_others(); PROF_T(9) }      // the buffer this tile's DMA refills: everybody through with the tile before this one?
        // row block 1 -> acc1; in the shadows: maximum of acc0 (from two steps in: its last MFMA has to retire first)
        uint32_t fl_ = 0;
#pragma unroll
        for (int st = KSTEPS; st < NS; ++st) {
            if (XP && st == (NS - D - 4 > KSTEPS ? NS - D - 4 : KSTEPS)) fl_ = *reinterpret_cast<volatile uint32_t *>(sync_l + (lane & 3));      // the landed words, on their way while three more steps run
            if (XP && st == NS - D - 1) {
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(fl_ >= tile_no + 2u) != ~0ull, 0)) wait_landed(tile_no + 2u);      // (rare: a DMA wave half a tile behind)
            }
            if (st + D < NS) rd(buf, st + D);
            else if (XP) rd(nbuf, st + D - NS);
#ifdef SHODH_HALF_LDS
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[(st & ~1) % RING], bq[st - KSTEPS], st == KSTEPS ? zero16 : acc1, 0, 0, 0);
#elif defined(SHODH_HALF_MFMA)
            if (st & 1) asm volatile("" ::"v"(ring[st % RING]));
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq[st - KSTEPS], st == KSTEPS ? zero16 : acc1, 0, 0, 0);
#else
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq[st - KSTEPS], st == KSTEPS ? zero16 : acc1, 0, 0, 0);
#endif
            if (MODE == MF_MODE_EMIT && st >= KSTEPS + 2) {
#pragma unroll
                for (int u = 0; u < PER0; ++u) if ((st - KSTEPS - 2) * PER0 + u < 16) m0 = fmaxf(m0, acc0[(st - KSTEPS - 2) * PER0 + u]);
            }
            if (MF_POLL_LATE ? ((st & 1) == 0 && ((st - KSTEPS) >> 1) < NPC) : ((st & 3) == 2 && (st >> 2) < NPC)) {
                constexpr int dummy_ = 0; (void)dummy_;
                if (dma_wave) { const int pi = MF_POLL_LATE ? ((st - KSTEPS) >> 1) : (st >> 2); glds16(psrc, srcoff[pi], pdst + pi * 4096); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        PROF_T(3)
        if (MODE == MF_MODE_EMIT) {
            if (__builtin_amdgcn_ballot_w64(m0 >= thr_l) != 0) emit_block(acc0, row0);
        }
        PROF_T(3)
        if (MODE == MF_MODE_EMIT) {
            have_prev = true;
            prev_sel = sel;
            // (the queue is drained below, after the dynamic threshold's loads have been waited for)
        } else if (MODE == MF_MODE_SCORES) {
            // value r of a lane is row (r & 3) + 8 (r >> 2) + 4 hi of its block: four consecutive values are four consecutive rows -- one 16-byte store, and the two
            // half-waves of a query fill 32 contiguous bytes per instruction
            float *dst = a.scores + ((size_t)pass * MF_BPAD + q_local) * a.scores_ld + (size_t)sel * a.tile_stride * MF_TR + 4 * hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4s v0, v1;
#pragma unroll
                for (int u = 0; u < 4; ++u) { v0[u] = acc0[4 * j + u] * MF_INV_SCALE2; v1[u] = acc1[4 * j + u] * MF_INV_SCALE2; }
                *reinterpret_cast<f32x4s *>(dst + 8 * j) = v0;      // (non-temporal stores measured slower: 313 against 306 us per search in tools/psel_probe.py)
                *reinterpret_cast<f32x4s *>(dst + 32 + 8 * j) = v1;
            }
        } else {
            // sample pass: the maximum score of this query over the whole 64-row tile
            const uint64_t tile_row0 = (uint64_t)sel * a.tile_stride * MF_TR;
            float m = -__builtin_inff();
            if (tile_row0 + MF_TR <= a.n_rows) {
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaxf(acc0[r], acc1[r]));
            } else {                                 // the last, partial tile: rows past the end do not count
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t g0 = tile_row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (g0 < a.n_rows) m = fmaxf(m, acc0[r]);
                    if (g0 + 32 < a.n_rows) m = fmaxf(m, acc1[r]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 32));        // the two half-waves hold different rows of the same query
            if (hi == 0) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + q_local] = m * MF_INV_SCALE2;
        }
      } else {
        if (dma_wave) {
            if (MF_POLL_LATE) wait_others();
#pragma unroll
            for (int i = 0; i < NPC; ++i) glds16(psrc, srcoff[i], pdst + i * 4096);
        }
        if (MODE == MF_MODE_BLOCKMAX && lane < 32) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + q_local] = 0.0f;   // padding queries: defined values
      }
        if (DYN && dyn_on && wave >= 6) {
            // (before the drain below: the look issued at the top of this tile is then the only memory operation of this wave in flight, and long landed)
            if (wave == 6 && (tile_no & (DYN_P - 1u)) == (1u & (DYN_P - 1u))) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // the highest offered level of each owned query whose counter has reached k is published (one owner per query: a plain store)
                // rows at or above level (lane % 16) + 1 = the histogram summed from the top of this lane's 16-lane row (row_shl: lane i reads lane i + n, 0 past the row)
                uint32_t cum = own_l[lane];
                cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x101, 0xF, 0xF, true);
                cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x102, 0xF, 0xF, true);
                cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x104, 0xF, 0xF, true);
                cum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cum, 0x108, 0xF, 0xF, true);
                cum += __shfl_xor(cum, 16);                    // ... over the four replicas (rows of the wave)
                cum += __shfl_xor(cum, 32);
                bool reached = own_ok && cum >= a.k && own_thr < 3.0e38f;
#ifdef SHODH_DIAG
                if (a.ablate & 16u) reached = false;      // diagnostics: everything of the mechanism runs, nothing is published (its cost at the static survivor count)
#endif
                const uint64_t m_ = __builtin_amdgcn_ballot_w64(reached);
                const uint32_t grp = (uint32_t)(m_ >> (lane & 48)) & 0xFFFFu;
                if (reached && lane < 16 && (grp >> (lane & 15)) == 1u) {
                    const uint32_t word = (__float_as_uint(own_thr) & ~31u) | ((uint32_t)(lane & 15) + 1u);
                    if (word > pub_l[own_q])        // (what wave 7 read a tile ago: not published yet, or not seen yet)
                        __hip_atomic_store(a.dpub + (size_t)pass * MF_BPAD + own_q, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (wave == 7 && (tile_no & (DYN_P - 1u)) == (2u & (DYN_P - 1u))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the words are in pub_l once it returns)
        }
        if (MODE == MF_MODE_EMIT && active) {
            // drain early when the queue is half full; with the dynamic threshold also at the end of every DYN_P-th tile whatever the fill, all waves on
            // the same tile, so that the counts the bound rises on are fresh. No vmcnt(0) after it (there was one until the drains became periodic: it
            // waited out the DMA two tiles ahead, ~1 us fifteen times per launch). The counted wait below stays SAFE with stores and atomics in flight:
            // loads return in order, so if a piece of the NEXT tile were still outstanding, all NPC younger pieces would be too and the count could not
            // be down to NPC; stores that are still outstanding only make the wait stricter.
            if (wq_n >= (uint32_t)MF_WQ_CAP / 2 || (DYN && dyn_on && wq_n && (tile_no & (DYN_P - 1u)) == DYN_P - 1u)) drain();
        }
        // hand-over: the tile after this one must have landed (the DMA issued during this tile may stay in flight
        // when there are three buffers); all LDS traffic of this wave done; then the workgroup barrier
        PROF_T(4)
        if (MF_POLL_EARLY && PF == 2) {}       // (the DMA waves wait for their pieces in the middle of the next tile)
        else if (PF == 2 && MODE == MF_MODE_EMIT) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PROF_T(5)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        if (MF_POLL) {
            // this wave is through with the tile (all its LDS reads have returned); a DMA wave has also seen its pieces of the NEXT tile land (the counted wait above)
            const uint32_t done = tile_no + 1u;
            if (lane == 0) {
                const uint32_t a0 = smem_lds + (uint32_t)((unsigned char *)sync_l - smem);
                asm volatile("ds_write_b32 %0, %1" ::"v"(a0 + (8u + (uint32_t)wave) * 4u), "v"(done) : "memory");
                if (dma_wave && !(MF_POLL_EARLY && PF == 2)) { const uint32_t ld_ = done + 1u; asm volatile("ds_write_b32 %0, %1" ::"v"(a0 + (uint32_t)(wave & 3) * 4u), "v"(ld_) : "memory"); }
            }
        } else {
#ifndef SHODH_NO_TILE_BARRIER      // (diagnostic builds: what the per-tile barrier costs; results invalid)
        __builtin_amdgcn_s_barrier();
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        PROF_T(6)
        PROF_N(7)
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
#ifdef SHODH_PROF
    if (MODE == MF_MODE_EMIT && blockIdx.x == 200 && blockIdx.y == 0 && lane == 0 && pt_[7])
        printf("wave %d tiles %lld | per tile: landed-wait %lld top %lld chain0 %lld emit1 %lld others-wait %lld chain1 %lld emit0 %lld vmcnt %lld hand-over %lld\n", wave, pt_[7],
               pt_[8] / pt_[7], pt_[0] / pt_[7], pt_[1] / pt_[7], pt_[2] / pt_[7], pt_[9] / pt_[7], pt_[3] / pt_[7], pt_[4] / pt_[7], pt_[5] / pt_[7], pt_[6] / pt_[7]);
    if (MODE == MF_MODE_EMIT && blockIdx.x == 200 && blockIdx.y == 0 && lane == 0 && pt_[7] && ep_[0])
        printf("wave %d emit: %lld blocks entered, %lld cycles each, %lld survivors | %lld drains, %lld cycles each, %lld entries\n", wave, ep_[0], ep_[1] / ep_[0], ep_[2], ep_[3], ep_[3] ? ep_[4] / ep_[3] : 0, ep_[5]);
#endif
#ifdef SHODH_PROF
    const long long wc2_ = wall_clock64();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the clamped tail DMA before the LDS is released
#ifdef SHODH_PROF
    if (MODE == MF_MODE_SCORES && tid == 0) atomicMax(&g_scores_end_ticks, (unsigned long long)wall_clock64());
#endif
    if (MODE == MF_MODE_EMIT) {
        if (have_prev && active) emit_block(acc1, (uint64_t)prev_sel * a.tile_stride * MF_TR + 32);   // (no maximum was folded for the last tile)
        drain();
    }
#ifdef SHODH_PROF
    if (MODE == MF_MODE_EMIT && (blockIdx.x == 100 || blockIdx.x == 200) && blockIdx.y == 0 && tid == 0)
        printf("block %u: prologue %lld loop %lld epilogue %lld (10 ns ticks)\n", blockIdx.x, wc1_ - wc0_, wc2_ - wc1_, (long long)wall_clock64() - wc2_);
#endif
}

// ---- dimensions 768 and 1024 ---------------------------------------------------------------------------------------------------
// 256 resident queries x 1024 dimensions of fp16 are 512 KiB -- the whole vector register file of a CU -- so the kernel above (a wave
// = 32 queries resident as dim/16 B fragments next to a second wave on the same SIMD) stops at dim 512. Here a workgroup is FOUR waves,
// one per SIMD with the full 512 registers: 128 queries per workgroup, and a 256-query pass is two workgroups (blockIdx.y = 2 * pass +
// sub-pass) that walk the same tiles side by side -- the launch is sized so that both halves of a pass are resident together on the same
// XCD (grid.x = CUs / 2), which makes the second reader of a tile an L2 hit rather than a second trip to HBM. A 64-row tile is 128 KiB at
// dim 1024, so it goes through the LDS as two 32-row halves (two buffers). One wave per SIMD has nobody to hide an LDS-DMA's issue stall
// behind (100-185 cycles each, see above), so the next half travels HBM -> registers (issued before the MFMA chain, 64 spare registers)
// -> ds_write after the chain. Everything downstream (tile numbering, thresholds, candidate slots, final stage) is shared with the
// 512-thread kernel: same tile = 64 rows, same 256-query pass layout.
constexpr int MFB_TR = 32;
template <int MODE, int KSTEPS>
__global__ __launch_bounds__(256, 1) void mfma_scan_big_kernel(MfmaArgs a) {
    constexpr int NT = 256;
    constexpr int DIM = KSTEPS * 16;
    constexpr int CPR = KSTEPS * 2;
    constexpr int PITCH = DIM * 2;
    constexpr int HALF_BYTES = MFB_TR * PITCH;
    constexpr int NPC = MFB_TR * CPR / NT;     // 16-B pieces per thread and half tile (12 / 16)
    constexpr int D = 6, RING = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *eq_key = reinterpret_cast<uint64_t *>(smem + 2 * HALF_BYTES);
    uint32_t *eq_q = reinterpret_cast<uint32_t *>(eq_key + MF_EQ_CAP);
    uint32_t *qcount = eq_q + MF_EQ_CAP;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t pass = blockIdx.y >> 1, sub = blockIdx.y & 1;
    const uint32_t q_local = sub * 128 + wave * 32 + l31;
    const bool active = (uint32_t)(pass * MF_BPAD + sub * 128 + wave * 32) < a.nq;      // wave-uniform
    if ((uint32_t)(pass * MF_BPAD + sub * 128) >= a.nq) {        // a sub-pass of nothing but padding: no reason to stream the corpus for it
        if (MODE != MF_MODE_EMIT)                                 // (its sampled maxima still get defined values)
            for (uint32_t sel = blockIdx.x; sel < a.n_sel_tiles; sel += gridDim.x)
                if (tid < 128) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + sub * 128 + tid] = 0.0f;
        return;
    }
    if (MODE == MF_MODE_EMIT) {
        qcount[tid] = 0;
        if (tid < 128) {
            uint64_t *sl = a.slots + (((size_t)pass * MF_BPAD + sub * 128 + tid) * gridDim.x + blockIdx.x) * MF_SLOTS;
#pragma unroll
            for (int j = 0; j < MF_SLOTS; ++j) sl[j] = KEY_NONE;
        }
    }
    // piece p = i * NT + tid of a half tile: LDS chunk (row, slot) <- global chunk (row, slot ^ (row & 15) within its group of 16)
    uint32_t srcoff[NPC], dstoff[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int pc = i * NT + tid, row = pc / CPR, slot = pc % CPR;
        const int c = (slot & ~15) | ((slot & 15) ^ (row & 15));
        srcoff[i] = (uint32_t)(row * PITCH + c * 16);
        dstoff[i] = (uint32_t)(row * PITCH + slot * 16);
    }
    const unsigned char *rows_b = reinterpret_cast<const unsigned char *>(a.rows_h);
    const size_t tile_bytes_g = (size_t)a.tile_stride * MF_TR * DIM * 2;

    half8 bq[KSTEPS];
    {
        const half8 *qp = reinterpret_cast<const half8 *>(a.q_h) + ((size_t)pass * 8 + sub * 4 + wave) * KSTEPS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) bq[ks] = qp[ks * 64];
    }
    float thr_l = (MODE == MF_MODE_EMIT) ? a.thr[(size_t)pass * MF_BPAD + q_local] * (MF_SCALE * MF_SCALE) : 0.0f;
    if (a.ablate & 8u) thr_l = __builtin_inff();
    const int sw = l31 & 15;
    int aoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aoff[j] = l31 * PITCH + (((2 * j + hi) ^ sw) << 4);

    // survivors: the same wave-private queues and candidate slots as the 512-thread kernel
    uint64_t *wq_key = eq_key + wave * MF_WQ_CAP;
    uint32_t *wq_q = eq_q + wave * MF_WQ_CAP;
    uint32_t wq_n = 0;
    uint64_t *my_slots = a.slots + ((size_t)pass * MF_BPAD * gridDim.x + blockIdx.x) * MF_SLOTS;
    auto emit_direct = [&](uint64_t key, uint32_t ql) {
        const uint32_t row = (uint32_t)key;
        if (a.deleted && ((a.deleted[row >> 5] >> (row & 31)) & 1u)) return;      // tombstoned (vamana.rs:1175-1177)
        const uint32_t s_ = atomicAdd(qcount + ql, 1u);
        if (s_ < (uint32_t)MF_SLOTS) {
            my_slots[(size_t)ql * gridDim.x * MF_SLOTS + s_] = key;
        } else {
            const size_t qi = (size_t)pass * MF_BPAD + ql;
            const uint32_t slot = atomicAdd(a.cand_cnt + qi, 1u);
            if (slot < a.cand_cap) a.cand[qi * a.cand_cap + slot] = key;
        }
    };
    auto drain = [&]() {
        const uint32_t n = wq_n < (uint32_t)MF_WQ_CAP ? wq_n : (uint32_t)MF_WQ_CAP;
        for (uint32_t i = lane; i < n; i += 64) emit_direct(wq_key[i], wq_q[i]);
        wq_n = 0;
    };
    auto emit_block = [&](const floatx16 &c, uint64_t brow0) {
        const uint64_t left = a.n_rows > brow0 + 4 * hi ? a.n_rows - (brow0 + 4 * hi) : 0;
        const uint32_t lim = left < 64 ? (uint32_t)left : 64u;
        const uint32_t wq_n0 = wq_n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t roff = (r & 3) + 8 * (r >> 2);
            const bool hit = c[r] >= thr_l && roff < lim;
            const uint64_t b = __builtin_amdgcn_ballot_w64(hit);
            if (__builtin_expect(b != 0, 0)) {
                const uint32_t slot = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (hit && slot < (uint32_t)MF_WQ_CAP) {
                    wq_key[slot] = make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)(brow0 + 4 * hi + roff));
                    wq_q[slot] = q_local;
                }
                wq_n = __builtin_amdgcn_readfirstlane(wq_n + (uint32_t)__builtin_popcountll(b));
            }
        }
        if (__builtin_expect(wq_n > (uint32_t)MF_WQ_CAP, 0)) {      // a dense block: straight to the candidate slots (see the kernel above)
            wq_n = wq_n0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t roff = (r & 3) + 8 * (r >> 2);
                if (c[r] >= thr_l && roff < lim) emit_direct(make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)(brow0 + 4 * hi + roff)), q_local);
            }
        }
    };

    // The tiles of this workgroup: sel = blockIdx.x + i * gridDim.x, each as two 32-row halves (LDS buffer 0 / 1). Every half is
    // fetched into registers TWO halves ahead (sets A and B, alternating) and written to its LDS buffer one half ahead: with a single
    // half in flight the kernel ran at HBM latency, not bandwidth (64 KiB per CU per ~2.5 us round trip: 874 us per 256 queries
    // at 1M x 1024).
    const uint32_t step = gridDim.x;
    const uint32_t n_mine = blockIdx.x < a.n_sel_tiles ? (a.n_sel_tiles - blockIdx.x + step - 1) / step : 0u;
    const uint32_t n_half = 2 * n_mine;
    u32x4 stg_a[NPC], stg_b[NPC];
    auto src_of = [&](uint32_t u) -> const unsigned char * {
        const uint32_t uc = u < n_half ? u : n_half - 1;              // past the end: a harmless repeat of the last half
        const uint32_t sel = blockIdx.x + (uc >> 1) * step;
        return rows_b + (size_t)sel * tile_bytes_g + (size_t)(uc & 1) * HALF_BYTES;      // the shadow slab is padded to whole tiles
    };
    auto stage = [&](uint32_t buf, const u32x4 *stg) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) *reinterpret_cast<u32x4 *>(smem + buf * HALF_BYTES + dstoff[i]) = stg[i];
    };
    const floatx16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float tile_m = -__builtin_inff();
    // one half tile: the MFMA chain over LDS buffer `h`, then its epilogue
    auto compute = [&](uint32_t h, uint32_t sel) {
        const unsigned char *buf = smem + h * HALF_BYTES;
        const uint64_t row0 = (uint64_t)sel * a.tile_stride * MF_TR + h * MFB_TR;
        if (active) {
            // (two accumulator chains, even / odd k-steps, measured no faster at 768 and slower at 1024, where they spill)
            floatx16 acc = zero16;
            half8 ring[RING];
#pragma unroll
            for (int st = 0; st < D; ++st) ring[st % RING] = *reinterpret_cast<const half8 *>(buf + aoff[st & 7] + (st >> 3) * 256);
#pragma unroll
            for (int st = 0; st < KSTEPS; ++st) {
                if (st + D < KSTEPS) ring[(st + D) % RING] = *reinterpret_cast<const half8 *>(buf + aoff[(st + D) & 7] + ((st + D) >> 3) * 256);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[st % RING], bq[st], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // keeps the A fragments D steps ahead (without it the scheduler folds the ring into one register
            }                                           // quad and every MFMA waits out its own LDS read)
            if (MODE == MF_MODE_EMIT) {
                float m = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                if (__builtin_amdgcn_ballot_w64(m >= thr_l) != 0) emit_block(acc, row0);
                if (wq_n >= (uint32_t)MF_WQ_CAP / 2) drain();
            } else {
                float m = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t g0 = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (g0 < a.n_rows) m = fmaxf(m, acc[r]);
                }
                tile_m = fmaxf(tile_m, m);
                if (h) {
                    tile_m = fmaxf(tile_m, __shfl_xor(tile_m, 32));
                    if (hi == 0) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + q_local] = tile_m * MF_INV_SCALE2;
                    tile_m = -__builtin_inff();
                }
            }
        } else if (MODE != MF_MODE_EMIT && h && lane < 32) {
            a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + q_local] = 0.0f;      // padding queries: defined values
        }
    };
    if (n_mine) {
        {
            const unsigned char *s0 = src_of(0), *s1 = src_of(1);
#pragma unroll
            for (int i = 0; i < NPC; ++i) stg_a[i] = *reinterpret_cast<const u32x4 *>(s0 + srcoff[i]);
#pragma unroll
            for (int i = 0; i < NPC; ++i) stg_b[i] = *reinterpret_cast<const u32x4 *>(s1 + srcoff[i]);
        }
        stage(0, stg_a);
    }
    __syncthreads();
    for (uint32_t i = 0; i < n_mine; ++i) {
        const uint32_t sel = blockIdx.x + i * step;
        {   // half 0 from buffer 0; set A takes half 0 of the next tile, set B (half 1 of this tile) goes to buffer 1
            const unsigned char *sn = src_of(2 * i + 2);
#pragma unroll
            for (int j = 0; j < NPC; ++j) stg_a[j] = *reinterpret_cast<const u32x4 *>(sn + srcoff[j]);
            compute(0, sel);
            stage(1, stg_b);
            __syncthreads();
        }
        {   // half 1 from buffer 1; set B takes half 1 of the next tile, set A goes to buffer 0
            const unsigned char *sn = src_of(2 * i + 3);
#pragma unroll
            for (int j = 0; j < NPC; ++j) stg_b[j] = *reinterpret_cast<const u32x4 *>(sn + srcoff[j]);
            compute(1, sel);
            stage(0, stg_a);
            __syncthreads();
        }
    }
    if (MODE == MF_MODE_EMIT) drain();
}

// ---- dimensions 768 and 1024, round 4: sixteen queries per wave ---------------------------------------------------------------------------
// mfma_scan_big_kernel leaves a SIMD idle whenever its one wave waits (first A fragments after a barrier, the staging writes, the epilogue): a half
// tile takes ~5 500 cycles where its 48 MFMAs take 1 536. What forces one wave per SIMD there is the resident query fragments -- 32 queries x 768
// dimensions are 192 registers. v_mfma_f32_16x16x32_f16 halves them: a wave holds SIXTEEN queries (dim/32 fragments of 4 registers: 96 / 128),
// eight waves = the 128 queries of a sub-pass, no exchange between waves, the accumulators of a 32-row half tile are 2 x 4 registers. Every
// wave reads the whole half tile from LDS (twice the LDS traffic per flop of the 32-query shape: 384 KiB per half tile at 768 dimensions =
// 1536 LDS cycles at the 256 B/clk of ds_read_b128, as many as the SIMD's 96 MFMAs of 16 cycles take), which the freed registers pay for: the
// half tiles arrive by LDS-DMA through a ring of three buffers two ahead (768) or two buffers one ahead (1024: 64 KiB per half tile) as in
// mfma_scan_kernel -- no staging registers, no ds_write pass -- and the DMA pieces are issued inside the chain. Measured at 1M rows, 256 queries, one
// box, kernel us / step ms: 768 dimensions 474 / 0.577 against the one-wave kernel's 607 / 0.709; 1024 dimensions 557 / 0.668 against 891 / 1.006.
// (Also built and measured at 768 dimensions, then dropped: the 32-query shape with the DIMENSIONS split over the two waves of a SIMD and their partial
// sums exchanged through LDS behind the half-tile barrier -- 537 us; it keeps the register staging and does not fit 1024 dimensions.)
// C layout (16x16): lane l holds query l & 15 and rows 4 (l >> 4) .. + 3 of the 16-row block.
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int KSTEPS>
__host__ __device__ constexpr int big3_nbuf() { return (3 * MFB_TR * KSTEPS * 32 + MF_EQ_CAP * 12 + MF_BPAD * 4 + 64 + 64 <= 160 * 1024) ? 3 : 2; }
// The half-tile hand-over without a workgroup barrier, as in mfma_scan_kernel (see SHODH_MF_POLL there): here all eight waves issue DMA, so there are eight
// "landed" and eight "finished" words; the pieces of a half tile are issued during the SECOND half of its chain, behind the look at the others' progress, and
// "landed" is posted there too. Three buffers only (768 dimensions): emit kernel 425 -> 421 us at k = 10, 464 -> 446 at k = 120 (same box). 0: the s_barrier of rounds 4 - 5.
#ifndef SHODH_BIG3_POLL
#define SHODH_BIG3_POLL 1
#endif
template <int MODE, int KSTEPS>
__global__ __launch_bounds__(512, 1) void mfma_scan_big3_kernel(MfmaArgs a) {
    constexpr int NT = 512;
    constexpr int DIM = KSTEPS * 16;
    constexpr int CPR = KSTEPS * 2;            // 16-B chunks per row
    constexpr int PITCH = DIM * 2;
    constexpr int HALF_BYTES = MFB_TR * PITCH;
    constexpr int NPC = HALF_BYTES / (NT * 16); // DMA pieces per thread and half tile (6 / 8)
    constexpr int KS32 = KSTEPS / 2;           // k-steps of 32
    constexpr int NS = 2 * KS32;               // MFMAs per wave and half tile (row block rb = st / KS32)
#ifndef SHODH_BIG3_D      // (diagnostic builds time other fragment distances)
#define SHODH_BIG3_D 6
#endif
    constexpr int D = SHODH_BIG3_D, RING = 8;
    constexpr int NBUF = big3_nbuf<KSTEPS>();
    constexpr int PF = NBUF - 1;
    static_assert(NPC * 6 <= NS, "DMA issue slots: one piece every sixth step");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *eq_key = reinterpret_cast<uint64_t *>(smem + NBUF * HALF_BYTES);
    uint32_t *eq_q = reinterpret_cast<uint32_t *>(eq_key + MF_EQ_CAP);
    uint32_t *qcount = eq_q + MF_EQ_CAP;
    uint32_t *sync_l = qcount + MF_BPAD;       // [16] SHODH_BIG3_POLL: [0 .. 7] half tiles whose pieces wave w has seen land, [8 .. 15] half tiles wave w is through with
    constexpr bool POLL = SHODH_BIG3_POLL != 0 && PF == 2;      // (two buffers, 1024 dimensions: the late issue leaves the pieces half a tile to land -- 544 -> 603 us; the barrier form stays)
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t pass = blockIdx.y >> 1, sub = blockIdx.y & 1;
    const uint32_t q_local = sub * 128 + wave * 16 + l15;
    const bool active = (uint32_t)(pass * MF_BPAD + sub * 128 + wave * 16) < a.nq;      // wave-uniform
    if ((uint32_t)(pass * MF_BPAD + sub * 128) >= a.nq) {        // a sub-pass of nothing but padding: no reason to stream the corpus for it
        if (MODE != MF_MODE_EMIT)
            for (uint32_t sel = blockIdx.x; sel < a.n_sel_tiles; sel += gridDim.x)
                if (tid < 128) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + sub * 128 + tid] = 0.0f;
        return;
    }
    if (MODE == MF_MODE_EMIT) {
        if (tid < MF_BPAD) qcount[tid] = 0;
        if (tid < 128) {
            uint64_t *sl = a.slots + (((size_t)pass * MF_BPAD + sub * 128 + tid) * gridDim.x + blockIdx.x) * MF_SLOTS;
#pragma unroll
            for (int j = 0; j < MF_SLOTS; ++j) sl[j] = KEY_NONE;
        }
    }
    // DMA: piece p = i * NT + tid lands at LDS byte p * 16 of the buffer (lane-linear); the lane that owns chunk (row, slot) fetches global chunk
    // (row, slot ^ (row & 15) within its group of 16)
    uint32_t srcoff[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int pc = i * NT + tid, row = pc / CPR, slot = pc % CPR;
        const int c = (slot & ~15) | ((slot & 15) ^ (row & 15));
        srcoff[i] = (uint32_t)(row * PITCH + c * 16);
    }
    const unsigned char *rows_b = reinterpret_cast<const unsigned char *>(a.rows_h);
    const size_t tile_bytes_g = (size_t)a.tile_stride * MF_TR * DIM * 2;
    const uint32_t wave_lds = smem_lds + (uint32_t)wave * 1024u;
    const uint32_t step = gridDim.x;
    const uint32_t n_mine = blockIdx.x < a.n_sel_tiles ? (a.n_sel_tiles - blockIdx.x + step - 1) / step : 0u;
    const uint32_t n_half = 2 * n_mine;
    auto src_of = [&](uint32_t u) -> const unsigned char * {
        const uint32_t uc = u < n_half ? u : (n_half ? n_half - 1 : 0);      // past the end: a harmless repeat of the last half
        const uint32_t sel = blockIdx.x + (uc >> 1) * step;
        return uniform_ptr(rows_b + (size_t)sel * tile_bytes_g + (size_t)(uc & 1) * HALF_BYTES);      // the shadow slab is padded to whole tiles
    };
    if (n_half) {
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            const unsigned char *src = src_of(b);
#pragma unroll
            for (int i = 0; i < NPC; ++i) glds16(src, srcoff[i], wave_lds + b * HALF_BYTES + i * (NT * 16));
        }
    }
    // resident B fragments of the 16x16x32 shape, gathered from the 32x32x16 fragment-major copy of the queries (convert_queries_kernel): lane l wants
    // query l & 15 of its wave, k = 32 ks + 8 (l >> 4) .. + 8 = k-step 2 ks + (l >> 5) of the 16-wide layout, half (l >> 4) & 1
    half8 bq[KS32];
    {
        const uint32_t ql = sub * 128 + wave * 16 + l15;
        const half8 *qp = reinterpret_cast<const half8 *>(a.q_h) + ((size_t)pass * 8 + (ql >> 5)) * KSTEPS * 64 + (ql & 31) + 32 * (lg & 1);
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) bq[ks] = qp[(2 * ks + (lg >> 1)) * 64];
    }
    float thr_l = (MODE == MF_MODE_EMIT) ? a.thr[(size_t)pass * MF_BPAD + q_local] * (MF_SCALE * MF_SCALE) : 0.0f;
    if (a.ablate & 8u) thr_l = __builtin_inff();
    // A fragment of row block rb, k-step ks: row rb * 16 + l15, chunk 4 ks + lg, swizzled by row & 15 = l15 inside its group of 16 chunks
    int aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[j] = l15 * PITCH + (((4 * j + lg) ^ l15) << 4);

    // survivors: wave-private queues and the workgroup's candidate slots, as in mfma_scan_kernel
    uint64_t *wq_key = eq_key + wave * MF_WQ_CAP;
    uint32_t *wq_q = eq_q + wave * MF_WQ_CAP;
    uint32_t wq_n = 0;
    uint64_t *my_slots = a.slots + ((size_t)pass * MF_BPAD * gridDim.x + blockIdx.x) * MF_SLOTS;
    auto emit_direct = [&](uint64_t key, uint32_t ql) {
        const uint32_t row = (uint32_t)key;
        if (a.deleted && ((a.deleted[row >> 5] >> (row & 31)) & 1u)) return;      // tombstoned (vamana.rs:1175-1177)
        const uint32_t s_ = atomicAdd(qcount + ql, 1u);
        if (s_ < (uint32_t)MF_SLOTS) {
            my_slots[(size_t)ql * gridDim.x * MF_SLOTS + s_] = key;
        } else {
            const size_t qi = (size_t)pass * MF_BPAD + ql;
            const uint32_t slot = atomicAdd(a.cand_cnt + qi, 1u);
            if (slot < a.cand_cap) a.cand[qi * a.cand_cap + slot] = key;
        }
    };
    auto drain = [&]() {
        const uint32_t n = wq_n < (uint32_t)MF_WQ_CAP ? wq_n : (uint32_t)MF_WQ_CAP;
        for (uint32_t i = lane; i < n; i += 64) emit_direct(wq_key[i], wq_q[i]);
        wq_n = 0;
    };
    // the survivors of one 16-row block (entered by the whole wave): value r of a lane is row 4 lg + r
    auto emit_block = [&](const floatx4 &c, uint64_t brow0) {
        const uint64_t left = a.n_rows > brow0 + 4 * lg ? a.n_rows - (brow0 + 4 * lg) : 0;
        const uint32_t lim = left < 4 ? (uint32_t)left : 4u;
        const uint32_t wq_n0 = wq_n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool hit = c[r] >= thr_l && (uint32_t)r < lim;
            const uint64_t b = __builtin_amdgcn_ballot_w64(hit);
            if (__builtin_expect(b != 0, 0)) {
                const uint32_t slot = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (hit && slot < (uint32_t)MF_WQ_CAP) {
                    wq_key[slot] = make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)(brow0 + 4 * lg + r));
                    wq_q[slot] = q_local;
                }
                wq_n = __builtin_amdgcn_readfirstlane(wq_n + (uint32_t)__builtin_popcountll(b));
            }
        }
        if (__builtin_expect(wq_n > (uint32_t)MF_WQ_CAP, 0)) {      // a dense block: straight to the candidate slots (see mfma_scan_kernel)
            wq_n = wq_n0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c[r] >= thr_l && (uint32_t)r < lim) emit_direct(make_key(-(c[r] * MF_INV_SCALE2), (uint32_t)(brow0 + 4 * lg + r)), q_local);
        }
    };

    if (POLL && tid < 16) sync_l[tid] = tid < 8 ? (uint32_t)PF : 0u;      // the prologue's PF half tiles have landed (waited for below); nobody has finished one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const floatx4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    float tile_m = -__builtin_inff();
    uint32_t cur = 0;
    const uint32_t sync_lds = smem_lds + (uint32_t)((unsigned char *)sync_l - smem);
    auto poll_words = [&](uint32_t first, uint32_t need) {      // words first .. first + 7 >= need (lane l looks at word first + l % 8)
        const uint32_t paddr = sync_lds + (first + (uint32_t)(lane & 7)) * 4u;
        for (;;) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(paddr) : "memory");
            if (__builtin_amdgcn_ballot_w64(v >= need) == ~0ull) break;
            __builtin_amdgcn_s_sleep(2);
        }
    };
    constexpr int DSTEP = (NS / 2) / NPC;      // late issue: one piece every DSTEP-th step of the chain's second half
    static_assert(!POLL || DSTEP >= 1, "DMA issue slots in the second half of the chain");
    for (uint32_t u = 0; u < n_half; ++u) {
        const uint32_t sel = blockIdx.x + (u >> 1) * step, h = u & 1;
        const unsigned char *buf = smem + cur * HALF_BYTES;
        const uint32_t pfb = cur + PF >= NBUF ? cur + PF - NBUF : cur + PF;
        const unsigned char *psrc = src_of(u + PF);
        const uint32_t pdst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + pfb * HALF_BYTES));
        const uint64_t row0 = (uint64_t)sel * a.tile_stride * MF_TR + h * MFB_TR;
        if (POLL) poll_words(0, u + 1u);      // this half tile's data: every wave has seen its pieces of it land
        auto before_dma = [&]() {      // the buffer this half tile's DMA refills: everybody through with the half tile before this one
            poll_words(8, u);
            if (PF == 2) {      // what this wave has in flight here (the pieces of half tile u + 1, its last stores) is a half tile old: "u + 1 has landed" can be said now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) { const uint32_t ld_ = u + 2u; asm volatile("ds_write_b32 %0, %1" ::"v"(sync_lds + (uint32_t)wave * 4u), "v"(ld_) : "memory"); }
            }
        };
        if (active) {
            floatx4 acc0 = zero4, acc1 = zero4;
            half8 ring[RING];
            auto rd = [&](int st) {
                const int rb = st / KS32, ks = st % KS32;
                ring[st % RING] = *reinterpret_cast<const half8 *>(buf + rb * 16 * PITCH + aoff[ks & 3] + (ks >> 2) * 256);
            };
#pragma unroll
            for (int st = 0; st < D; ++st) rd(st);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                if (st + D < NS) rd(st + D);
                if (st < KS32) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[st % RING], bq[st], acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[st % RING], bq[st - KS32], acc1, 0, 0, 0);
                if (POLL) {
                    if (st == NS / 2) before_dma();
                    if (st >= NS / 2 && (st - NS / 2) % DSTEP == 0 && (st - NS / 2) / DSTEP < NPC) glds16(psrc, srcoff[(st - NS / 2) / DSTEP], pdst + ((st - NS / 2) / DSTEP) * (NT * 16));
                } else if (st % 6 == 2 && st / 6 < NPC) glds16(psrc, srcoff[st / 6], pdst + (st / 6) * (NT * 16));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == MF_MODE_EMIT) {
                const float m0 = fmaxf(fmaxf(acc0[0], acc0[1]), fmaxf(acc0[2], acc0[3])), m1 = fmaxf(fmaxf(acc1[0], acc1[1]), fmaxf(acc1[2], acc1[3]));
                bool emitted = false;
                if (__builtin_amdgcn_ballot_w64(m0 >= thr_l) != 0) { emit_block(acc0, row0); emitted = true; }
                if (__builtin_amdgcn_ballot_w64(m1 >= thr_l) != 0) { emit_block(acc1, row0 + 16); emitted = true; }
                if (wq_n >= (uint32_t)MF_WQ_CAP / 2) { drain(); emitted = true; }
                if (!POLL && emitted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores / atomics share the VM counter with the DMA: nothing of them may be pending at the counted wait
                (void)emitted;
            } else {
                float m = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t g0 = row0 + 4 * lg + r;
                    if (g0 < a.n_rows) m = fmaxf(m, acc0[r]);
                    if (g0 + 16 < a.n_rows) m = fmaxf(m, acc1[r]);
                }
                tile_m = fmaxf(tile_m, m);
                if (h) {
                    tile_m = fmaxf(tile_m, __shfl_xor(tile_m, 16));
                    tile_m = fmaxf(tile_m, __shfl_xor(tile_m, 32));
                    if (lg == 0) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + q_local] = tile_m * MF_INV_SCALE2;
                    tile_m = -__builtin_inff();
                }
            }
        } else {
            if (POLL) before_dma();
#pragma unroll
            for (int i = 0; i < NPC; ++i) glds16(psrc, srcoff[i], pdst + i * (NT * 16));
            if (MODE != MF_MODE_EMIT && h && lane < 16) a.blockmax[((size_t)pass * a.n_sel_tiles + sel) * MF_BPAD + q_local] = 0.0f;      // padding queries: defined values
        }
        // hand-over: the next half tile must have landed (with three buffers the pieces issued during this half tile may stay in flight)
        // (the sample pass stores its maxima every other half tile: stores and loads share the counter and may return out of order with each other, so it waits for everything)
        if (POLL) {
            if (PF != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // two buffers: the pieces issued during this half tile are the next one's
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of the half tile have returned
            if (lane == 0) {
                const uint32_t done = u + 1u;
                asm volatile("ds_write_b32 %0, %1" ::"v"(sync_lds + (8u + (uint32_t)wave) * 4u), "v"(done) : "memory");
                if (PF != 2) { const uint32_t ld_ = u + 2u; asm volatile("ds_write_b32 %0, %1" ::"v"(sync_lds + (uint32_t)wave * 4u), "v"(ld_) : "memory"); }
            }
        } else {
        if (PF == 2 && MODE == MF_MODE_EMIT) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the clamped tail DMA before the LDS is released
    if (MODE == MF_MODE_EMIT) drain();
}

// ---- conversions ------------------------------------------------------------------------------------
// rows f32 -> fp16(256 x) shadow; also folds max row norm^2, max |x| and a non-finite flag into stats.
// stats[0] = max norm^2 (float bits, atomicMax on uint works for non-negative floats)
// stats[1] = max |x| (float bits), stats[2] = non-finite count, stats[3] = max |row - shadow row|^2 (float bits)
__global__ __launch_bounds__(256) void convert_rows_kernel(const float *rows, uint64_t first, uint64_t n, uint32_t dim,
                                                           _Float16 *rows_h, uint32_t *stats) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint64_t wave_cnt = ((uint64_t)gridDim.x * 256) >> 6;
    float mx_n = 0.0f, mx_a = 0.0f, mx_r = 0.0f;
    uint32_t bad = 0;
    for (uint64_t r = first + wave_gid; r < first + n; r += wave_cnt) {
        const float *src = rows + r * dim;
        _Float16 *dst = rows_h + r * dim;
        float ss = 0.0f, rs = 0.0f;
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
                const float d = x - (float)h[j] * (1.0f / MF_SCALE);      // what the rounding to fp16 took away (the difference of two close floats: exact)
                rs = __builtin_fmaf(d, d, rs);
            }
            *reinterpret_cast<uint2 *>(dst + i) = *reinterpret_cast<const uint2 *>(h);
        }
        for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); rs += __shfl_xor(rs, off); }
        mx_n = fmaxf(mx_n, ss);
        mx_r = fmaxf(mx_r, (__builtin_fabsf(rs) <= 3.0e38f) ? rs : 0.0f);      // (a non-finite row is rejected by the caller: keep the maximum finite)
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx_a = fmaxf(mx_a, __shfl_xor(mx_a, off));
        bad += __shfl_xor(bad, off);
    }
    if (lane == 0) {
        atomicMax(stats + 0, __float_as_uint(mx_n));
        atomicMax(stats + 1, __float_as_uint(mx_a));
        if (bad) atomicAdd(stats + 2, bad);
        atomicMax(stats + 3, __float_as_uint(mx_r));      // max over the rows of |row - fp16 copy|^2: the corpus side of the pre-scan's error bound (eps_coefficients)
    }
}

// number of NaN / Inf values in x[0..n) added to *counter (indexes without a shadow copy: the check convert_rows_kernel does)
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float *x, uint64_t n, uint32_t *counter) {
    uint32_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        if (!(__builtin_fabsf(x[i]) <= 3.0e38f)) bad++;
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(counter, bad);
}

// zero (tombstone) or restore one row of the shadow copy
__global__ void shadow_set_row_kernel(const float *rows, _Float16 *rows_h, uint64_t row, uint32_t dim, int zero) {
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x)
        rows_h[row * dim + i] = zero ? (_Float16)0.0f : (_Float16)(rows[row * dim + i] * MF_SCALE);
}
// zero a list of rows of the shadow copy (mark_deleted for many ids at once); one workgroup per listed row
__global__ void shadow_zero_rows_kernel(_Float16 *rows_h, const uint32_t *list, uint32_t dim) {
    const uint64_t row = list[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) rows_h[row * dim + i] = (_Float16)0.0f;
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
    _Float16 *q_h;           // [n_slots * dim] fragment-major, see the kernel
    float *qnorm;            // [n_slots]
    float *qres;             // [n_slots] |q - fp16 copy| (what the query's rounding took away), a hair generous; lives in the workspace's eps array until threshold_kernel replaces it by eps
    uint32_t *cand_cnt;      // [n_slots]
    uint32_t *fallback;      // [n_slots] 1 = must go through the exact scan
    uint32_t *fb_count;      // single counter, zeroed here
    uint32_t *stats;         // [4] zeroed here: emitted, rescored, overflowed, -
};
__global__ __launch_bounds__(256) void convert_queries_kernel(QueryPrep p) {
    const int lane = threadIdx.x & 63;
    const uint32_t slot = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (blockIdx.x == 0 && threadIdx.x < 5) p.stats[threadIdx.x] = 0;       // (word 4: arrivals of the final stage, see its end)
    if (blockIdx.x == 0 && threadIdx.x == 4) *p.fb_count = 0;
    if (slot >= p.n_slots) return;
    float ss = 0.0f, mx = 0.0f, rs = 0.0f;
    uint32_t bad = 0;
    for (uint32_t i = lane; i < p.dim; i += 64) {
        float x = 0.0f;
        if (slot < p.nq) x = p.q[(size_t)slot * p.dim + i];
        if (!(__builtin_fabsf(x) <= 3.0e38f)) bad++;
        mx = fmaxf(mx, __builtin_fabsf(x));
        ss = __builtin_fmaf(x, x, ss);
        { const float d = x - (float)(_Float16)(x * MF_SCALE) * (1.0f / MF_SCALE); rs = __builtin_fmaf(d, d, rs); }
        // fragment-major: [pass][wave = q/32][k-step = i/16][lane = ((i%16)/8)*32 + q%32][i%8], so that the scan kernel's
        // 32x32x16 B fragment of one k-step is ONE contiguous 1 KiB wave load (row-major made every load instruction touch 32
        // different 128-B lines for 32 B each: ~9 us of prologue per launch)
        {
            const uint32_t ql = slot % MF_BPAD, ps = slot / MF_BPAD;
            const size_t frag = (((size_t)ps * 8 + ql / 32) * (p.dim / 16) + i / 16) * 64 + ((i % 16) / 8) * 32 + ql % 32;
            p.q_h[frag * 8 + i % 8] = (_Float16)(x * MF_SCALE);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        rs += __shfl_xor(rs, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
        bad += __shfl_xor(bad, off);
    }
    if (lane == 0) {
        p.qnorm[slot] = __builtin_sqrtf(ss) * 1.00001f;
        p.qres[slot] = (__builtin_fabsf(rs) <= 3.0e38f) ? __builtin_sqrtf(rs) * 1.0001f : 0.0f;      // (an unquantisable / non-finite query goes to the exact scan: its eps is never used)
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
    float eps_rel_maxnorm;   // coefficient of |q|: eps_rel * maxnorm + maxres (eps_coefficients)
    float eps_abs_a;         // coefficient of (|q| + maxnorm)
    float maxnorm;
    float res_mul;           // coefficient of the query's own rounding residual: maxnorm + maxres
    float *thr;              // [n_slots] emit threshold
    float *eps;              // [n_slots] in: the query's rounding residual (convert_queries_kernel); out: eps
    float eps2_rel_maxnorm;  // level-2 (f32 FMA re-score) bound: eps2 = eps2_rel_maxnorm * |q| + eps2_abs_a * (|q| + maxnorm) + 1e-9
    float eps2_abs_a;
    float *eps2;             // [n_slots]
    // dynamic threshold of the emit scan (MfmaArgs::dyn; nullptr = off): the level table of every query is laid out here
    uint32_t *dcnt;          // [n_slots][MF_DYN_NB] counters, zeroed here
    float *dthr;             // [n_slots][MF_DYN_NB] scaled emit threshold of each level (+inf: not offered)
    uint32_t *dpub;          // [n_slots] published bounds, zeroed here
    uint32_t *dpar;          // [n_slots] x {K_B, step}
    uint32_t r_top;          // the sample rank at which the FINAL k-th best is expected (k x sampled fraction): the levels span [k-th, r_top-th] sampled maximum
};
constexpr int THR_STAGE = 4096;
__global__ __launch_bounds__(256) void threshold_kernel(ThrArgs a) {
    __shared__ uint32_t scratch[KTH_SCRATCH_U32];
    __shared__ uint32_t staged[THR_STAGE];
    const int tid = threadIdx.x;
    const uint32_t slot = blockIdx.x;            // pass*256 + q
    const uint32_t pass = slot / MF_BPAD, ql = slot % MF_BPAD;
    PROF_DECL
    const float qn = a.qnorm[slot];
    const float qres = a.eps[slot];
    __syncthreads();             // (everybody has read the residual before thread 0 puts eps in its place)
    const float eps = a.eps_rel_maxnorm * qn + qres * a.res_mul + a.eps_abs_a * (qn + a.maxnorm) + 1e-9f;
    if (tid == 0) a.eps2[slot] = a.eps2_rel_maxnorm * qn + a.eps2_abs_a * (qn + a.maxnorm) + 1e-9f;
    if (slot >= a.nq || a.fallback[slot]) {
        if (tid == 0) { a.thr[slot] = __builtin_inff(); a.eps[slot] = eps; }   // never emits
        if (a.dcnt && tid < MF_DYN_REP * MF_DYN_NB) a.dcnt[(size_t)slot * MF_DYN_REP * MF_DYN_NB + tid] = 0u;
        if (a.dcnt && tid < MF_DYN_NB) a.dthr[(size_t)slot * MF_DYN_NB + tid] = __builtin_inff();
        if (a.dcnt && tid == 0) { a.dpar[2 * slot] = 0u; a.dpar[2 * slot + 1] = 0u; a.dpub[slot] = 0u; }
        return;
    }
    // k-th LARGEST tile maximum == k-th smallest order_key(-max). The maxima of one query are 1 KiB apart in memory and
    // are needed twice (filter, gather): the first THR_STAGE of them are staged in LDS.
    auto key_glb = [&](uint32_t j) -> uint32_t { return order_key(-a.blockmax[((size_t)pass * a.J + j) * MF_BPAD + ql]); };
    const uint32_t n_st = a.J < (uint32_t)THR_STAGE ? a.J : (uint32_t)THR_STAGE;
    for (uint32_t j0 = 0; j0 < n_st; j0 += 1024) {      // four unconditional loads in flight per thread (clamped index)
        uint32_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t j = j0 + u * 256 + tid; kv[u] = key_glb(j < n_st ? j : 0); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t j = j0 + u * 256 + tid; if (j < n_st) staged[j] = kv[u]; }
    }
    __syncthreads();
    PROF_T(0)
    auto key_at = [&](uint32_t j) -> uint32_t { return j < n_st ? staged[j] : key_glb(j); };
    bool ovf = false;
    const uint32_t kk = a.k ? block_kth_u32<256>(key_at, a.J, a.k, scratch, &ovf) : 0xFFFFFFFFu;
    PROF_T(1)
    // dynamic threshold: where the levels end. The FINAL k-th best has rank ~ k among all rows = rank ~ k / S among the sampled maxima: the r_top-th
    // largest of them (a second, much smaller selection) is where the bound is expected to end up, so the MF_DYN_NB levels are spread from the
    // k-th maximum (level 0 = the sampled bound) to a little beyond the r_top-th.
    // (Not a second selection -- that cost 5.6 us per step: the first one left the minima of k groups of the keys in scratch[0 .. k) when it filtered
    // (k <= 128 and J >= 4 k, topk.h); their r_top-th smallest is the r_top-th smallest key unless two of the best r_top share a group, and then a little
    // on the low side: good enough to place levels. Otherwise: the best sampled maximum.)
    __shared__ uint32_t s_ktop;
    if (tid == 0) s_ktop = 0xFFFFFFFFu;
    __syncthreads();
    if (a.dcnt && a.k && a.r_top && a.r_top < a.k) {
        if (a.k <= 128u && a.J >= 4u * a.k) {
            if ((uint32_t)tid < a.k) {
                const uint32_t v = scratch[tid];
                uint32_t r = 0;
                // ((value, index) as ONE 64-bit key: a compare and an add per pair instead of three compares and their mask arithmetic -- the loop is bound by its
                // instruction chain, not by the LDS reads: 8 000 cycles at k = 120, `profiles/r6_threshold_phases_before.txt`)
                const uint64_t v64 = ((uint64_t)v << 32) | (uint32_t)tid;
                for (uint32_t j = 0; j < a.k; ++j) { const uint64_t w64 = ((uint64_t)scratch[j] << 32) | j; r += (uint32_t)(w64 < v64); }
                if (r == a.r_top - 1u) s_ktop = v;
            }
        } else {
            uint32_t m = 0xFFFFFFFFu;
            for (uint32_t j = tid; j < a.J; j += 256) { const uint32_t v = key_at(j); m = v < m ? v : m; }
            atomicMin(&s_ktop, m);
        }
    }
    __syncthreads();
    const uint32_t ktop = s_ktop;
    PROF_T(2)
    {
        // (every thread holds kk, ktop and eps: the bound and the level parameters are computed by all, thread 0 stores the per-query words and sixteen threads one level
        // each -- as one thread's loop this tail was 3 600 cycles of every launch, `profiles/r6_threshold_phases_before.txt`)
        float t = -__builtin_inff();
        bool dyn_ok = false;
        if (kk != 0xFFFFFFFFu) {
            const float kth = -order_key_inv(kk);
            // a positive k-th tile max is backed by k LIVE rows (tombstoned rows score exactly 0)
            if (kth > 0.0f) { t = kth - (2.001f * eps + 1e-7f * __builtin_fabsf(kth)); dyn_ok = true; }
        }
        if (tid == 0) {
            a.thr[slot] = t;   // -inf => emit everything => list overflow => exact fallback
            a.eps[slot] = eps;
        }
        if (a.dcnt) {
            // level j <=> order key <= K_B - j * step <=> score >= E_j = -order_key_inv(K_B - j * step); E_0 = the sampled k-th maximum
            uint32_t kb = 0u, step = 1u;
            if (dyn_ok && ktop != 0xFFFFFFFFu && ktop <= kk) {
                kb = kk;
                uint32_t span = kk - ktop;
                if (span < 1024u) span = 1024u;
                step = (uint32_t)(((uint64_t)span + (span >> 3) + MF_DYN_NB - 1) / MF_DYN_NB);      // the levels end 1/8 beyond the expected final bound
            }
            if (tid == 0) { a.dpar[2 * slot] = kb; a.dpar[2 * slot + 1] = 0xFFFFFFFFu / step; a.dpub[slot] = 0u; }      // (floor((2^32 - 1) / step) <= floor(2^32 / step): never long)
            if (tid < MF_DYN_NB) {
                const uint32_t j = (uint32_t)tid + 1u;
                float tj = __builtin_inff();
                const uint64_t drop = (uint64_t)j * step;
                // (keys of positive scores are > 0x3F800000-ish: a level whose edge would leave the range of scores <= 2.0 is never offered)
                if (kb && drop < (uint64_t)kb && (kb - (uint32_t)drop) > 0x3F000000u) {
                    const float ej = -order_key_inv(kb - (uint32_t)drop);
                    if (ej > 0.0f && ej < 4.0f) tj = (ej - (2.001f * eps + 1e-7f * __builtin_fabsf(ej))) * (MF_SCALE * MF_SCALE);
                    if (!(tj > 0.0f)) tj = __builtin_inff();      // (the published word orders like the float only for positive thresholds)
                }
                for (int rp = 0; rp < MF_DYN_REP; ++rp) a.dcnt[((size_t)slot * MF_DYN_REP + rp) * MF_DYN_NB + (j - 1)] = 0u;
                a.dthr[(size_t)slot * MF_DYN_NB + (j - 1)] = tj;
            }
        }
    }
#ifdef SHODH_PROF
    PROF_T(3)
    if (tid == 0 && (slot % 37) == 0) printf("thr slot %u J %u k %u | stage %lld kth %lld ktop %lld tail %lld\n", slot, a.J, a.k, pt_[0], pt_[1], pt_[2], pt_[3]);
#endif
}



// ---- final stage: k-th best approximate score, 2-eps window, exact re-score, sort, write -----------------
struct FinalArgs {
    const float *rows;        // f32 master rows
    uint32_t dim;
    const float *q;           // [nq][dim] f32 queries
    uint32_t nq, k, cap;      // cap: TopKBuf capacity
    uint64_t *slots;          // [n_slots][nb][MF_SLOTS] private (query, workgroup) slots of the pre-scan, KEY_NONE = empty (level 2 rewrites them in place)
    uint32_t nb;              // workgroups of the pre-scan launch
    uint64_t *cand;           // [n_slots][cand_cap] shared overflow list (level 2 rewrites it in place)
    const uint32_t *cand_cnt;
    uint32_t cand_cap;
    const float *eps;         // [n_slots]
    const float *eps2;        // [n_slots] bound of the level-2 f32 score
    uint32_t fcap;            // capacity of the re-score list
    uint32_t stage_cap;       // candidate keys staged in LDS (<= FS_STAGE; the rest, if any, is re-read from global)
    uint32_t ch_rows;         // AVX2 / sequential order: rows of the window staged at a time (<= FS_CH)
    uint32_t order;
    uint32_t id_base;
    uint32_t *fallback;       // [n_slots] in/out
    uint32_t *fb_list;        // [n_slots] compacted list of fallback queries
    uint32_t *fb_count;
    uint32_t *ids;            // [nq][k]
    float *dist;
    uint32_t *counts;
    uint32_t *stats;          // [0] emitted, [1] rescored, [2] overflowed queries, [3] queries that went through level 2
    uint32_t *cnt_reset;      // single-query path: its overflow counter lives across calls and is handed back zeroed (nullptr otherwise)
    uint32_t *stats_mirror;   // one-query host calls: the four statistics words are copied here (host-mapped memory) by the one workgroup, after its own updates
    // Set mode (IVF-PQ probe selection, spann.rs:595-607: only WHICH partitions are probed reaches the result -- the postings of all of them are merged by
    // (distance, id) afterwards): a candidate whose approximate score beats the k-th best by more than 2 eps is in the exact top-k whatever its exact
    // distance is (at most k - 1 rows can have an exact score above kth~ + eps), one that loses by more than 2 eps is out; only the rows in between are
    // scored in the reference's order. With the strictly sequential sum of the centroid distance -- a 384-step dependent chain per row -- that is 1-3
    // rows per query instead of the ~50 of the whole window (the final stage was 74 of the 125 us of probe selection at 1024 queries x 4096 centroids).
    // Output: the k ids (the sure ones first, by approximate score), distances are placeholders (sure rows: -100 - s~).
    uint32_t set_only;
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

constexpr int FS_CH = 16;     // AVX2 order: rows of the re-score window staged raw in LDS at a time
constexpr int FS_RW = 16;      // scalar-4 order: rows in flight per wave; a re-score chunk is (waves x FS_RW) = NT / 4 rows whose per-group partial sums are staged
constexpr int FS_STAGE = 2048; // candidate keys staged in LDS (the rest, if any, is re-read from global)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// nt = threads of the workgroup (256 or 512): the scalar-4 re-score stages nt / 4 rows at a time (FS_RW rows in flight per wave)
__host__ __device__ inline size_t final_stage_region_bytes(uint32_t dim, uint32_t order, uint32_t ch_rows, uint32_t nt) {
    const size_t a = (size_t)(nt / 4) * (dim / 4 + 1) * 4;
    const size_t b = (size_t)ch_rows * (dim + 8) * 4 + (size_t)ch_rows * 8 * 4;
    return ((order == SHODH_ORDER_SCALAR4 ? a : b) + 15) & ~(size_t)15;
}

// NT = 256: several workgroups share a CU (thousands of queries: k-means assignment, IVF encoding). NT = 512 (up to a few hundred queries,
// one workgroup per CU anyway): twice the rows of the re-score in flight, twice the lanes in the selections -- at recall's index-level
// k = 120 the 256-thread form spent 47 % of its 67 us in five sequential 64-row re-score chunks and 32 % ranking 280 keys on 256 lanes.
template <int ORDER, int NT>
__global__ __launch_bounds__(NT) void final_stage_kernel(FinalArgs a) {
    constexpr int SJ = 1024 / NT;          // staged loads per thread (1024 slots / up to 1024 query floats)
    constexpr int NW = NT / 64;            // waves
    constexpr int TC = NT / 4;             // scalar-4 order: rows per re-score chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = reinterpret_cast<float *>(smem);                          // [dim]
    uint64_t *keys = reinterpret_cast<uint64_t *>(qs + a.dim);            // [cap]
    uint64_t *mins = keys + a.cap;                                        // [2 * NT]
    uint64_t *ekeys = mins + 2 * NT;                                         // [fcap] exact keys of the window
    uint64_t *thr = ekeys + a.fcap;
    uint32_t *flist = reinterpret_cast<uint32_t *>(thr + 1);              // [fcap] rows of the window
    uint32_t *cnt = flist + a.fcap;
    uint32_t *fcnt = cnt + 1;
    float *region = reinterpret_cast<float *>(fcnt + 1);                  // 16-B aligned: thr sits at a 16-B boundary
    uint32_t *sel32 = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(region) + final_stage_region_bytes(a.dim, ORDER, a.ch_rows, NT));   // [KTH_SCRATCH_U32]
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= a.nq) return;
    // candidates of query q: the workgroups' private slots (mostly empty: KEY_NONE) followed by the shared overflow list.
    // Everything the block needs first is requested in one go (unconditional loads on clamped indices: a load behind a
    // runtime condition makes hipcc wait for each one separately -- measured as 4 + 2 + 2 dependent round trips here).
    const uint32_t n_main = a.nb * MF_SLOTS;
    uint64_t *slots_q = a.slots + (size_t)q * n_main;
    uint64_t kv[SJ];
#pragma unroll
    for (int j = 0; j < SJ; ++j) { const uint32_t i = j * NT + tid; kv[j] = slots_q[i < n_main ? i : 0]; }
    float qv[SJ];
#pragma unroll
    for (int j = 0; j < SJ; ++j) { const uint32_t i = j * NT + tid; qv[j] = a.q[(size_t)q * a.dim + (i < a.dim ? i : 0)]; }   // dim <= 1024 on this path
    const uint32_t n_ovf = a.cand_cnt[q];
    if (a.cnt_reset) {                  // block-uniform; every thread has read the counter before it is handed back
        __syncthreads();
        if (tid == 0) a.cnt_reset[q] = 0;
    }
    bool bad = a.fallback[q] != 0 || n_ovf > a.cand_cap;
    const uint32_t n = n_main + (bad ? 0u : n_ovf);
    uint32_t *ecnt = sel32 + KTH_SCRATCH_U32 - 1;    // number of real candidates (statistics)
    uint32_t *skey = sel32 + KTH_SCRATCH_U32;        // [stage_cap] score keys (high words) of the candidates, staged once
    uint32_t *srow = skey + a.stage_cap;             // [stage_cap] their rows (low words)
    PROF_DECL
    if (!bad) {
#pragma unroll
        for (int j = 0; j < SJ; ++j) { const uint32_t i = j * NT + tid; if (i < a.dim) qs[i] = qv[j]; }
        if (tid == 0) { *fcnt = 0; *ecnt = 0; }
        TopKBuf buf{keys, cnt, thr, a.cap, a.k};
        uint64_t *list = a.cand + (size_t)q * a.cand_cap;
        auto key_glb = [&](uint32_t i) -> uint64_t { return i < n_main ? slots_q[i] : list[i - n_main]; };
        // the candidate keys are read three times (filter, gather, window): stage them in LDS once
        const uint32_t n_st = n < a.stage_cap ? n : a.stage_cap;
        const uint32_t st_main = n_main < n_st ? n_main : n_st;
#pragma unroll
        for (int j = 0; j < SJ; ++j) { const uint32_t i = j * NT + tid; if (i < st_main) { skey[i] = (uint32_t)(kv[j] >> 32); srow[i] = (uint32_t)kv[j]; } }
        for (uint32_t i0 = 1024; i0 < st_main; i0 += 1024) {      // more than 256 workgroups in the pre-scan
            uint64_t kw[SJ];
#pragma unroll
            for (int j = 0; j < SJ; ++j) { const uint32_t i = i0 + j * NT + tid; kw[j] = slots_q[i < st_main ? i : 0]; }
#pragma unroll
            for (int j = 0; j < SJ; ++j) { const uint32_t i = i0 + j * NT + tid; if (i < st_main) { skey[i] = (uint32_t)(kw[j] >> 32); srow[i] = (uint32_t)kw[j]; } }
        }
        for (uint32_t i = n_main + tid; i < n_st; i += NT) { const uint64_t key = list[i - n_main]; skey[i] = (uint32_t)(key >> 32); srow[i] = (uint32_t)key; }
        __syncthreads();
        auto key_at = [&](uint32_t i) -> uint64_t { return i < n_st ? (((uint64_t)skey[i] << 32) | srow[i]) : key_glb(i); };
        // pass A: k-th best approximate score (only its VALUE matters, so 32-bit score keys suffice for small k)
        PROF_T(0)
        // lower end of the window around the k-th best of the current candidate scores: kth - 2 eps (-inf while fewer than k exist)
        float kth_seen = __builtin_inff();       // the k-th best approximate score the last window_floor found (+inf: none)
        auto window_floor = [&](float eps_q) -> float {
            float lo_ = -__builtin_inff();
            kth_seen = __builtin_inff();
            if (a.k > 0 && a.k <= 128) {
                auto key32 = [&](uint32_t i) -> uint32_t { return i < n_st ? skey[i] : (uint32_t)(key_glb(i) >> 32); };    // empty slot -> 0xFFFFFFFF, ignored
                bool ovf = false;
                const uint32_t kk = block_kth_u32<NT>(key32, n, a.k, sel32, &ovf);
                if (kk != 0xFFFFFFFFu) {
                    const float kth = -order_key_inv(kk);
                    lo_ = kth - (2.001f * eps_q + 1e-7f * __builtin_fabsf(kth));
                    kth_seen = kth;
                }
            } else if (a.k > 0) {
                auto key_a = [&](uint64_t i) -> uint64_t { return key_at((uint32_t)i); };
                const uint32_t ma = block_select_topk<NT>(key_a, n, buf, mins);
                if (ma == a.k && buf.keys[a.k - 1] != KEY_NONE) {
                    const float kth = -order_key_inv((uint32_t)(buf.keys[a.k - 1] >> 32));
                    lo_ = kth - (2.001f * eps_q + 1e-7f * __builtin_fabsf(kth));
                }
            }
            return lo_;
        };
        // every candidate with score >= lo_ goes to the re-score list (all of them if fewer than k exist); returns the count
        // set mode: candidates with score > hi_ are sure members; their keys go to the END of ekeys (downwards), *scnt counts them
        uint32_t *scnt = ecnt - 1;                       // (the word before the statistics counter in the selection scratch: unused by block_kth_u32)
        auto collect_window = [&](float lo_, bool count_real, float hi_ = __builtin_inff()) -> uint32_t {
            __syncthreads();
            if (tid == 0) { *fcnt = 0; *scnt = 0; }
            __syncthreads();
            uint32_t real = 0;
            for (uint32_t i = tid; i < n; i += NT) {
                const uint64_t key = key_at(i);
                if (key == KEY_NONE) continue;
                ++real;
                const float s = -order_key_inv((uint32_t)(key >> 32));
                if (s > hi_) {
                    const uint32_t slot = atomicAdd(scnt, 1u);
                    if (slot < a.fcap) ekeys[a.fcap - 1u - slot] = make_key(-100.0f - s, a.id_base + (uint32_t)key);
                } else if (s >= lo_) {
                    const uint32_t slot = atomicAdd(fcnt, 1u);
                    if (slot < a.fcap) flist[slot] = (uint32_t)key;
                }
            }
            if (count_real && real) atomicAdd(ecnt, real);
            __syncthreads();
            return *fcnt;
        };
        const float lo = window_floor(a.eps[q]);
        PROF_T(1)
        // set mode needs a k-th best to stand on (block-uniform); the margin above it is the one below it
        const bool set_mode = a.set_only && kth_seen < 3.0e38f;
        const float hi = set_mode ? kth_seen + (2.001f * a.eps[q] + 1e-7f * __builtin_fabsf(kth_seen)) : __builtin_inff();
        uint32_t nf = collect_window(lo, true, hi);
        uint32_t n_sure = set_mode ? *scnt : 0u;
        if (nf + n_sure > a.fcap) {          // (cannot happen with n_sure < k <= fcap / 4 unless the window itself overflows: the plain path decides)
            nf = collect_window(lo, false);
            n_sure = 0;
        }
        PROF_T(2)
        if (nf > a.fcap) {
            // ---- level 2 (block-uniform branch) -----------------------------------------------------------------------
            // The fp16 window holds more rows than the reference-order re-score is sized for: a dense corpus (many rows within
            // 2 eps ~ 2e-3 of the k-th best score). Every row of that window gets an f32 score s2 (FMA, any order) from the
            // f32 master rows -- |s2 - dot_ref| <= eps2 ~ dim * 2^-23 |q| maxnorm, ~20x tighter than the fp16 bound -- which
            // replaces its fp16 score IN PLACE; the window is then rebuilt around the k-th best s2 with 2 eps2. Cost is
            // proportional to the survivors (one 4*dim-byte row read each), not to the corpus.
            // One wave per group of 64 candidates; up to 8 member rows in flight per wave.
            const uint32_t wv = tid >> 6, ln = tid & 63;
            const uint32_t dim = a.dim, d4 = dim >> 2;
            const f32x4 *qs4 = reinterpret_cast<const f32x4 *>(qs);
            const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            const f32x4 qa = ln < d4 ? qs4[ln] : z4, qb = ln + 64 < d4 ? qs4[ln + 64] : z4;      // two float4 groups per lane up to dim 512,
            const uint32_t ga = ln < d4 ? ln : 0u, gb = ln + 64 < d4 ? ln + 64 : 0u;
            const bool wide = d4 > 128;                                                             // four up to dim 1024 (block-uniform)
            const f32x4 qc = ln + 128 < d4 ? qs4[ln + 128] : z4, qd = ln + 192 < d4 ? qs4[ln + 192] : z4;
            const uint32_t gc = ln + 128 < d4 ? ln + 128 : 0u, gd = ln + 192 < d4 ? ln + 192 : 0u;
            const uint32_t n_groups = (n + 63) >> 6;
            for (uint32_t g = wv; g < n_groups; g += NW) {
                const uint32_t i = g * 64 + ln;
                const uint64_t key = i < n ? key_at(i) : KEY_NONE;
                const bool in = key != KEY_NONE && -order_key_inv((uint32_t)(key >> 32)) >= lo;
                const uint32_t myrow = (uint32_t)key;
                uint64_t mask = __builtin_amdgcn_ballot_w64(in);
                float s2 = 0.0f;
                while (mask) {                                   // wave-uniform
                    constexpr int R = 8;
                    int src[R];
                    uint32_t cnt_r = 0;
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        if (mask) { src[u] = __builtin_ctzll(mask); mask &= mask - 1; ++cnt_r; }
                        else src[u] = src[u > 0 ? u - 1 : 0];     // clamp: a harmless repeat of the last member
                    }
                    f32x4 va[R], vb[R];
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const uint32_t r_u = (uint32_t)__builtin_amdgcn_readlane((int)myrow, src[u]);
                        const f32x4 *rp = reinterpret_cast<const f32x4 *>(a.rows + (size_t)r_u * dim);
                        va[u] = rp[ga]; vb[u] = rp[gb];
                    }
                    float pp[R];
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        float p = qa.x * va[u].x;
                        p = __builtin_fmaf(qa.y, va[u].y, p); p = __builtin_fmaf(qa.z, va[u].z, p); p = __builtin_fmaf(qa.w, va[u].w, p);
                        p = __builtin_fmaf(qb.x, vb[u].x, p); p = __builtin_fmaf(qb.y, vb[u].y, p);
                        p = __builtin_fmaf(qb.z, vb[u].z, p); p = __builtin_fmaf(qb.w, vb[u].w, p);
                        pp[u] = p;
                    }
                    if (wide) {
#pragma unroll
                        for (int u = 0; u < R; ++u) {
                            const uint32_t r_u = (uint32_t)__builtin_amdgcn_readlane((int)myrow, src[u]);
                            const f32x4 *rp = reinterpret_cast<const f32x4 *>(a.rows + (size_t)r_u * dim);
                            va[u] = rp[gc]; vb[u] = rp[gd];
                        }
#pragma unroll
                        for (int u = 0; u < R; ++u) {
                            float p = pp[u];
                            p = __builtin_fmaf(qc.x, va[u].x, p); p = __builtin_fmaf(qc.y, va[u].y, p); p = __builtin_fmaf(qc.z, va[u].z, p); p = __builtin_fmaf(qc.w, va[u].w, p);
                            p = __builtin_fmaf(qd.x, vb[u].x, p); p = __builtin_fmaf(qd.y, vb[u].y, p); p = __builtin_fmaf(qd.z, vb[u].z, p); p = __builtin_fmaf(qd.w, vb[u].w, p);
                            pp[u] = p;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        float p = pp[u];
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
                        if ((uint32_t)u < cnt_r && (int)ln == src[u]) s2 = p;
                    }
                }
                if (i < n) {
                    const uint64_t k2 = in ? make_key(-s2, myrow) : KEY_NONE;
                    if (i < n_main) slots_q[i] = k2; else list[i - n_main] = k2;
                    if (i < n_st) { skey[i] = (uint32_t)(k2 >> 32); srow[i] = (uint32_t)k2; }
                }
            }
            __syncthreads();
            const float lo2 = window_floor(a.eps2[q]);
            nf = collect_window(lo2, false);
            n_sure = 0;                        // (a window this dense is scored as a whole)
            if (tid == 0) atomicAdd(a.stats + 3, 1u);
        }
        if (nf > a.fcap) bad = true;       // block-uniform: even the f32 window is too large (thousands of near-duplicates): exact scan
        if (!bad) {
            // pass B: exact reference-order scores of the window, then top-k by (dist, id).
            // The f32 rows of the window are cold in HBM, so every load of a chunk is put in flight at once
            // (coalesced float4) and only then reduced in the reference order:
            //   scalar-4: each float4 group is reduced straight out of its register to the group sum t_g
            //             (distance_inline.rs:165-168 inner expression) and parked in LDS; one thread per row then adds
            //             the dim/4 group sums sequentially, as the reference's `sum += t` does;
            //   AVX2:     raw rows are staged; eight FMA chains per row (:77-97), lanes 0..7 summed in order (:100-108).
            const uint32_t dim = a.dim, d4 = dim >> 2;
            const f32x4 *qs4 = reinterpret_cast<const f32x4 *>(qs);
            if (ORDER == SHODH_ORDER_SCALAR4) {
                float *tbuf = region;
                const uint32_t tp = d4 + 1;
                const uint32_t wv = tid >> 6, ln = tid & 63;
                for (uint32_t c0 = 0; c0 < nf; c0 += TC) {
                    const uint32_t nc = (nf - c0) < (uint32_t)TC ? (nf - c0) : (uint32_t)TC;
                    // wave w takes rows w, w+4, ...; its lanes take float4 groups ln, ln+64, ... (one row = one
                    // coalesced d4*16-byte read); FS_RW rows' loads are in flight per wave before any is reduced (the rows are
                    // cold in HBM: every round trip is ~2 us)
                    for (uint32_t cb = wv; cb < nc; cb += NW * FS_RW) {
                        for (uint32_t g0 = 0; g0 < d4; g0 += 128) {
                            f32x4 r[FS_RW][2];
#pragma unroll
                            for (int u = 0; u < FS_RW; ++u) {
                                // unconditional loads on clamped indices (see the staging loop above): all FS_RW x 2 in flight
                                const uint32_t c = cb + NW * u < nc ? cb + NW * u : nc - 1;
                                const float *rp_ = a.rows + (size_t)flist[c0 + c] * dim;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const uint32_t g = g0 + h * 64 + ln;
                                    r[u][h] = *reinterpret_cast<const f32x4 *>(rp_ + (g < d4 ? g : d4 - 1) * 4);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < FS_RW; ++u) {
                                const uint32_t c = cb + NW * u;
                                if (c < nc) {
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        const uint32_t g = g0 + h * 64 + ln;
                                        if (g < d4) {
                                            const f32x4 w = qs4[g];
                                            float t = w.x * r[u][h].x;
                                            t = t + w.y * r[u][h].y;
                                            t = t + w.z * r[u][h].z;
                                            t = t + w.w * r[u][h].w;
                                            tbuf[c * tp + g] = t;
                                        }
                                    }
                                }
                            }
                        }
                    }
                    __syncthreads();
                    if (tid < nc) {
                        float sum = 0.0f;
                        const float *tr = tbuf + tid * tp;
                        uint32_t g = 0;
                        for (; g + 8 <= d4; g += 8) {
                            float t8[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) t8[u] = tr[g + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) sum = sum + t8[u];
                        }
                        for (; g < d4; ++g) sum = sum + tr[g];
                        for (uint32_t j = d4 * 4; j < dim; ++j) sum = sum + qs[j] * a.rows[(size_t)flist[c0 + tid] * dim + j];
                        ekeys[c0 + tid] = make_key(-sum, a.id_base + flist[c0 + tid]);
                    }
                    __syncthreads();
                }
            } else {
                const uint32_t rp = dim + 8;
                float *rowbuf = region;
                f32x4 *rowbuf4 = reinterpret_cast<f32x4 *>(rowbuf);
                const uint32_t ch = a.ch_rows;
                float *tbuf = rowbuf + (size_t)ch * rp;
                for (uint32_t c0 = 0; c0 < nf; c0 += ch) {
                    const uint32_t nc = (nf - c0) < ch ? (nf - c0) : ch;
                    for (uint32_t e = tid; e < nc * d4; e += NT) {
                        const uint32_t c = e / d4, j = e % d4;
                        rowbuf4[c * (rp >> 2) + j] = *reinterpret_cast<const f32x4 *>(a.rows + (size_t)flist[c0 + c] * dim + j * 4);
                    }
                    __syncthreads();
                    if (ORDER == SHODH_ORDER_SEQ_1M) {
                        // SpannIndex::compute_distance, NormalizedDotProduct arm (spann.rs:562-571): `1 - sum(x*y)`, the sum
                        // strictly in index order from 0.0 (one product, one add per element; no FMA). One thread per row.
                        if (tid < nc) {
                            const float *rr = rowbuf + tid * rp;
                            float dot = 0.0f;
                            for (uint32_t i = 0; i < dim; i += 8) {
                                float pr[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) pr[u] = qs[i + u] * rr[i + u];
#pragma unroll
                                for (int u = 0; u < 8; ++u) dot = dot + pr[u];
                            }
                            ekeys[c0 + tid] = make_key(1.0f - dot, a.id_base + flist[c0 + tid]);
                        }
                    } else {
                        if (tid < nc * 8) {
                            const uint32_t c = tid >> 3, l = tid & 7;
                            float acc = 0.0f;
                            for (uint32_t i = 0; i < dim; i += 8) acc = __builtin_fmaf(qs[i + l], rowbuf[c * rp + i + l], acc);
                            tbuf[c * 8 + l] = acc;
                        }
                        __syncthreads();
                        if (tid < nc) {
                            const float *p8 = tbuf + tid * 8;
                            float r = p8[0] + p8[1];
                            r = r + p8[2]; r = r + p8[3]; r = r + p8[4]; r = r + p8[5]; r = r + p8[6]; r = r + p8[7];
                            ekeys[c0 + tid] = make_key(-r, a.id_base + flist[c0 + tid]);
                        }
                    }
                    __syncthreads();
                }
            }
            PROF_T(3)
            // (set mode: the exact keys of the window in ekeys[0 .. nf), the sure members' placeholder keys -- all smaller -- at the end of ekeys)
            auto key_b = [&](uint64_t i) -> uint64_t { return i < nf ? ekeys[i] : ekeys[a.fcap - 1u - (uint32_t)(i - nf)]; };
            const uint32_t m = block_select_topk<NT, true>(key_b, nf + n_sure, buf, mins);      // (one key per row of the window: unique)
            PROF_T(4)
#ifdef SHODH_PROF
            if (tid == 0 && (q % 37) == 0) printf("final q %u n %u nf %u | load-q %lld selectA %lld window %lld rescore %lld selectB %lld\n", q, n, nf, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4]);
#endif
            for (uint32_t i = tid; i < a.k; i += NT) {
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
                atomicAdd(a.stats + 0, *ecnt);
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
    if (a.stats_mirror && tid == 0) {         // (thread 0 made every update of this workgroup itself)
        bool last = gridDim.x == 1;
        if (!last) {
            // several queries, results written straight into host memory (no copy command behind the kernel): the LAST workgroup to finish mirrors the four
            // statistics words -- word 4 counts arrivals (zeroed with the others by convert_queries_kernel)
            __threadfence();
            last = atomicAdd(a.stats + 4, 1u) == gridDim.x - 1u;
            if (last) __threadfence();
        }
        if (last) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a.stats_mirror[i] = __atomic_load_n(a.stats + i, __ATOMIC_RELAXED);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------
struct MfmaPlan {
    uint32_t passes, n_slots, ksteps;
    uint32_t tile_stride, n_sel_tiles, J;
    uint64_t n_tiles;
    uint32_t cand_cap, fcap, topk_cap;
    int grid_x;
    uint32_t set_only;      // the caller needs the SET of the k nearest, not their exact distances or order (IVF probe selection): final_stage_kernel, "set mode"
};

uint32_t topk_capacity(uint32_t k);   // flat_exact.hip

bool mfma_supported(uint32_t dim) { return dim == 128 || dim == 256 || dim == 384 || dim == 512 || dim == 768 || dim == 1024; }   // (768 / 1024: mfma_scan_big_kernel; final_stage_kernel holds the query in four loads per thread: dim <= 1024)

MfmaPlan mfma_plan(uint64_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int cus) {
    MfmaPlan p{};
    p.passes = (uint32_t)ceil_div(nq, MF_BPAD);
    p.n_slots = p.passes * MF_BPAD;
    p.ksteps = dim / 16;
    p.n_tiles = ceil_div(n_rows, MF_TR);
    // sample: every S-th tile. The sample costs ~6 + 13*(16/S) us and leaves ~16*S*k/16 survivors per query to emit and
    // re-score, so S shrinks with k: 16 up to k = 40 (k = 10: S = 8/16/32 measured 277/271/272 us per step; k = 40: 8/12/16
    // all 302), 16*sqrt(40/k) above (k = 120: S = 4/6/8/10/12/16 measured 375/364/354/350/356/400 us). SHODH_SAMPLE_STRIDE
    // overrides.
    // Round 6: with the bound that tightens during the emit scan (MfmaArgs::dcnt: dim <= 384) the sample only has to START the bound, and for k <= 40 every
    // 32nd tile does (k = 10, stride 16 / 24 / 32 / 48 / 64: 264.7 / 261.6 / 260.6 / 260.5 / 261.3 us per step; k = 120 keeps its 9: 345 against 352 at 24).
    static const uint32_t stride_env = getenv("SHODH_SAMPLE_STRIDE") ? (uint32_t)atoi(getenv("SHODH_SAMPLE_STRIDE")) : 0u;
    static const bool dyn_env = !(getenv("SHODH_DYN_THR") && atoi(getenv("SHODH_DYN_THR")) == 0);
    // Final library of round 6 (measured error bound, cheaper threshold kernel), k = 40 -- `search_ids` with limit 10 asks for k_index = 4 x limit -- stride 32 / 28 / 24 / 20 / 16:
    // 0.2585 / 0.2560 / 0.2538 / 0.2538 / 0.2564 ms per step on the contract corpus, 263.9 / - / 259.9 / - / - us on a random one: every 24th tile from k = 17 to 40
    // (k = 10 keeps 32: 0.2301 against 0.2358 at 24; k = 120 keeps 9: 0.336 against 0.338 - 0.346 at 16 - 24 on the contract corpus, the reverse by 2 % on a clustered one).
    uint32_t sample_stride = (dyn_env && dim <= 384) ? (k > 16 ? 24 : 32) : 16;
    if (k > 40) {
        sample_stride = (uint32_t)(16.0 * __builtin_sqrt(40.0 / (double)k) + 0.5);
        if (sample_stride < 2) sample_stride = 2;
    }
    if (stride_env) sample_stride = stride_env;
    uint64_t want = p.n_tiles / (sample_stride ? sample_stride : 16u);
    const uint64_t min_tiles = (uint64_t)k * 2 + 64;
    if (want < min_tiles) want = min_tiles;
    if (want > p.n_tiles) want = p.n_tiles;
    p.tile_stride = (uint32_t)(p.n_tiles / want);
    if (p.tile_stride < 1) p.tile_stride = 1;
    p.n_sel_tiles = (uint32_t)ceil_div(p.n_tiles, p.tile_stride);
    p.J = p.n_sel_tiles;
    // shared overflow list per query. It used to be 192 k (>= 4096) entries, and a query that outgrew it went to the exact
    // scan of the whole corpus (a 24x cliff on dense corpora). Now it takes what 512 MiB of workspace allow, up to 65536
    // entries per query (128 MiB at 256 queries): whatever lands in it is narrowed by the final stage's level-2 f32 filter at a
    // cost proportional to the entries. Calls with thousands of queries (k-means assignment) keep the old size.
    uint32_t cc = 192u * (k ? k : 1);
    if (cc < 4096) cc = 4096;
    cc = next_pow2(cc);
    uint64_t room = (512ull << 20) / ((uint64_t)p.passes * MF_BPAD * 8);
    uint32_t big = 65536;
    while (big > cc && big > room) big >>= 1;
    p.cand_cap = big > cc ? big : cc;
    uint32_t fc = 4u * (k ? k : 1);
    if (fc < 2048) fc = 2048;
    p.fcap = next_pow2(fc);
    p.topk_cap = topk_capacity(k);
    p.grid_x = dim > 512 ? (cus >= 2 ? cus / 2 : 1) : cus;      // big dimensions: two workgroups per pass and tile sequence (mfma_scan_big_kernel)
    return p;
}

struct MfmaWorkspace {
    _Float16 *q_h; float *qnorm; float *thr; float *eps; uint32_t *cand_cnt; uint32_t *fallback; uint32_t *fb_list;
    uint32_t *fb_count; uint32_t *stats; float *blockmax; uint64_t *cand; uint64_t *slots; float *eps2;
    uint32_t *dcnt; float *dthr; uint32_t *dpub; uint32_t *dpar;
    uint32_t *stats_mirror = nullptr;
};

size_t mfma_workspace_bytes(const MfmaPlan &p, uint32_t dim, size_t *offs /*[MFMA_WS_PARTS = 16]*/) {
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
    offs[10] = take((size_t)p.n_slots * p.cand_cap * 8);   // cand (shared overflow lists)
    offs[11] = take((size_t)p.n_slots * p.grid_x * MF_SLOTS * 8);   // slots (private per query and workgroup)
    offs[12] = take((size_t)p.n_slots * 4);              // eps2
    // dynamic threshold: counters and thresholds of the levels, then {K_B, step} and the published bound per query (one block: offs[] has 16 entries)
    offs[13] = take((size_t)p.n_slots * MF_DYN_NB * 4 * (MF_DYN_REP + 1));
    offs[14] = take((size_t)p.n_slots * 12);
    offs[15] = o;
    return o;
}

template <int MODE>
// ev0 / ev1 (optional): start / stop timestamps of THIS dispatch (hipExtLaunchKernel attaches them to the kernel's own packet;
// separate hipEventRecord calls are extra packets in the stream and cost 3-5 us each between two kernels)
static int launch_scan(const MfmaArgs &a, const MfmaPlan &p, uint32_t n_sel, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    if (a.dim > 512) {
        if constexpr (MODE == MF_MODE_SCORES) { set_error("MFMA scan: score output needs dim <= 512"); return SHODH_ERR_UNSUPPORTED; } else {
        dim3 grid((uint32_t)p.grid_x, p.passes * 2);
        if ((uint32_t)p.grid_x > n_sel) grid.x = n_sel ? n_sel : 1;
        static const int big_v1 = getenv("SHODH_BIG_SCAN_V1") ? atoi(getenv("SHODH_BIG_SCAN_V1")) : 0;      // diagnostic: 1 = the round-2 kernel (one wave per SIMD)
        if (!big_v1) {
            const size_t half = (size_t)MFB_TR * a.dim * 2;
            const size_t nb = (3 * half + (size_t)MF_EQ_CAP * 12 + MF_BPAD * 4 + 64 + 64 <= 160 * 1024) ? 3 : 2;      // == big3_nbuf<KSTEPS>()
            const size_t lds3 = nb * half + (size_t)MF_EQ_CAP * 12 + MF_BPAD * 4 + 64 + 32;
#define SHODH_LAUNCH_BIG3(KS)                                                                                       \
    case KS:                                                                                                       \
        SHODH_TRY(ensure_dynamic_lds((const void *)mfma_scan_big3_kernel<MODE, KS>, lds3));                         \
        if (ev0 && ev1) hipExtLaunchKernelGGL((mfma_scan_big3_kernel<MODE, KS>), grid, dim3(512), (uint32_t)lds3, st, ev0, ev1, 0u, a);  \
        else hipLaunchKernelGGL((mfma_scan_big3_kernel<MODE, KS>), grid, dim3(512), lds3, st, a);                   \
        break;
            switch (p.ksteps) {
                SHODH_LAUNCH_BIG3(48)
                SHODH_LAUNCH_BIG3(64)
                default: set_error("MFMA scan: unsupported dim %u", a.dim); return SHODH_ERR_UNSUPPORTED;
            }
#undef SHODH_LAUNCH_BIG3
            SHODH_HIP_TRY(hipGetLastError());
            return SHODH_OK;
        }
        const size_t lds = 2ull * MFB_TR * a.dim * 2 + (size_t)MF_EQ_CAP * 12 + MF_BPAD * 4 + 32;
#define SHODH_LAUNCH_BIG(KS)                                                                                        \
    case KS:                                                                                                       \
        SHODH_TRY(ensure_dynamic_lds((const void *)mfma_scan_big_kernel<MODE, KS>, lds));                           \
        if (ev0 && ev1) hipExtLaunchKernelGGL((mfma_scan_big_kernel<MODE, KS>), grid, dim3(256), (uint32_t)lds, st, ev0, ev1, 0u, a);  \
        else hipLaunchKernelGGL((mfma_scan_big_kernel<MODE, KS>), grid, dim3(256), lds, st, a);                     \
        break;
        switch (p.ksteps) {
            SHODH_LAUNCH_BIG(48)
            SHODH_LAUNCH_BIG(64)
            default: set_error("MFMA scan: unsupported dim %u", a.dim); return SHODH_ERR_UNSUPPORTED;
        }
#undef SHODH_LAUNCH_BIG
        SHODH_HIP_TRY(hipGetLastError());
        return SHODH_OK;
        }
    }
    const size_t nbuf = (3ull * MF_TR * a.dim * 2 + (size_t)MF_EQ_CAP * 12 + MF_BPAD * 4 + MF_BPAD * 12 + 256 + 64 + 64 <= 160 * 1024) ? 3 : 2;      // == mfma_scan_nbuf<KSTEPS>()
    const size_t lds = nbuf * MF_TR * a.dim * 2 + (size_t)MF_EQ_CAP * 12 + MF_BPAD * 4 + MF_BPAD * 12 + 256 + 64 + 32;
    dim3 grid((uint32_t)p.grid_x, p.passes);
    if ((uint32_t)p.grid_x > n_sel) grid.x = n_sel ? n_sel : 1;
#define SHODH_LAUNCH_KS(KS)                                                                                        \
    case KS:                                                                                                       \
        SHODH_TRY(ensure_dynamic_lds((const void *)mfma_scan_kernel<MODE, KS>, lds));                               \
        if (ev0 && ev1) hipExtLaunchKernelGGL((mfma_scan_kernel<MODE, KS>), grid, dim3(512), (uint32_t)lds, st, ev0, ev1, 0u, a);  \
        else hipLaunchKernelGGL((mfma_scan_kernel<MODE, KS>), grid, dim3(512), lds, st, a);                         \
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

// one workgroup per query: k-th best approximate score, window, reference-order re-score, sort, write (final_stage_kernel)
static int launch_final_stage(const float *rows, uint32_t dim, const float *d_q, uint32_t nq, uint32_t k, uint32_t order, uint32_t id_base,
                              const MfmaPlan &p, const MfmaWorkspace &w, uint32_t nb_emit, const uint32_t *cand_cnt, uint32_t *cnt_reset,
                              uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st) {
    // LDS of the final stage. One workgroup per query: with a few hundred queries every workgroup has a CU to itself and the
    // generous sizes cost nothing; with thousands of queries (nearest-centroid searches of k-means / IVF encoding) the
    // workgroups per CU are what counts, so the re-score list, the key staging and the row staging shrink to what such
    // searches need (a query that outgrows them goes to the exact scan like any other overflow).
    uint32_t fcap = p.fcap, stage_cap = (uint32_t)FS_STAGE, ch_rows = (uint32_t)FS_CH;
    if (nq >= 1024) {
        fcap = next_pow2(4u * (k ? k : 1) > 256u ? 4u * (k ? k : 1) : 256u);
        if (fcap > p.fcap) fcap = p.fcap;
        stage_cap = next_pow2(nb_emit * (uint32_t)MF_SLOTS + 256u);
        if (stage_cap > (uint32_t)FS_STAGE) stage_cap = (uint32_t)FS_STAGE;
        ch_rows = next_pow2(k ? k : 1);
        ch_rows = ch_rows < 4u ? 4u : (ch_rows > (uint32_t)FS_CH ? (uint32_t)FS_CH : ch_rows);
    }
    // Up to a few hundred queries every workgroup has a CU to itself: 512 threads (twice the re-score rows in flight and twice the lanes in the
    // selections) where its LDS fits; thousands of queries keep 256 so that several workgroups share a CU, and so do shapes whose 512-thread
    // layout would not fit 160 KiB (768 / 1024 dimensions). SHODH_FINAL_NT=256 forces the small form (speed only).
    static const int nt_env = getenv("SHODH_FINAL_NT") ? atoi(getenv("SHODH_FINAL_NT")) : 0;
    uint32_t nt = (nt_env == 256 || nt_env == 512) ? (uint32_t)nt_env : (nq >= 1024 ? 256u : 512u);
    uint32_t tcap = 0;
    size_t flds = 0;
    const uint32_t stage_cap0 = stage_cap, fcap0 = fcap;
    for (;;) {
        // the selection buffer takes up to 2 * nt pushes between two compaction checks on top of the k keys it keeps (topk.h: a push past the capacity is dropped)
        tcap = k + 2 * nt;
        if (tcap < 2 * k) tcap = 2 * k;
        if (tcap < 1024) tcap = 1024;
        tcap = next_pow2(tcap);
        // qs[dim] | keys[tcap] | mins[2 nt] | ekeys[fcap] | thr | flist[fcap] | cnt, fcnt | region | sel32 | skey, srow [stage_cap]   (every part a multiple of 8 B; region at 16 B)
        flds = (size_t)dim * 4 + (size_t)tcap * 8 + (size_t)2 * nt * 8 + (size_t)fcap * 8 + 8 + (size_t)fcap * 4 + 8 +
               final_stage_region_bytes(dim, order, ch_rows, nt) + (size_t)KTH_SCRATCH_U32 * 4 + (size_t)stage_cap * 8 + 16;
        if (nt == 512 && flds > 160 * 1024) {
            // 768 dimensions: the 512-thread layout misses 160 KiB by the generous key staging and re-score list of small batches -- halve them first
            // (a query that outgrows them re-reads keys from global memory or takes the level-2 filter, like any other overflow): 36 -> 20 us per step
            if (stage_cap > 1024 && k <= 128) { stage_cap = 1024; continue; }
            if (fcap > 1024 && 4u * k <= 1024u) { fcap = 1024; continue; }
            nt = 256; stage_cap = stage_cap0; fcap = fcap0; continue;
        }
        break;
    }
    FinalArgs f{rows, dim, d_q, nq, k, tcap, w.slots, nb_emit, w.cand, cand_cnt, p.cand_cap, w.eps, w.eps2, fcap, stage_cap, ch_rows, order, id_base,
                w.fallback, w.fb_list, w.fb_count, d_ids, d_dist, d_counts, w.stats, cnt_reset, w.stats_mirror,
                (p.set_only && order == SHODH_ORDER_SEQ_1M && k >= 1 && k <= 128) ? 1u : 0u};
#define SHODH_LAUNCH_FINAL(ORD, NTV)                                                                         \
    do {                                                                                                     \
        SHODH_TRY(ensure_dynamic_lds((const void *)final_stage_kernel<ORD, NTV>, flds));                     \
        hipLaunchKernelGGL((final_stage_kernel<ORD, NTV>), dim3(nq), dim3(NTV), flds, st, f);                \
    } while (0)
    if (order == SHODH_ORDER_SEQ_1M) { if (nt == 512) SHODH_LAUNCH_FINAL(SHODH_ORDER_SEQ_1M, 512); else SHODH_LAUNCH_FINAL(SHODH_ORDER_SEQ_1M, 256); }
    else if (order == SHODH_ORDER_AVX2) { if (nt == 512) SHODH_LAUNCH_FINAL(SHODH_ORDER_AVX2, 512); else SHODH_LAUNCH_FINAL(SHODH_ORDER_AVX2, 256); }
    else { if (nt == 512) SHODH_LAUNCH_FINAL(SHODH_ORDER_SCALAR4, 512); else SHODH_LAUNCH_FINAL(SHODH_ORDER_SCALAR4, 256); }
#undef SHODH_LAUNCH_FINAL
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}


// ---- single-query scan ------------------------------------------------------------------------------------------------
// `recall` asks the index one query at a time (mod.rs:2875-2904): a batch-shaped pipeline spends a third of such a call on launches
// (query conversion, sampled scan, threshold, emit scan, final stage). For ONE query the scores of a workgroup's slice of the corpus fit
// in LDS, so a single pass does it all: every workgroup streams its slice of the fp16 shadow (a row = 16 lanes, f32 query in registers,
// f32 FMA: the error bound of the MFMA scan without the query's rounding), keeps the order keys of the scores in LDS, selects its LOCAL
// k-th best and hands every live row within 2 eps of it to the final stage. The global k-th best is at least the local one, so the union
// of the local windows contains the global window {s~ >= kth(s~) - 2 eps}: the same proof, the same final stage, the same results.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int SOLO_NT = 1024;
constexpr int SOLO_MAX_SLICE = 16384;      // rows per workgroup (their keys: 64 KiB of LDS)
constexpr int SOLO_EM = 1024;              // survivors of a workgroup staged in LDS before they are handed over with ONE global atomic
constexpr uint32_t SOLO_MAX_K = 32;        // local thresholds: nb * k candidates reach the final stage. Above it (recall's own k = 120, retrieval.rs:927):
// GLOBAL threshold mode -- the scan writes every row's order key to HBM (4 bytes per row) and each wave its best key; a second, tiny
// kernel takes the k-th best of those nb * 16 wave maxima (scores of distinct rows, so a lower bound of the true k-th best, exactly
// the argument of the batch pipeline's sampled tile maxima -- only that here the maxima cover the whole corpus and the bound is nearly
// tight) and hands over the rows within 2 eps of it: about k of them instead of nb * k.
constexpr uint32_t SOLO_MAX_K_GLOBAL = 2048;
constexpr int SOLO_WAVES = 16;             // waves per workgroup of the scan = maxima per workgroup
struct SoloArgs {
    const _Float16 *rows_h; uint64_t n_rows; uint32_t dim;
    const float *q;            // [dim]
    const uint32_t *deleted;
    uint32_t k, slice;         // slice: rows per workgroup
    float eps_rel_maxnorm, eps_abs_a, maxnorm, eps2_rel_maxnorm, eps2_abs_a;
    uint64_t *slots;           // [gridDim.x][MF_SLOTS]
    uint64_t *cand; uint32_t *cand_cnt; uint32_t cand_cap;      // cand_cnt: zero on entry (the final stage hands it back zeroed)
    float *eps, *eps2; uint32_t *fallback, *fb_count, *stats;   // written by workgroup 0 for the final stage
    uint32_t ablate;           // diagnostic builds (-DSHODH_DIAG): 1 = no selection / hand-over, 2 = no arithmetic (results invalid)
    uint32_t *skeys;           // global threshold mode: [n_rows] order keys of the scores (0xFFFFFFFF = tombstoned)
    uint32_t *tops;            // global threshold mode: [gridDim.x][SOLO_WAVES] best key of each wave
    uint32_t ilv;              // 1: workgroup b takes the row blocks b, b + grid, b + 2 grid, ... (one block = the 64 U rows of an iteration) instead of a contiguous slice
    uint32_t stream;           // 1 (384-d, implies ilv): the rows arrive by LDS-DMA through a ring of 32-row tiles at smem + ring_off, followed by dwl_cap tombstone words and 64 keys
    uint32_t ring_off, dwl_cap;
};

template <int DIM, bool GLOBAL_THR = false>
__global__ __launch_bounds__(SOLO_NT) void solo_scan_kernel(SoloArgs a) {
    constexpr int EPL = DIM / 16;          // halfs per lane
    constexpr int LPL = EPL / 8;           // 16-byte loads per lane and row
    constexpr int U = EPL <= 24 ? 4 : (EPL <= 32 ? 2 : 1);   // groups of 4 rows in flight per wave (12 KiB per wave and 16 waves per CU at 384-d)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *skey = reinterpret_cast<uint32_t *>(smem);                       // [slice] order_key(-score), 0xFFFFFFFF = tombstoned
    uint64_t *em = reinterpret_cast<uint64_t *>(skey + a.slice + (a.slice & 1u));   // [SOLO_EM]
    uint32_t *scratch = reinterpret_cast<uint32_t *>(em + SOLO_EM);            // [KTH_SCRATCH_U32]
    uint32_t *ctl = scratch + KTH_SCRATCH_U32;                                 // [2]
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sr = lane >> 4, seg = lane & 15;
    // the query: this lane's EPL values -- the 16-byte chunks seg, 16 + seg, ... of a row, so that the 16 lanes of a row read 256
    // contiguous bytes per load instruction (whole cache lines; one lane = one contiguous piece of the row made every line the business
    // of three instructions); norm, largest magnitude and finiteness over the 16 lanes of a row group
    float qv[EPL];
    float ss = 0.0f, mx = 0.0f;
    uint32_t badv = 0;
#pragma unroll
    for (int c = 0; c < LPL; ++c) {
        const float4 v0 = *reinterpret_cast<const float4 *>(a.q + (c * 16 + seg) * 8), v1 = *reinterpret_cast<const float4 *>(a.q + (c * 16 + seg) * 8 + 4);
        qv[c * 8] = v0.x; qv[c * 8 + 1] = v0.y; qv[c * 8 + 2] = v0.z; qv[c * 8 + 3] = v0.w;
        qv[c * 8 + 4] = v1.x; qv[c * 8 + 5] = v1.y; qv[c * 8 + 6] = v1.z; qv[c * 8 + 7] = v1.w;
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        if (!(__builtin_fabsf(qv[j]) <= 3.0e38f)) badv++;
        mx = fmaxf(mx, __builtin_fabsf(qv[j]));
        ss = __builtin_fmaf(qv[j], qv[j], ss);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
        badv += __shfl_xor(badv, off);
    }
    const float qn = __builtin_sqrtf(ss) * 1.00001f;
    const bool bad_q = badv != 0 || mx * MF_SCALE > 60000.0f;       // the conditions of convert_queries_kernel
    const float eps = a.eps_rel_maxnorm * qn + a.eps_abs_a * (qn + a.maxnorm) + 1e-9f;      // (the query is not rounded here: no residual term of its own)
    if (blockIdx.x == 0) {
        if (tid < 4) a.stats[tid] = 0;
        if (tid == 4) *a.fb_count = 0;
        if (tid == 5) { a.eps[0] = eps; a.eps2[0] = a.eps2_rel_maxnorm * qn + a.eps2_abs_a * (qn + a.maxnorm) + 1e-9f; a.fallback[0] = bad_q ? 1u : 0u; }
    }
    uint64_t *my_slots = a.slots + (size_t)blockIdx.x * MF_SLOTS;
    // Which rows are this workgroup's. Contiguous slices (round 2) make 256 streams megabytes apart; with the blocks of an iteration dealt round-robin
    // (a.ilv) the workgroups walk the corpus side by side -- at any moment the whole chip reads one ~50 MB window front to back. Measured on the bare stream
    // (tools/ubench/solo_stream.hip): 6.40 against 6.0 TB/s, whoever issues the loads. Local index i <-> global row grow(i); everything downstream of the
    // scores (keys in LDS, k-th best, hand-over) works on local indices and only names a row through grow().
    constexpr bool CAN_STREAM = DIM == 384;
    const bool stream = CAN_STREAM && a.stream != 0;      // uniform: the rows arrive through an LDS ring (below); implies ilv, in blocks of one 32-row tile
    const uint32_t ISH = stream ? 5u : (U == 4 ? 8u : (U == 2 ? 7u : 6u)), IBLK = 1u << ISH;      // rows per dealt block
    const bool ilv = a.ilv != 0;             // uniform
    const uint64_t row0 = ilv ? 0 : (uint64_t)blockIdx.x * a.slice;
    uint32_t n_loc;
    if (ilv) {
        const uint64_t nblk = (a.n_rows + IBLK - 1) / IBLK;
        const uint64_t mine = blockIdx.x < nblk ? (nblk - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        n_loc = (uint32_t)(mine * IBLK);
        if (mine && (nblk - 1) % gridDim.x == blockIdx.x) n_loc -= (uint32_t)(nblk * IBLK - a.n_rows);      // the corpus' last, partial block is this workgroup's last
    } else {
        n_loc = row0 >= a.n_rows ? 0u : (a.n_rows - row0 < a.slice ? (uint32_t)(a.n_rows - row0) : a.slice);
    }
    auto grow = [&](uint32_t i) -> uint64_t { return ilv ? (((uint64_t)(i >> ISH) * gridDim.x + blockIdx.x) << ISH) + (i & (IBLK - 1u)) : row0 + i; };
    if (bad_q || n_loc == 0) {          // block-uniform
        if (GLOBAL_THR) { if (lane == 0) a.tops[blockIdx.x * SOLO_WAVES + wave] = 0xFFFFFFFFu; }
        else if (tid < (uint32_t)MF_SLOTS) my_slots[tid] = KEY_NONE;
        return;
    }
    uint32_t best = 0xFFFFFFFFu;        // (GLOBAL_THR) smallest key = best score among this lane's rows
    // ---- scores of the slice ------------------------------------------------------------------------------------------
    const _Float16 *base = a.rows_h + (size_t)seg * 8;
    if (stream) {
        // ---- the slice as a stream (round 6) ---------------------------------------------------------------------------
        // Sixteen waves loading their rows straight into registers reach 5.6 - 5.9 TB/s whatever the lane layout (round 2); LDS-DMA of whole tiles, issued by
        // few waves, with the tiles dealt round-robin over the workgroups reaches 6.4 (tools/ubench/solo_stream.hip, dma_issue.hip: the fewer waves issue, the
        // better the stream). So: 32-row tiles (24 KiB) into a ring of five, waves 0 and 1 issue nothing else (twelve 1-KiB pieces each per tile, 48 in
        // flight per wave: vmcnt has six bits), ONE s_barrier per tile, and the other fourteen waves take the tile's eight 4-row groups in rotation -- the
        // same 16 lanes per row, the same chunks and the same FMA order as below, read from LDS. The tombstone words of the workgroup's tiles (one word per
        // tile: a tile is 32 aligned rows) are fetched once, up front; in global-threshold mode a tile's 32 keys leave as ONE 128-byte store a tile later.
        constexpr uint32_t TR = 32, TILE = TR * DIM * 2, NBUF = 5, PER = TILE / 1024 / 2;
        unsigned char *ring = smem + a.ring_off;
        uint32_t *dwl = reinterpret_cast<uint32_t *>(ring + NBUF * TILE);      // [dwl_cap] tombstone words of this workgroup's tiles
        uint32_t *tkeys = dwl + a.dwl_cap;                                      // [2][32] global-threshold mode: the keys of the last two tiles
        const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)ring;
        const uint32_t n_tiles = (n_loc + TR - 1) / TR;
        const unsigned char *rows_b = reinterpret_cast<const unsigned char *>(a.rows_h);
        auto issue = [&](uint32_t j) {      // this wave's half of tile j of the workgroup (global tile j * grid + block) into buffer j % NBUF
            const unsigned char *src = uniform_ptr(rows_b + ((size_t)j * gridDim.x + blockIdx.x) * TILE);
            const uint32_t dst = ring_lds + (j % NBUF) * TILE;
#pragma unroll
            for (uint32_t i = 0; i < PER; ++i) {
                const uint32_t n = wave * PER + i;
                glds16(src, n * 1024u + lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + n * 1024u)));
            }
        };
        if (wave < 2) {
            for (uint32_t j = 0; j < NBUF - 1 && j < n_tiles; ++j) issue(j);
        } else {
            for (uint32_t j = tid - 128; j < n_tiles; j += SOLO_NT - 128) dwl[j] = a.deleted ? a.deleted[(size_t)j * gridDim.x + blockIdx.x] : 0u;
        }
        for (uint32_t t = 0; t < n_tiles; ++t) {
            if (wave < 2) {      // tile t has landed: only the pieces of the (up to three) tiles issued after it may still be in flight
                const uint32_t after = n_tiles - 1 - t < NBUF - 2 ? n_tiles - 1 - t : NBUF - 2;
                if (after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * PER) : "memory");
                else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PER) : "memory");
                else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(1 * PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();      // tile t is there for everybody; everybody is through with tile t - 1 (whose buffer the next issue refills) and its keys
            if (wave < 2) {
                if (t + NBUF - 1 < n_tiles) issue(t + NBUF - 1);
            } else {
                const uint32_t g = (wave - 2u + 14u - (t * 8u) % 14u) % 14u;      // eight groups of four rows over fourteen waves, in rotation
                if (g < 8u) {
                    const uint32_t rt = g * 4 + sr, rl = t * TR + rt;
                    const unsigned char *rp = ring + (t % NBUF) * TILE + rt * (DIM * 2) + seg * 16;
                    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                    for (int c = 0; c < LPL; ++c) {
                        const u32x4 hc = *reinterpret_cast<const u32x4 *>(rp + c * 256);
                        const _Float16 *hv = reinterpret_cast<const _Float16 *>(&hc);
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            p0 = __builtin_fmaf(qv[c * 8 + e], (float)hv[e], p0);
                            p1 = __builtin_fmaf(qv[c * 8 + e + 1], (float)hv[e + 1], p1);
                        }
                    }
                    float pr = p0 + p1;
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) pr += __shfl_xor(pr, off);
                    const bool del = (dwl[t] >> rt) & 1u;
                    const uint32_t key = (del || rl >= n_loc) ? 0xFFFFFFFFu : order_key(-(pr * (1.0f / MF_SCALE)));
                    if (GLOBAL_THR) {
                        best = key < best ? key : best;
                        if (seg == 0) tkeys[(t & 1u) * TR + rt] = key;
                    } else if (seg == 0 && rl < n_loc) skey[rl] = key;
                }
                if (GLOBAL_THR && wave == 15 && t > 0 && lane < TR) {      // the tile before this one: its keys were complete at the barrier
                    const uint32_t i = (t - 1) * TR + lane;
                    if (i < n_loc) a.skeys[grow(i)] = tkeys[((t - 1) & 1u) * TR + lane];
                }
            }
        }
        __syncthreads();
        if (GLOBAL_THR && wave == 15 && n_tiles && lane < TR) {
            const uint32_t i = (n_tiles - 1) * TR + lane;
            if (i < n_loc) a.skeys[grow(i)] = tkeys[((n_tiles - 1) & 1u) * TR + lane];
        }
    } else
    for (uint32_t r0 = 0; r0 < n_loc; r0 += 64 * U) {
        u32x4 h[U][LPL];
        uint32_t dw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // local mode: group u of the iteration is 64 consecutive rows over the 16 waves; global mode: a wave takes 4 U consecutive rows,
            // so that their keys leave in one store instruction
            const uint32_t rl = GLOBAL_THR ? r0 + wave * (4 * U) + u * 4 + sr : r0 + u * 64 + wave * 4 + sr;
            const uint32_t rc = rl < n_loc ? rl : n_loc - 1;        // unconditional loads on clamped rows: all of them in flight at once
            const uint64_t gr = grow(rc);
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(base + (size_t)gr * DIM);
#pragma unroll
            for (int c = 0; c < LPL; ++c) h[u][c] = rp[c * 16];
            dw[u] = a.deleted ? a.deleted[gr >> 5] : 0u;
        }
        uint32_t keyu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t rl = GLOBAL_THR ? r0 + wave * (4 * U) + u * 4 + sr : r0 + u * 64 + wave * 4 + sr;
            float p0 = 0.0f, p1 = 0.0f;
            if (a.ablate & 2u) { p0 = __uint_as_float(h[u][0].x ^ h[u][LPL - 1].w); }
            else
#pragma unroll
            for (int c = 0; c < LPL; ++c) {
                const _Float16 *hv = reinterpret_cast<const _Float16 *>(&h[u][c]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    p0 = __builtin_fmaf(qv[c * 8 + e], (float)hv[e], p0);
                    p1 = __builtin_fmaf(qv[c * 8 + e + 1], (float)hv[e + 1], p1);
                }
            }
            float pr = p0 + p1;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) pr += __shfl_xor(pr, off);
            if (GLOBAL_THR) {           // (the butterfly left the sum in all 16 lanes of the row)
                const uint32_t row = (uint32_t)grow(rl < n_loc ? rl : n_loc - 1);
                const bool del = (dw[u] >> (row & 31)) & 1u;
                keyu[u] = (del || rl >= n_loc) ? 0xFFFFFFFFu : order_key(-(pr * (1.0f / MF_SCALE)));
                best = keyu[u] < best ? keyu[u] : best;
            } else if (seg == 0 && rl < n_loc) {
                const uint32_t row = (uint32_t)grow(rl);
                const bool del = (dw[u] >> (row & 31)) & 1u;
                skey[rl] = del ? 0xFFFFFFFFu : order_key(-(pr * (1.0f / MF_SCALE)));
            }
        }
        if (GLOBAL_THR) {
            // lane (sr, seg = u) stores the key of row 4 u + sr of the wave's block: 4 U dwords of one 16 U-byte segment in ONE instruction
            uint32_t mykey = keyu[0];
#pragma unroll
            for (int u = 1; u < U; ++u) mykey = (int)seg == u ? keyu[u] : mykey;
            const uint32_t rs = r0 + wave * (4 * U) + seg * 4 + sr;
            if (seg < (uint32_t)U && rs < n_loc) a.skeys[grow(rs)] = mykey;      // (16 U consecutive rows of one block: still one segment)
        }
    }
    if (GLOBAL_THR) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)best, off); best = o < best ? o : best; }
        if (lane == 0) a.tops[blockIdx.x * SOLO_WAVES + wave] = best;
        return;
    }
    if (tid == 0) ctl[0] = 0;
    __syncthreads();
    if (a.ablate & 1u) { if (tid < (uint32_t)MF_SLOTS) my_slots[tid] = KEY_NONE; return; }
    // ---- local k-th best, window, hand-over ---------------------------------------------------------------------------
    auto key_at = [&](uint32_t i) -> uint32_t { return skey[i]; };
    bool ovf = false;
    const uint32_t kk = block_kth_u32<SOLO_NT>(key_at, n_loc, a.k, scratch, &ovf);
    uint32_t klim = 0xFFFFFFFEu;           // fewer than k live rows in the slice: all of them
    if (kk != 0xFFFFFFFFu) {
        const float kth = -order_key_inv(kk);
        const float lo = kth - (2.001f * eps + 1e-7f * __builtin_fabsf(kth));
        klim = order_key(-lo);
    }
    for (uint32_t i = tid; i < n_loc; i += SOLO_NT) {
        const uint32_t key = skey[i];
        if (key != 0xFFFFFFFFu && key <= klim) {
            const uint64_t k64 = ((uint64_t)key << 32) | (uint32_t)grow(i);
            const uint32_t idx = atomicAdd(&ctl[0], 1u);
            if (idx < (uint32_t)SOLO_EM) em[idx] = k64;
            else {                           // a dense slice: straight to the shared list
                const uint32_t slot = atomicAdd(a.cand_cnt, 1u);
                if (slot < a.cand_cap) a.cand[slot] = k64;
            }
        }
    }
    __syncthreads();
    const uint32_t total = ctl[0] < (uint32_t)SOLO_EM ? ctl[0] : (uint32_t)SOLO_EM;
    const uint32_t extra = total > (uint32_t)MF_SLOTS ? total - MF_SLOTS : 0u;
    if (tid == 0) ctl[1] = extra ? atomicAdd(a.cand_cnt, extra) : 0u;
    __syncthreads();
    const uint32_t lbase = ctl[1];
    const uint32_t n_out = total > (uint32_t)MF_SLOTS ? total : (uint32_t)MF_SLOTS;
    for (uint32_t i = tid; i < n_out; i += SOLO_NT) {
        if (i < (uint32_t)MF_SLOTS) my_slots[i] = i < total ? em[i] : KEY_NONE;
        else if (lbase + (i - MF_SLOTS) < a.cand_cap) a.cand[lbase + (i - MF_SLOTS)] = em[i];
    }
}

// global threshold mode, second kernel: threshold from the wave maxima, then every workgroup hands over its slice's survivors
struct SoloEmitArgs {
    const uint32_t *skeys; const uint32_t *tops; uint32_t n_tops;
    uint64_t n_rows; uint32_t slice, k;
    const float *eps; const uint32_t *fallback;
    uint64_t *slots; uint64_t *cand; uint32_t *cand_cnt; uint32_t cand_cap;
};
__global__ __launch_bounds__(256) void solo_emit_kernel(SoloEmitArgs a) {
    __shared__ uint32_t scratch[KTH_SCRATCH_U32];
    __shared__ uint64_t em[SOLO_EM];
    __shared__ uint32_t ctl[2];
    const uint32_t tid = threadIdx.x;
    uint64_t *my_slots = a.slots + (size_t)blockIdx.x * MF_SLOTS;
    const uint64_t row0 = (uint64_t)blockIdx.x * a.slice;
    const uint32_t n_loc = row0 >= a.n_rows ? 0u : (a.n_rows - row0 < a.slice ? (uint32_t)(a.n_rows - row0) : a.slice);
    if (a.fallback[0] || n_loc == 0) {          // block-uniform
        if (tid < (uint32_t)MF_SLOTS) my_slots[tid] = KEY_NONE;
        return;
    }
    if (tid == 0) ctl[0] = 0;
    auto top_at = [&](uint32_t i) -> uint32_t { return a.tops[i]; };
    bool ovf = false;
    const uint32_t kk = block_kth_u32<256>(top_at, a.n_tops, a.k, scratch, &ovf);      // (starts with a barrier: ctl is visible after it)
    uint32_t klim = 0xFFFFFFFEu;           // fewer than k waves saw a live row: every live row goes on
    if (kk != 0xFFFFFFFFu) {
        const float kth = -order_key_inv(kk);
        const float lo = kth - (2.001f * a.eps[0] + 1e-7f * __builtin_fabsf(kth));
        klim = order_key(-lo);
    }
    for (uint32_t i0 = 0; i0 < n_loc; i0 += 1024) {          // four loads in flight per thread (clamped index)
        uint32_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + u * 256 + tid; kv[u] = a.skeys[row0 + (i < n_loc ? i : n_loc - 1)]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * 256 + tid;
            if (i < n_loc && kv[u] != 0xFFFFFFFFu && kv[u] <= klim) {
                const uint64_t k64 = ((uint64_t)kv[u] << 32) | (uint32_t)(row0 + i);
                const uint32_t idx = atomicAdd(&ctl[0], 1u);
                if (idx < (uint32_t)SOLO_EM) em[idx] = k64;
                else {
                    const uint32_t slot = atomicAdd(a.cand_cnt, 1u);
                    if (slot < a.cand_cap) a.cand[slot] = k64;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t total = ctl[0] < (uint32_t)SOLO_EM ? ctl[0] : (uint32_t)SOLO_EM;
    const uint32_t extra = total > (uint32_t)MF_SLOTS ? total - MF_SLOTS : 0u;
    if (tid == 0) ctl[1] = extra ? atomicAdd(a.cand_cnt, extra) : 0u;
    __syncthreads();
    const uint32_t lbase = ctl[1];
    const uint32_t n_out = total > (uint32_t)MF_SLOTS ? total : (uint32_t)MF_SLOTS;
    for (uint32_t i = tid; i < n_out; i += 256) {
        if (i < (uint32_t)MF_SLOTS) my_slots[i] = i < total ? em[i] : KEY_NONE;
        else if (lbase + (i - MF_SLOTS) < a.cand_cap) a.cand[lbase + (i - MF_SLOTS)] = em[i];
    }
}

static void unpack_workspace(MfmaWorkspace &w, unsigned char *ws_base, const size_t *offs) {
    w.q_h = (_Float16 *)(ws_base + offs[0]); w.qnorm = (float *)(ws_base + offs[1]); w.thr = (float *)(ws_base + offs[2]);
    w.eps = (float *)(ws_base + offs[3]); w.cand_cnt = (uint32_t *)(ws_base + offs[4]); w.fallback = (uint32_t *)(ws_base + offs[5]);
    w.fb_list = (uint32_t *)(ws_base + offs[6]); w.fb_count = (uint32_t *)(ws_base + offs[7]); w.stats = (uint32_t *)(ws_base + offs[8]);
    w.blockmax = (float *)(ws_base + offs[9]); w.cand = (uint64_t *)(ws_base + offs[10]); w.slots = (uint64_t *)(ws_base + offs[11]);
    w.eps2 = (float *)(ws_base + offs[12]);
    w.dcnt = (uint32_t *)(ws_base + offs[13]); w.dthr = (float *)(w.dcnt + (offs[14] - offs[13]) / 4 / (MF_DYN_REP + 1) * MF_DYN_REP);      // n_slots * REP * NB counters, then n_slots * NB thresholds
    w.dpar = (uint32_t *)(ws_base + offs[14]); w.dpub = w.dpar + (offs[15] - offs[14]) / 12 * 2;
}

// eps (DESIGN.md "error bound"). The pre-scan multiplies q~ = fp16(256 q) / 256 by c~ = fp16(256 c) / 256 exactly (fp16 x fp16 products are exact in f32) and adds in f32:
//   |s~ - dot_ref| <= |(q~ - q) . c~| + |q . (c~ - c)| + accumulation + the reference's own rounding
//                  <= |q~ - q| (|c| + |c~ - c|) + |q| |c~ - c| + dim 2^-23 1.01 |q||c| + max(1e-5, dim 2^-24 1.01) |q||c|      (Cauchy-Schwarz)
// Rounds 1 - 5 charged the two rounding terms at their worst case, 2^-11 per element each: 9.7704e-4 |q||c|. Round 6 measures them: convert_rows_kernel keeps the largest
// |c - c~| over the rows (maxres; index.hip), convert_queries_kernel the query's own |q - q~| -- on unit vectors ~2.5e-4 each, half the worst case, so the window of the
// final stage and the emit threshold's margin shrink accordingly (k = 120: 25 % fewer rows re-scored). What a flushed / gradually underflowing fp16 denormal adds on top
// of the stored value stays charged per element in the absolute term, as before. The single-query scan keeps the query in f32: no residual term of its own.
//   eps = rel maxnorm |q| + maxres |q| + qres (maxnorm + maxres) + abs_a (|q| + maxnorm) + 1e-9        rel = dim 1.1921e-7 1.01 + max(1e-5, dim 5.9605e-8 1.01)
struct EpsCoef { float rel, abs_a, rel2, abs2_a; };
static EpsCoef eps_coefficients(uint32_t dim, uint32_t order) {
    EpsCoef c;
    // accumulation of the pre-scan (dim terms, charged a whole ulp each: the matrix pipe's internal rounding mode is not documented) + the reference's own sum:
    // any summation order of dim products is within gamma_dim = dim 2^-24 (1 + ...) of the exact dot (rounds 1 - 5 charged a flat 1e-5 here, short of the
    // strict worst case from dim = 168 up; with the rounding terms measured there is no slack left elsewhere to lean on)
    c.rel = (float)dim * 1.1921e-7f * 1.01f + fmaxf(1.0e-5f, (float)dim * 5.9605e-8f * 1.01f);
    c.abs_a = 2.3842e-7f * __builtin_sqrtf((float)dim) * 1.01f + (order == SHODH_ORDER_SEQ_1M ? 1.0e-6f : 0.0f);   // SEQ_1M: + the rounding of `1 - dot` (<= 2^-24 |1 - dot|, |dot| <= qn*maxnorm), charged generously
    // level 2 (final stage, dense corpora only): s2 = f32 FMA dot in any order. Both s2 and the reference's sum are within
    // gamma_dim * |q||c| of the exact dot (gamma_n = n 2^-24 / (1 - n 2^-24)), so |s2 - dot_ref| <= dim * 2^-23 * |q| * maxnorm
    c.rel2 = (float)dim * 1.1921e-7f * 1.01f;
    c.abs2_a = (order == SHODH_ORDER_SEQ_1M ? 1.0e-6f : 0.0f);
    return c;
}

// room for the global threshold mode's keys and wave maxima behind query 0's overflow list (the lists of the other 255 query slots of a
// pass are unused by a single-query call)
static bool solo_global_fits(const MfmaPlan &p, uint64_t n_rows, uint64_t nb) {
    const uint64_t room = ((uint64_t)p.n_slots - 1) * p.cand_cap * 8;
    return n_rows * 4 + nb * SOLO_WAVES * 4 + 512 <= room;
}
static uint64_t solo_workgroups(uint64_t n_rows, int cus) {
    uint64_t nb = (uint64_t)(cus > 0 ? cus : 1);
    if (nb > ceil_div(n_rows, 64)) nb = ceil_div(n_rows, 64);
    return nb < 1 ? 1 : nb;
}
bool solo_supported(uint32_t nq, uint32_t k, uint64_t n_rows, int cus, const MfmaPlan &p) {
    static const bool off = getenv("SHODH_SOLO") && atoi(getenv("SHODH_SOLO")) == 0;
    static const uint32_t max_k = getenv("SHODH_SOLO_MAX_K") ? (uint32_t)atoi(getenv("SHODH_SOLO_MAX_K")) : SOLO_MAX_K;
    if (off || nq != 1 || k < 1) return false;
    if (k <= max_k) return true;
    return k <= SOLO_MAX_K_GLOBAL && solo_global_fits(p, n_rows, solo_workgroups(n_rows, cus));
}

// The single-query pipeline: solo_scan_kernel [+ solo_emit_kernel] + final stage; like launch_mfma_pipeline it leaves an unresolved query in
// fb_list / fb_count. `solo_cnt`: one u32 that is zero between calls (allocated zeroed by the caller, handed back zeroed by the final stage).
int launch_solo_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim, const uint32_t *deleted, const float *d_q,
                         uint32_t k, uint32_t order, uint32_t id_base, float maxnorm, float maxres, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs,
                         uint32_t *solo_cnt, int cus, uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                         hipEvent_t ev_scan_done, hipEvent_t ev_select_done, hipEvent_t ev0, hipEvent_t ev1, uint32_t *stats_ext, uint32_t *stats_mirror) {
    MfmaWorkspace w;
    unpack_workspace(w, ws_base, offs);
    if (stats_ext) w.stats = stats_ext;
    w.stats_mirror = stats_mirror;
    static const uint32_t max_k_local = getenv("SHODH_SOLO_MAX_K") ? (uint32_t)atoi(getenv("SHODH_SOLO_MAX_K")) : SOLO_MAX_K;
    const bool global_thr = k > max_k_local;
    uint64_t nb = solo_workgroups(n_rows, cus);
    uint64_t slice = ceil_div(n_rows, nb);
    if (!global_thr && slice > (uint64_t)SOLO_MAX_SLICE) { slice = SOLO_MAX_SLICE; nb = ceil_div(n_rows, slice); }    // (the slice's keys live in LDS)
    if (slice < 1) slice = 1;
    const uint64_t slice_c = slice;                   // the contiguous partition (solo_emit_kernel walks the keys by it whatever the scan's mapping was)
    // blocks dealt round-robin (see the kernel) once every workgroup has a few of them; the keys of a workgroup's rows must still fit its LDS
    static const int ilv_env = getenv("SHODH_SOLO_ILV") ? atoi(getenv("SHODH_SOLO_ILV")) : -1;
    const uint32_t epl = dim / 16, iblk = 64u * (epl <= 24 ? 4u : (epl <= 32 ? 2u : 1u));      // == the kernel's 64 U
    uint32_t ilv = (ilv_env >= 0 ? ilv_env != 0 : false) && n_rows >= nb * iblk * 4 ? 1u : 0u;      // (with the direct loads: no gain measured, 184 -> 187 us per call at k = 10; off)
    // ... and, at 384 dimensions, streamed through LDS (32-row tiles; see the kernel) when a workgroup has enough tiles to fill the ring and its keys leave room for it
    static const int stream_env = getenv("SHODH_SOLO_STREAM") ? atoi(getenv("SHODH_SOLO_STREAM")) : -1;
    uint32_t stream = 0, ring_off = 0, dwl_cap = 0;
    if (dim == 384 && (stream_env >= 0 ? stream_env != 0 : true) && n_rows >= nb * 32 * 8) {
        const uint64_t per = ceil_div(ceil_div(n_rows, (uint64_t)32), nb) * 32;
        const size_t head = global_thr ? 0 : ((((size_t)per + (per & 1)) * 4 + (size_t)SOLO_EM * 8 + (size_t)KTH_SCRATCH_U32 * 4 + 16 + 1023) & ~(size_t)1023);
        const size_t need = head + 5 * (size_t)(32 * 384 * 2) + (size_t)(per / 32) * 4 + 2 * 32 * 4;
        if ((global_thr || per <= (uint64_t)SOLO_MAX_SLICE) && need <= 160 * 1024) { stream = 1; ilv = 1; slice = per; ring_off = (uint32_t)head; dwl_cap = (uint32_t)(per / 32); }
    }
    static const bool solo_dbg = getenv("SHODH_SOLO_DEBUG") && atoi(getenv("SHODH_SOLO_DEBUG"));
    if (solo_dbg) fprintf(stderr, "[shodh] single-query scan: %llu rows, %llu workgroups, k %u (%s threshold), %s\n", (unsigned long long)n_rows, (unsigned long long)nb, k,
                          global_thr ? "global" : "local", stream ? "LDS ring of dealt 32-row tiles" : (ilv ? "direct loads, dealt blocks" : "direct loads, contiguous slices"));
    if (ilv && !stream) {
        const uint64_t per = ceil_div(ceil_div(n_rows, (uint64_t)iblk), nb) * iblk;
        if (!global_thr && per > (uint64_t)SOLO_MAX_SLICE) ilv = 0;
        else slice = per;
    }
    if (nb > (uint64_t)p.grid_x * MF_BPAD) { set_error("single-query scan: %llu workgroups do not fit the slot table", (unsigned long long)nb); return SHODH_ERR_UNSUPPORTED; }
    const EpsCoef c = eps_coefficients(dim, order);
    uint32_t *skeys = reinterpret_cast<uint32_t *>(w.cand + p.cand_cap);          // behind query 0's list (solo_global_fits)
    uint32_t *tops = skeys + ((n_rows + 63) & ~(uint64_t)63);
    SoloArgs a{rows_h, n_rows, dim, d_q, deleted, k, (uint32_t)slice, c.rel * maxnorm + maxres, c.abs_a, maxnorm, c.rel2 * maxnorm, c.abs2_a,
               w.slots, w.cand, solo_cnt, p.cand_cap, w.eps, w.eps2, w.fallback, w.fb_count, w.stats, 0u, skeys, tops, ilv, stream, ring_off, dwl_cap};
#ifdef SHODH_DIAG
    a.ablate = getenv("SHODH_SOLO_ABLATE") ? (uint32_t)atoi(getenv("SHODH_SOLO_ABLATE")) : 0u;
#endif
    const size_t lds = stream ? (size_t)ring_off + 5 * (size_t)(32 * 384 * 2) + (size_t)dwl_cap * 4 + 2 * 32 * 4
                              : (global_thr ? 0 : ((size_t)slice + (slice & 1)) * 4 + (size_t)SOLO_EM * 8 + (size_t)KTH_SCRATCH_U32 * 4 + 16);
#define SHODH_LAUNCH_SOLO_K(KERN)                                                                                     \
        SHODH_TRY(ensure_dynamic_lds((const void *)KERN, lds));                                                       \
        if (ev0 && ev1) hipExtLaunchKernelGGL(KERN, dim3((uint32_t)nb), dim3(SOLO_NT), (uint32_t)lds, st, ev0, ev1, 0u, a);  \
        else hipLaunchKernelGGL(KERN, dim3((uint32_t)nb), dim3(SOLO_NT), lds, st, a);
#define SHODH_LAUNCH_SOLO(D)                                                                                          \
    case D:                                                                                                           \
        if (global_thr) { SHODH_LAUNCH_SOLO_K((solo_scan_kernel<D, true>)) } else { SHODH_LAUNCH_SOLO_K((solo_scan_kernel<D, false>)) }  \
        break;
    switch (dim) {
        SHODH_LAUNCH_SOLO(128)
        SHODH_LAUNCH_SOLO(256)
        SHODH_LAUNCH_SOLO(384)
        SHODH_LAUNCH_SOLO(512)
        SHODH_LAUNCH_SOLO(768)
        SHODH_LAUNCH_SOLO(1024)
        default: set_error("single-query scan: unsupported dim %u", dim); return SHODH_ERR_UNSUPPORTED;
    }
#undef SHODH_LAUNCH_SOLO
#undef SHODH_LAUNCH_SOLO_K
    SHODH_HIP_TRY(hipGetLastError());
    if (global_thr) {
        SoloEmitArgs e{skeys, tops, (uint32_t)(nb * SOLO_WAVES), n_rows, (uint32_t)slice_c, k, w.eps, w.fallback, w.slots, w.cand, solo_cnt, p.cand_cap};
        hipLaunchKernelGGL(solo_emit_kernel, dim3((uint32_t)nb), dim3(256), 0, st, e);
        SHODH_HIP_TRY(hipGetLastError());
    }
    if (ev_scan_done) SHODH_HIP_TRY(hipEventRecord(ev_scan_done, st));
    SHODH_TRY(launch_final_stage(rows, dim, d_q, 1, k, order, id_base, p, w, (uint32_t)nb, solo_cnt, solo_cnt, d_ids, d_dist, d_counts, st));
    if (ev_select_done) SHODH_HIP_TRY(hipEventRecord(ev_select_done, st));
    return SHODH_OK;
}

// Enqueues the whole pre-scan + re-score pipeline. Queries that could not be resolved here
// (list overflow, unusable threshold, unquantisable query) are left in ws.fb_list / ws.fb_count
// for the exact scan, which the caller enqueues right after (flat_exact.hip, device-side count).
int launch_mfma_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim,
                         const uint32_t *deleted, const float *d_q, uint32_t nq, uint32_t k, uint32_t order,
                         uint32_t id_base, float maxnorm, float maxres, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs,
                         uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                         hipEvent_t ev_scan_done, hipEvent_t ev_select_done, hipEvent_t ev_emit0, hipEvent_t ev_emit1, uint32_t *stats_ext, uint32_t *stats_mirror) {
    MfmaWorkspace w;
    unpack_workspace(w, ws_base, offs);
    if (stats_ext) w.stats = stats_ext;      // host-pointer calls: the statistics words of the caller's output block
    w.stats_mirror = stats_mirror;

    QueryPrep qp{d_q, nq, dim, p.n_slots, w.q_h, w.qnorm, w.eps, w.cand_cnt, w.fallback, w.fb_count, w.stats};
    hipLaunchKernelGGL(convert_queries_kernel, dim3((p.n_slots * 64 + 255) / 256), dim3(256), 0, st, qp);
    SHODH_HIP_TRY(hipGetLastError());

#ifdef SHODH_DIAG      // diagnostic build only (-DSHODH_DIAG): switches parts of the emit scan off to time the rest, results invalid
    static const uint32_t ablate = getenv("SHODH_ABLATE") ? (uint32_t)atoi(getenv("SHODH_ABLATE")) : 0u;
#else
    const uint32_t ablate = 0u;
#endif
    MfmaArgs a{rows_h, n_rows, dim, w.q_h, w.thr, deleted, w.slots, w.cand, w.cand_cnt, p.cand_cap, w.blockmax, p.tile_stride, p.n_sel_tiles, 0u, nq, nullptr, nullptr, nullptr, nullptr, k};
    // the threshold that tightens during the emit scan (MfmaArgs::dyn): dim <= 384 (the 512-d kernel has no registers left), SHODH_DYN_THR=0 turns it off
    static const bool dyn_env = !(getenv("SHODH_DYN_THR") && atoi(getenv("SHODH_DYN_THR")) == 0);
    const bool dyn = dyn_env && dim <= 384 && k >= 1;
    SHODH_TRY(launch_scan<MF_MODE_BLOCKMAX>(a, p, p.n_sel_tiles, st));

    const EpsCoef c = eps_coefficients(dim, order);
    const float eps_rel = c.rel, eps_abs_a = c.abs_a, eps2_rel = c.rel2, eps2_abs_a = c.abs2_a;
    uint32_t r_top = p.tile_stride ? k / p.tile_stride : 0u;
    if (r_top < 1) r_top = 1;
    ThrArgs t{w.blockmax, p.J, k, p.topk_cap, nq, w.qnorm, w.fallback, eps_rel * maxnorm + maxres, eps_abs_a, maxnorm, maxnorm + maxres, w.thr, w.eps,
              eps2_rel * maxnorm, eps2_abs_a, w.eps2, dyn ? w.dcnt : nullptr, w.dthr, w.dpub, w.dpar, r_top};
    const size_t tlds = (size_t)p.topk_cap * 8 + 512 * 8 + 8 + 4 + 16;
    (void)tlds;
    hipLaunchKernelGGL(threshold_kernel, dim3(p.n_slots), dim3(256), 0, st, t);
    SHODH_HIP_TRY(hipGetLastError());

    a.tile_stride = 1;
    a.n_sel_tiles = (uint32_t)p.n_tiles;
    a.ablate = ablate;
    if (dyn) { a.dcnt = w.dcnt; a.dthr = w.dthr; a.dpub = w.dpub; a.dpar = w.dpar; }
    SHODH_TRY(launch_scan<MF_MODE_EMIT>(a, p, (uint32_t)p.n_tiles, st, ev_emit0, ev_emit1));
    if (ev_scan_done) SHODH_HIP_TRY(hipEventRecord(ev_scan_done, st));

    const uint32_t nb_emit = (uint32_t)p.grid_x > (uint32_t)p.n_tiles ? (p.n_tiles ? (uint32_t)p.n_tiles : 1u) : (uint32_t)p.grid_x;   // = launch_scan's grid.x
    SHODH_TRY(launch_final_stage(rows, dim, d_q, nq, k, order, id_base, p, w, nb_emit, w.cand_cnt, nullptr, d_ids, d_dist, d_counts, st));
    if (ev_select_done) SHODH_HIP_TRY(hipEventRecord(ev_select_done, st));
    return SHODH_OK;
}

// ---- IVF probe selection on a small table (round 6) ----------------------------------------------------------------------------------------------
// SpannIndex::search asks the centroid table for the nprobe nearest partitions of every query (spann.rs:595-607) -- thousands of queries against a few
// thousand rows, k <= 64 -- and only WHICH partitions come back reaches the result (set mode, FinalArgs::set_only). Through the general pipeline that was
// seven launches and 117 us of configs[3]'s step (sample scan, threshold, emit scan, a 57-us final stage of 1024 four-wave workgroups, two empty fallback
// launches): built for a corpus that does not fit anywhere, where a table of 4096 rows x 1024 queries is 16 MiB of scores. Here:
//   1. mfma_scan_kernel<MF_MODE_SCORES>: the same pre-scan (same fp16 shadow, same error bound), every score written out;
//   2. probe_select_kernel: four waves per query hold the query's scores in registers, find the k-th best, classify every row against
//      kth +- 2 eps -- above: a sure member (at most k - 1 rows can score above kth + eps exactly); below: out; in between: scored in the reference's order
//      (spann.rs:562-571: 1 - sum, strictly sequential: the products by everybody, the sum of a row by one thread) -- and write the k ids: the sure ones by approximate score, then the best of the
//      rest by (distance, id). No overflow paths: an unquantisable query, or fewer rows than k, makes every row "in between" (sixteen rows per round, the k best
//      kept across rounds), a table of near-equal rows likewise. Same sets as the final stage's set mode (tests/test_probe_select_gpu.py), 30 us.
struct PselArgs {
    const float *rows; uint32_t n_rows, dim, id_base;
    const float *q; uint32_t nq, k;
    const float *scores; uint32_t ld;
    const float *qnorm; const uint32_t *unquantisable;
    const float *qres; float res_mul;      // the query's rounding residual (convert_queries_kernel) and its coefficient maxnorm + maxres
    float eps_rel_maxnorm, eps_abs_a, maxnorm, eps2_rel_maxnorm, eps2_abs_a;
    uint32_t *ids; float *dist; uint32_t *counts;
};
constexpr int PSEL_MAX_K = 64;
constexpr int PSEL_NT = 256;         // four waves per query: a quarter of the scores each, and four waves per SIMD over the launch to hide one another's latencies
constexpr int PSEL_LIST = 256;       // keys at or under the first bound ranked by counting (more -- ties -- : bisection)
#ifndef SHODH_PSEL_RB      // (diagnostic builds time other shapes)
#define SHODH_PSEL_RB 16
#endif
constexpr int PSEL_RB = SHODH_PSEL_RB;          // rows in between scored per sub-round (their products: PSEL_RB x (dim + 4) floats of LDS)

constexpr int PSEL_R2 = 64;          // level 2: rows whose f32 score is formed per sub-round (their partial sums: PSEL_R2 x dim / 4 floats of the same LDS)
constexpr int PSEL_K2 = 512;         // level 2 takes up to this many rows in between (more: they are all scored exactly)
__host__ __device__ inline size_t psel_region_floats(uint32_t dim) {
    const size_t a = (size_t)PSEL_RB * (dim + 4), b = (size_t)PSEL_R2 * (dim / 4);
    return a > b ? a : b;
}
__host__ __device__ inline size_t psel_lds_bytes(uint32_t dim, uint32_t rows_cap) {
    return (size_t)dim * 4 + psel_region_floats(dim) * 4 + 128 * 8 + 64 * 8 + 64 * 8 + (size_t)PSEL_LIST * 4 + 64 * 4 + (size_t)PSEL_K2 * 4 + 64 + (((size_t)rows_cap * 2 + 15) & ~(size_t)15);
}

#ifdef SHODH_PROF      // (wall-clock phase timers: s_memrealtime, 10 ns ticks)
#define WPROF_DECL long long wp_[5] = {0, 0, 0, 0, 0}, wq_ = wall_clock64(); const long long wt0_ = wq_;
#define WPROF_T(i) { const long long t_ = wall_clock64(); wp_[i] += t_ - wq_; wq_ = t_; }
#else
#define WPROF_DECL
#define WPROF_T(i)
#endif
template <int NV4>      // 16-byte groups of scores per thread: tables of up to 1024 * NV4 rows; row of value (j, tid, u) = 4 (256 j + tid) + u
__global__ __launch_bounds__(PSEL_NT) void probe_select_kernel(PselArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = reinterpret_cast<float *>(smem);                                 // [dim]
    float *prod = qs + a.dim;                                                    // [PSEL_RB][dim + 4] products q[i] * row[i] of the rows being scored
    uint64_t *best = reinterpret_cast<uint64_t *>(prod + psel_region_floats(a.dim));       // [128] exact keys: the best so far, this sub-round's
    uint64_t *skeys = best + 128;                                                // [64] sure members' keys
    uint64_t *s2keys = skeys + 64;                                               // [64] level 2: sure by their f32 score
    uint32_t *klist = reinterpret_cast<uint32_t *>(s2keys + 64);                 // [PSEL_LIST] keys at or under the first bound; level 2: the rows still in between
    uint32_t *mins = klist + PSEL_LIST;                                          // [64] group minima
    uint32_t *k2 = mins + 64;                                                    // [PSEL_K2] level 2: order keys of the f32 scores of the rows in between
    uint32_t *ctl = k2 + PSEL_K2;                                                // [16] counters and hand-over words
    uint16_t *rlist = reinterpret_cast<uint16_t *>(ctl + 16);                    // [rows] the rows in between
    const uint32_t tid = threadIdx.x, q = blockIdx.x, n = a.n_rows, k = a.k, dim = a.dim, pitch = dim + 4;
    const uint32_t d4inv = ((1u << 20) + dim / 4 - 1) / (dim / 4);                   // e / (dim / 4) = (e * d4inv) >> 20 for e < 2^13
    WPROF_DECL
    // the query (for the exact sums), this thread's scores as ascending order keys (a row past the table: 0xFFFFFFFF); everything requested before anything is used
    uint32_t key[NV4][4];
    {
        const f32x4s *sp = reinterpret_cast<const f32x4s *>(a.scores + (size_t)q * a.ld);
        const uint32_t g_max = a.ld / 4;
        f32x4s v[NV4];
#pragma unroll
        for (int j = 0; j < NV4; ++j) { const uint32_t g = j * PSEL_NT + tid; v[j] = sp[g < g_max ? g : 0]; }
        for (uint32_t i = tid; i < dim; i += PSEL_NT) qs[i] = a.q[(size_t)q * dim + i];
        if (tid < 16) ctl[tid] = tid == 1 ? 0xFFFFFFFFu : 0u;                      // [0] list count, [1] T0 / k-th key, [2] sure, [3] in between, [4 .. 6] bisection counts, [7] level-2 key, [8] sure by it, [9] still in between
#pragma unroll
        for (int j = 0; j < NV4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) key[j][u] = (uint32_t)(4 * (j * PSEL_NT + tid) + u) < n ? order_key(-v[j][u]) : 0xFFFFFFFFu;
    }
    uint32_t lmin = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < NV4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) lmin = key[j][u] < lmin ? key[j][u] : lmin;
    {   // the minimum of every four neighbouring threads: 64 group minima (ranking all 256 thread minima cost more than everything else together)
        uint32_t g = lmin;
        uint32_t o = (uint32_t)__shfl_xor((int)g, 1); g = o < g ? o : g;
        o = (uint32_t)__shfl_xor((int)g, 2); g = o < g ? o : g;
        if ((tid & 3u) == 0) mins[tid >> 2] = g;
    }
    const float qn = a.qnorm[q];
    const float eps = a.eps_rel_maxnorm * qn + a.qres[q] * a.res_mul + a.eps_abs_a * (qn + a.maxnorm) + 1e-9f;      // (threshold_kernel's expression)
    const bool usual = !a.unquantisable[q] && k >= 1 && k <= n;                  // block-uniform. Otherwise every row is "in between"
    __syncthreads();
    WPROF_T(0)
    float lo = -__builtin_inff(), hi = __builtin_inff();
    if (usual) {
        // k-th smallest key. First a bound: the k-th smallest of the 64 group minima (k distinct keys are at or under it: the first k groups hold rows, n >= 16 k; a group without rows holds 0xFFFFFFFF) ...
        if (tid < 64) {
            const uint32_t gm = mins[tid];
            uint32_t rk = 0;
            for (uint32_t t = 0; t < 64; t += 4) {
                const u32x4 w = *reinterpret_cast<const u32x4 *>(mins + t);
#pragma unroll
                for (int u = 0; u < 4; ++u) rk += (w[u] < gm) || (w[u] == gm && t + u < tid);
            }
            if (rk == k - 1) ctl[1] = gm;                                         // exactly one thread
        }
        __syncthreads();
        const uint32_t T0 = ctl[1];
        // ... then the keys at or under it (a few dozen), ranked by counting -- or bisected when ties make them many
#pragma unroll
        for (int j = 0; j < NV4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (key[j][u] <= T0) { const uint32_t pos = atomicAdd(&ctl[0], 1u); if (pos < (uint32_t)PSEL_LIST) klist[pos] = key[j][u]; }
        __syncthreads();
        const uint32_t c = ctl[0];
        if (c <= (uint32_t)PSEL_LIST) {
            if (tid < c) {
                const uint32_t v = klist[tid];
                uint32_t r = 0;
                for (uint32_t t = 0; t < c; ++t) { const uint32_t w = klist[t]; r += (w < v) || (w == v && t < tid); }
                if (r == k - 1) ctl[1] = v;                                       // (one element has rank k - 1; T0 is at least it)
            }
            __syncthreads();
        } else {
            uint32_t blo = 0, bhi = T0;                  // smallest K with count(key <= K) >= k; block-uniform
            for (uint32_t it = 0; blo < bhi; ++it) {
                const uint32_t mid = blo + ((bhi - blo) >> 1);
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < NV4; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) cnt += key[j][u] <= mid;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
                uint32_t *slot = &ctl[4 + (it % 3u)];                            // (three slots in turn: the one cleared here is the one read two rounds ago)
                if ((tid & 63u) == 0) atomicAdd(slot, cnt);
                if (tid == 0) ctl[4 + ((it + 1) % 3u)] = 0u;
                __syncthreads();
                if (*slot >= k) bhi = mid; else blo = mid + 1;
            }
            __syncthreads();
            if (tid == 0) ctl[1] = blo;
            __syncthreads();
        }
        const float kth = -order_key_inv(ctl[1]);
        const float margin = 2.001f * eps + 1e-7f * __builtin_fabsf(kth);            // (final_stage_kernel's window)
        lo = kth - margin; hi = kth + margin;
    }
    WPROF_T(1)
    const uint32_t key_hi = order_key(-hi), key_lo = order_key(-lo);                 // score > hi <=> key < key_hi; score >= lo <=> key <= key_lo
    // sure members -> skeys; rows in between -> rlist (in any order: what follows orders by key)
#pragma unroll
    for (int j = 0; j < NV4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t r = 4 * (j * PSEL_NT + tid) + u;
            if (r < n && key[j][u] <= key_lo) {
                if (key[j][u] < key_hi) {
                    const uint32_t pos = atomicAdd(&ctl[2], 1u);
                    if (pos < 64u) skeys[pos] = make_key(-100.0f + order_key_inv(key[j][u]), a.id_base + r);      // = make_key(-100 - s~, id): under every distance
                } else {
                    rlist[atomicAdd(&ctl[3], 1u)] = (uint16_t)r;
                }
            }
        }
    __syncthreads();
    uint32_t n_sure = ctl[2];
    const uint32_t n_amb0 = ctl[3];
    if (k && n_sure > k - 1u) n_sure = k - 1u;                                       // (cannot happen: fewer than k rows score above the k-th best)
    uint32_t want = k > n_sure ? k - n_sure : 0u;                                    // rows to take from the ones in between
    uint32_t n_amb = n_amb0, n_sure2 = 0;
    WPROF_T(2)
    // Level 2 (final_stage_kernel's, for the same reason): the band of the fp16 pre-scan is ~1e-3 wide and on a trained table most of the k nearest lie inside
    // it (configs[3]: 50 - 90 rows in between per query, ten sure) -- and every one of them would cost a 384-step dependent chain. An f32 score s2 (FMA, any
    // order: |s2 - dot_ref| <= eps2 ~ dim 2^-23 |q| maxnorm, twenty times tighter) of each such row, formed in parallel, settles all but the few within 2 eps2
    // of the want-th best s2: above -> a member for sure, below -> out, only the rest is summed in the reference's order.
    const uint32_t d4 = dim / 4;
    if (want && n_amb > (uint32_t)PSEL_RB && n_amb <= (uint32_t)PSEL_K2) {            // block-uniform
        float *psum = prod;                                                          // [PSEL_R2][d4] sums of four products
        for (uint32_t r0 = 0; r0 < n_amb; r0 += PSEL_R2) {
            const uint32_t nb = n_amb - r0 < (uint32_t)PSEL_R2 ? n_amb - r0 : (uint32_t)PSEL_R2, ng = nb * d4;
            for (uint32_t e0 = 0; e0 < ng; e0 += 8 * PSEL_NT) {
                f32x4s cv[8];
                uint32_t ee[8], eg[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t e = e0 + u * PSEL_NT + tid, ec = e < ng ? e : ng - 1, er = (ec * d4inv) >> 20;      // (= ec / d4: exact for ec < 2^13, d4 <= 128)
                    ee[u] = ec; eg[u] = ec - er * d4;
                    cv[u] = *reinterpret_cast<const f32x4s *>(a.rows + (size_t)rlist[r0 + er] * dim + 4 * eg[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (e0 + u * PSEL_NT + tid < ng) {
                        const f32x4s w = *reinterpret_cast<const f32x4s *>(qs + 4 * eg[u]);
                        float p = w[0] * cv[u][0];
                        p = __builtin_fmaf(w[1], cv[u][1], p); p = __builtin_fmaf(w[2], cv[u][2], p); p = __builtin_fmaf(w[3], cv[u][3], p);
                        psum[ee[u]] = p;                                              // (row er, group eg: index er * d4 + eg = ec)
                    }
                }
            }
            __syncthreads();
            {   // four threads per row: a quarter of its group sums each, then the four quarters (a fixed order: the same bits every run)
                const uint32_t row = tid >> 2, part = tid & 3u, per = d4 / 4;
                float acc = 0.0f;
                if (row < nb) { const float *pp = psum + (size_t)row * d4 + part * per; for (uint32_t g = 0; g < per; ++g) acc += pp[g]; }
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                if (row < nb && part == 0) k2[r0 + row] = order_key(-acc);
            }
            __syncthreads();
        }
        // the want-th best s2 (rank by counting), the band around it, the rows above / inside it
        for (uint32_t i = tid; i < n_amb; i += PSEL_NT) {
            const uint32_t v = k2[i];
            uint32_t r = 0;
            for (uint32_t t = 0; t < n_amb; ++t) { const uint32_t w = k2[t]; r += (w < v) || (w == v && t < i); }
            if (r == want - 1) ctl[7] = v;
        }
        __syncthreads();
        const float t2 = -order_key_inv(ctl[7]);
        const float eps2 = a.eps2_rel_maxnorm * qn + a.eps2_abs_a * (qn + a.maxnorm) + 1e-9f;      // (threshold_kernel's expression)
        const float m2 = 2.001f * eps2 + 1e-7f * __builtin_fabsf(t2);
        const uint32_t key_hi2 = order_key(-(t2 + m2)), key_lo2 = order_key(-(t2 - m2));
        for (uint32_t i = tid; i < n_amb; i += PSEL_NT) {
            const uint32_t v = k2[i], r = rlist[i];
            if (v < key_hi2) {
                const uint32_t pos = atomicAdd(&ctl[8], 1u);
                if (pos < 64u) s2keys[pos] = make_key(-50.0f + order_key_inv(v), a.id_base + r);      // = make_key(-50 - s2, id): after the sure members, before every distance
            } else if (v <= key_lo2) {
                const uint32_t pos = atomicAdd(&ctl[9], 1u);
                if (pos < (uint32_t)PSEL_LIST) klist[pos] = r;
            }
        }
        __syncthreads();
        const uint32_t ns2 = ctl[8], nmid = ctl[9];
        if (nmid <= (uint32_t)PSEL_LIST && ns2 < want) {                             // (otherwise -- hundreds of rows within 2 eps2: ties -- everything in between is summed exactly, as without level 2)
            n_sure2 = ns2;
            want -= ns2;
            if (tid < nmid) rlist[tid] = (uint16_t)klist[tid];
            n_amb = nmid;
            __syncthreads();
        }
    }
    // The rows in between, PSEL_RB at a time: the workgroup fetches them (whole rows, coalesced, all requested at once), every thread multiplies what it fetched,
    // thread r then adds row r's products strictly in index order (spann.rs:562-571: one product, one add per element) and forms 1 - sum. The `want` best by
    // (distance, id) are kept, in order, in best[0 .. have).
    uint32_t have = 0;
    for (uint32_t r0 = 0; r0 < n_amb && want; r0 += PSEL_RB) {
        const uint32_t nb = n_amb - r0 < (uint32_t)PSEL_RB ? n_amb - r0 : (uint32_t)PSEL_RB;
        {                                                                            // PSEL_RB * 128 groups of four at most: eight per thread, all in flight
            f32x4s cv[8];
            uint32_t er[8], eg[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t e = u * PSEL_NT + tid, ec = e < nb * d4 ? e : nb * d4 - 1;
                er[u] = (ec * d4inv) >> 20; eg[u] = ec - er[u] * d4;
                cv[u] = *reinterpret_cast<const f32x4s *>(a.rows + (size_t)rlist[r0 + er[u]] * dim + 4 * eg[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u * PSEL_NT + tid < nb * d4) {
                    const f32x4s w = *reinterpret_cast<const f32x4s *>(qs + 4 * eg[u]);
                    f32x4s pr;
                    pr[0] = w[0] * cv[u][0]; pr[1] = w[1] * cv[u][1]; pr[2] = w[2] * cv[u][2]; pr[3] = w[3] * cv[u][3];
                    *reinterpret_cast<f32x4s *>(prod + (size_t)er[u] * pitch + 4 * eg[u]) = pr;
                }
            }
        }
        __syncthreads();
        if (tid < (uint32_t)PSEL_RB) {
            uint64_t mine = KEY_NONE;
            if (tid < nb) {
                const float *pp = prod + (size_t)tid * pitch;
                float dot = 0.0f;
                for (uint32_t g = 0; g < d4; g += 2) {
                    const f32x4s p0 = *reinterpret_cast<const f32x4s *>(pp + 4 * g), p1 = *reinterpret_cast<const f32x4s *>(pp + 4 * g + 4);      // (dim % 8 == 0)
                    dot = dot + p0[0]; dot = dot + p0[1]; dot = dot + p0[2]; dot = dot + p0[3];
                    dot = dot + p1[0]; dot = dot + p1[1]; dot = dot + p1[2]; dot = dot + p1[3];
                }
                mine = make_key(1.0f - dot, a.id_base + (uint32_t)rlist[r0 + tid]);
            }
            best[64 + tid] = mine;
        }
        __syncthreads();
        // merge: rank every kept key (threads 0 .. 63) and every new one (threads 64 .. 64 + PSEL_RB) among all of them (keys are unique: ids are); the `want` smallest stay
        uint64_t kx = KEY_NONE;
        if (tid < have) kx = best[tid];
        else if (tid >= 64 && tid < 64 + nb) kx = best[tid];
        uint32_t rkx = 0;
        if (kx != KEY_NONE) {
            for (uint32_t t = 0; t < have; ++t) rkx += best[t] < kx;
            for (uint32_t t = 0; t < nb; ++t) rkx += best[64 + t] < kx;
        }
        const bool keep = kx != KEY_NONE && rkx < want;
        __syncthreads();                                                             // everybody has read what it ranks
        if (keep) best[rkx] = kx;
        const uint32_t total = have + nb;
        have = total < want ? total : want;                                          // (every key is real: the smallest `want` of `total` stay)
        __syncthreads();
    }
    WPROF_T(3)
    // the sure members in their order (rank by counting), then the kept ones (already in order)
    if (tid < n_sure) {
        const uint64_t v = skeys[tid];
        uint32_t r = 0;
        for (uint32_t t = 0; t < n_sure; ++t) r += skeys[t] < v;
        a.ids[(size_t)q * k + r] = (uint32_t)v;
        a.dist[(size_t)q * k + r] = order_key_inv((uint32_t)(v >> 32));
    }
    if (tid < n_sure2) {
        const uint64_t v = s2keys[tid];
        uint32_t r = 0;
        for (uint32_t t = 0; t < n_sure2; ++t) r += s2keys[t] < v;
        a.ids[(size_t)q * k + n_sure + r] = (uint32_t)v;
        a.dist[(size_t)q * k + n_sure + r] = order_key_inv((uint32_t)(v >> 32));
    }
    if (tid < have) {
        const uint64_t v = best[tid];
        a.ids[(size_t)q * k + n_sure + n_sure2 + tid] = (uint32_t)v;
        a.dist[(size_t)q * k + n_sure + n_sure2 + tid] = order_key_inv((uint32_t)(v >> 32));
    }
    const uint32_t m = n_sure + n_sure2 + have;
    if (tid >= m && tid < k) { a.ids[(size_t)q * k + tid] = 0xFFFFFFFFu; a.dist[(size_t)q * k + tid] = __builtin_inff(); }
    WPROF_T(4)
#ifdef SHODH_PROF
    if (tid == 0 && k > 1 && ((q % 97) == 5 || wp_[0] + wp_[1] + wp_[2] + wp_[3] + wp_[4] > 2200)) printf("psel q %u n_sure %u n_amb %u | start %lld after the score kernel's last workgroup, end %lld | load %lld kth %lld classify %lld rescore %lld out %lld (10 ns ticks)\n", q, n_sure, n_amb0, wt0_ - (long long)g_scores_end_ticks, (long long)wall_clock64() - (long long)g_scores_end_ticks, wp_[0], wp_[1], wp_[2], wp_[3], wp_[4]);
#endif
    if (tid == 0) a.counts[q] = m;      // (no statistics: two same-address atomics per query were 18 of the kernel's 38 us at 1024 queries)
}

bool probe_select_supported(uint64_t n_rows, uint32_t dim, uint32_t k, uint32_t order, bool has_deleted, const MfmaPlan &p) {
    const char *ev = getenv("SHODH_PROBE_SELECT");                    // tests / A-B runs (read per call): 0 = the general pipeline (set mode of the final stage)
    const bool off = ev && atoi(ev) == 0;
    return !off && p.set_only && order == SHODH_ORDER_SEQ_1M && k >= 1 && k <= (uint32_t)PSEL_MAX_K && n_rows >= 256 && n_rows >= 16ull * k && n_rows <= 8192 && dim <= 512 && (dim % 32) == 0 &&      /* (k groups of sixteen rows each hold a row: the first bound) */
          
           !has_deleted && (uint64_t)p.cand_cap * 2 >= ceil_div(n_rows, MF_TR) * MF_TR;      // (the scores live in the overflow lists' room: cand_cap u64 per query)
}

// convert_queries_kernel, score pre-scan, probe_select_kernel: three launches, nothing left over for the exact scan
int launch_probe_select_pipeline(const float *rows, const _Float16 *rows_h, uint64_t n_rows, uint32_t dim, const float *d_q, uint32_t nq, uint32_t k, uint32_t id_base,
                                 float maxnorm, float maxres, const MfmaPlan &p, unsigned char *ws_base, const size_t *offs, uint32_t *d_ids, float *d_dist, uint32_t *d_counts, hipStream_t st,
                                 uint32_t *stats_ext) {
    MfmaWorkspace w;
    unpack_workspace(w, ws_base, offs);
    if (stats_ext) w.stats = stats_ext;
    QueryPrep qp{d_q, nq, dim, p.n_slots, w.q_h, w.qnorm, w.eps, w.cand_cnt, w.fallback, w.fb_count, w.stats};
    hipLaunchKernelGGL(convert_queries_kernel, dim3((p.n_slots * 64 + 255) / 256), dim3(256), 0, st, qp);
    const uint32_t ld = (uint32_t)(p.n_tiles * MF_TR);
    float *scores = reinterpret_cast<float *>(w.cand);                             // [n_slots][ld] f32 <= [n_slots][cand_cap] u64
    MfmaArgs a{rows_h, n_rows, dim, w.q_h, w.thr, nullptr, w.slots, w.cand, w.cand_cnt, p.cand_cap, w.blockmax, 1u, (uint32_t)p.n_tiles, 0u, nq, nullptr, nullptr, nullptr, nullptr, k, scores, ld};
    SHODH_TRY(launch_scan<MF_MODE_SCORES>(a, p, (uint32_t)p.n_tiles, st));
    const EpsCoef c = eps_coefficients(dim, SHODH_ORDER_SEQ_1M);
    PselArgs s{rows, (uint32_t)n_rows, dim, id_base, d_q, nq, k, scores, ld, w.qnorm, w.fallback, w.eps, maxnorm + maxres, c.rel * maxnorm + maxres, c.abs_a, maxnorm, c.rel2 * maxnorm, c.abs2_a, d_ids, d_dist, d_counts};
    const uint32_t nv4 = (uint32_t)ceil_div(n_rows, 4 * PSEL_NT);
    const size_t lds = psel_lds_bytes(dim, nv4 * 4 * PSEL_NT);
#define SHODH_LAUNCH_PSEL(NVV)                                                                                   \
    do {                                                                                                         \
        SHODH_TRY(ensure_dynamic_lds((const void *)probe_select_kernel<NVV>, lds));                              \
        hipLaunchKernelGGL((probe_select_kernel<NVV>), dim3(nq), dim3(PSEL_NT), lds, st, s);                     \
    } while (0)
    if (nv4 <= 1) SHODH_LAUNCH_PSEL(1); else if (nv4 <= 2) SHODH_LAUNCH_PSEL(2); else if (nv4 <= 4) SHODH_LAUNCH_PSEL(4); else SHODH_LAUNCH_PSEL(8);
#undef SHODH_LAUNCH_PSEL
    SHODH_HIP_TRY(hipGetLastError());
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
int launch_count_nonfinite(const float *x, uint64_t n, uint32_t *counter, hipStream_t st) {
    if (n == 0) return SHODH_OK;
    uint64_t blocks = ceil_div(n, 1024);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, x, n, counter);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int launch_shadow_set_row(const float *rows, _Float16 *rows_h, uint64_t row, uint32_t dim, int zero, hipStream_t st) {
    hipLaunchKernelGGL(shadow_set_row_kernel, dim3(1), dim3(128), 0, st, rows, rows_h, row, dim, zero);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
int launch_shadow_zero_rows(_Float16 *rows_h, const uint32_t *d_list, uint64_t n, uint32_t dim, hipStream_t st) {
    for (uint64_t o = 0; o < n; o += 1u << 30) {
        const uint64_t m = n - o < (1u << 30) ? n - o : (1u << 30);
        hipLaunchKernelGGL(shadow_zero_rows_kernel, dim3((uint32_t)m), dim3(128), 0, st, rows_h, d_list + o, dim);
    }
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

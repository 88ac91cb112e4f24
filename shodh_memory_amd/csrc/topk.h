// topk.h -- block-level "k smallest 64-bit keys" selection in LDS.
//
// Keys are make_key(dist, id): ascending key order == the reference's result order
// (dist by f32::total_cmp, then id; vamana.rs:1185, spann.rs:689-690). Keys are unique (ids are),
// so "k smallest" is a well-defined set.
//
// TopKBuf is a streaming threshold filter: push() appends a key only if it beats the current
// k-th best (thr); when the buffer could overflow it is compacted with a block-wide bitonic sort
// to the k best, which also tightens thr. block_select_topk() is the two-pass form for inputs
// that can be enumerated twice: pass 1 takes each thread's minimum, the k-th smallest of those
// NT minima is a valid upper bound T of the k-th smallest key (k distinct keys <= T exist), and
// pass 2 streams only the keys <= T (about k of them) through the TopKBuf.
#pragma once
#include "common.h"

namespace shodh {

struct TopKBuf {
    uint64_t *keys;   // [cap] in LDS, cap is a power of two
    uint32_t *cnt;    // LDS counter
    uint64_t *thr;    // LDS: push accepts key < thr (KEY_NONE until k keys are known)
    uint32_t cap;
    uint32_t k;
};

__device__ __forceinline__ uint32_t pow2_ge(uint32_t n) { return n <= 1 ? 1u : (1u << (32 - __builtin_clz(n - 1))); }

// block-wide ascending bitonic sort of keys[0..size), size a power of two; entries >= n become KEY_NONE
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, uint32_t size, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = n + tid; i < size; i += NT) keys[i] = KEY_NONE;
    __syncthreads();
    for (uint32_t len = 2; len <= size; len <<= 1) {
        for (uint32_t stride = len >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (size >> 1); t += NT) {
                const uint32_t lo = 2 * t - (t & (stride - 1));   // index with bit `stride` cleared
                const uint32_t hi = lo + stride;
                const bool up = ((lo & len) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
}

// call with the whole block; must be preceded by a __syncthreads() after the last push
template <int NT>
__device__ __forceinline__ void topk_compact(TopKBuf &b) {
    uint32_t n = *b.cnt;
    if (n > b.cap) n = b.cap;
    __syncthreads();
    uint32_t size = pow2_ge(n);
    if (size < 2) size = 2;
    if (size > b.cap) size = b.cap;
    bitonic_sort_lds<NT>(b.keys, size, n);
    if (threadIdx.x == 0) {
        const uint32_t m = n < b.k ? n : b.k;
        *b.cnt = m;
        if (b.k == 0) *b.thr = 0;                       // k == 0: nothing is ever accepted
        else if (m == b.k) { const uint64_t t = b.keys[b.k - 1]; if (t < *b.thr) *b.thr = t; }
    }
    __syncthreads();
}

// "Compact if fewer than `room` slots are free": call with the WHOLE block after the pushes of an iteration (it starts with
// the barrier that ends them). The counter is read between two barriers: a thread that had passed a single-barrier check and
// already pushed for the next iteration could change the value a slower thread was still about to read, the two then disagreed
// about entering the compaction and its barriers paired up with the wrong ones (seen as run-to-run differences in the
// IVF-PQ list scan at 10M rows with 16 waves per workgroup; every kernel used the same idiom).
template <int NT>
__device__ __forceinline__ void topk_compact_if_short(TopKBuf &b, uint32_t room) {
    __syncthreads();
    const uint32_t n = *b.cnt;
    __syncthreads();
    if (n + room > b.cap) topk_compact<NT>(b);
}

__device__ __forceinline__ void topk_push(TopKBuf &b, uint64_t key) {
    if (key < *b.thr) {
        const uint32_t slot = atomicAdd(b.cnt, 1u);
        if (slot < b.cap) b.keys[slot] = key;
    }
}

// ranks the first n (<= 2*NT) keys among themselves by counting (keys are unique; duplicates -- only KEY_NONE
// padding -- are ordered by position): n broadcast LDS reads per key, one barrier, no sorting network.
// Writes the keys in ascending order to `out` (2*NT entries).
// UNIQUE: the caller's keys are distinct (flat index: one key per row) -- a plain '<' ranks them (a 64-bit compare and an add per pair instead of two compares
// and the position tie-break), and while 2 n <= NT several threads share a key's comparisons (the final stage at recall's k = 120 ranks 150 - 280 keys on
// 512 threads: 22 000 cycles of the 105 000 it took, a third of that now).
template <int NT, bool UNIQUE = false>
__device__ __forceinline__ void rank_sort_lds(const uint64_t *in, uint64_t *out, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    if (UNIQUE) {
        uint32_t sh = 0;                                   // 2^sh threads per key (block-uniform)
        while (sh < 3 && (n << (sh + 1)) <= (uint32_t)NT) ++sh;
        const uint32_t S = 1u << sh, i = tid >> sh, part = tid & (S - 1u);
        const uint32_t per = (n + S - 1u) >> sh, j0 = part * per, j1 = j0 + per < n ? j0 + per : n;
        uint32_t r = 0;
        const uint64_t v = i < n ? in[i] : 0ull;
        if (i < n) {
            uint32_t j = j0;
            for (; j + 8 <= j1; j += 8) {
                uint64_t w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = in[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += w[u] < v;
            }
            for (; j < j1; ++j) r += in[j] < v;
        }
        for (uint32_t o = 1; o < S; o <<= 1) r += (uint32_t)__shfl_xor((int)r, (int)o);      // (the S threads of a key are neighbours in one wave: S <= 8 divides 64)
        for (uint32_t i2 = i + (NT >> sh); part == 0 && i2 < n; i2 += (NT >> sh)) {           // (n > NT: never with S > 1; the remaining keys one thread each)
            const uint64_t v2 = in[i2];
            uint32_t r2 = 0;
            for (uint32_t j = 0; j < n; ++j) r2 += in[j] < v2;
            out[r2] = v2;
        }
        if (i < n && part == 0) out[r] = v;
        __syncthreads();
        return;
    }
#pragma unroll
    for (uint32_t e = 0; e < 2; ++e) {
        const uint32_t i = tid + e * NT;
        if (i < n) {
            const uint64_t v = in[i];
            // keys are distinct except for KEY_NONE padding, which never reaches this function's callers with
            // n counting it -- so plain '<' ranks; 8 LDS reads in flight per step (a dependent read per
            // iteration costs the full LDS latency)
            uint32_t r = 0;
            uint32_t j = 0;
            for (; j + 8 <= n; j += 8) {
                uint64_t w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = in[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += (w[u] < v) || (w[u] == v && (j + u) < i);
            }
            for (; j < n; ++j) { const uint64_t w = in[j]; r += (w < v) || (w == v && j < i); }
            out[r] = v;
        }
    }
    __syncthreads();
}

// k smallest of the keys key_at(0..n) (KEY_NONE entries are skipped). Result: b.keys[0..m) sorted
// ascending, m = min(k, #valid) returned and left in *b.cnt. `mins` is 2*NT u64 of LDS scratch.
// Every thread of the block must call this with the same arguments.
template <int NT, bool UNIQUE = false, class KeyFn>
__device__ __forceinline__ uint32_t block_select_topk(KeyFn key_at, uint64_t n, TopKBuf &b, uint64_t *mins) {
    const uint32_t tid = threadIdx.x;
    uint64_t T = KEY_NONE;
    if (b.k > 0 && b.k <= NT && n > (uint64_t)2 * NT) {
        // pass 1: the k-th smallest of the NT per-thread minima bounds the k-th smallest key from above
        uint64_t tmin = KEY_NONE;
        for (uint64_t i = tid; i < n; i += NT) { const uint64_t key = key_at(i); tmin = key < tmin ? key : tmin; }
        mins[tid] = tmin;
        __syncthreads();
        uint32_t r = 0;
        for (uint32_t j = 0; j < NT; j += 8) {
            uint64_t w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = mins[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += (w[u] < tmin) || (w[u] == tmin && (j + u) < tid);
        }
        if (tid == 0) *b.thr = KEY_NONE;
        __syncthreads();
        if (r == b.k - 1) *b.thr = tmin;          // exactly one thread holds rank k-1
        __syncthreads();
        T = *b.thr;
        __syncthreads();
    }
    if (UNIQUE && b.k > 0 && n <= (uint64_t)2 * NT && n <= (uint64_t)b.cap) {
        // every key is real and distinct and they all fit: straight into the buffer (no filter to pass, no counter to queue at), ranked by counting
        for (uint32_t i = tid; i < (uint32_t)n; i += NT) b.keys[i] = key_at(i);
        __syncthreads();
        rank_sort_lds<NT, true>(b.keys, mins, (uint32_t)n);
        const uint32_t m = (uint32_t)n < b.k ? (uint32_t)n : b.k;
        if (tid < m) b.keys[tid] = mins[tid];
        if (tid + NT < m) b.keys[tid + NT] = mins[tid + NT];
        if (tid == 0) { *b.cnt = m; *b.thr = m == b.k ? mins[b.k - 1] : KEY_NONE; }
        __syncthreads();
        return m;
    }
    if (tid == 0) { *b.cnt = 0; *b.thr = b.k ? (T == KEY_NONE ? KEY_NONE : T + 1) : 0; }
    __syncthreads();
    const uint64_t n_iter = (n + NT - 1) / NT;
    for (uint64_t it = 0; it < n_iter; ++it) {
        const uint64_t i = it * NT + tid;
        if (i < n) { const uint64_t key = key_at(i); if (key != KEY_NONE) topk_push(b, key); }
        if ((it & 1) == 1 || it + 1 == n_iter) topk_compact_if_short<NT>(b, 2 * NT);   // at most 2*NT pushes between checks
    }
    __syncthreads();
    const uint32_t cnt = *b.cnt;
    if (cnt <= 2 * NT) {
        // the usual case (about k survivors): order them by counting instead of a sorting network
        __syncthreads();
        rank_sort_lds<NT, UNIQUE>(b.keys, mins, cnt);
        if (tid < cnt) b.keys[tid] = mins[tid];
        if (tid + NT < cnt) b.keys[tid + NT] = mins[tid + NT];
        if (tid == 0) {
            const uint32_t m = cnt < b.k ? cnt : b.k;
            *b.cnt = m;
            if (b.k == 0) *b.thr = 0;
            else if (m == b.k) { const uint64_t t = mins[b.k - 1]; if (t < *b.thr) *b.thr = t; }
        }
        __syncthreads();
    } else {
        topk_compact<NT>(b);
    }
    return *b.cnt;
}

// k-th smallest (k >= 1) of n 32-bit keys key_at(0..n) -- the VALUE only, duplicates allowed. Returns
// 0xFFFFFFFF if fewer than k keys exist (*overflow is always false now).
//   1. filter: the keys are dealt into k groups (by owning thread, tid % k); the k-th smallest key is <= the LARGEST
//      of the k group minima (k distinct keys are <= it), and only ~k*H(k) keys pass that bound on random input;
//   2. the survivors are gathered into LDS;
//   3. few survivors: rank counting (O(c) per thread); many: 4 x 8-bit radix select over the LDS copy.
// A rank loop over c keys costs ~25 cycles * c per thread (wave64 VALU ops issue over 4 cycles), which is why
// the first two steps exist. scratch: KTH_SCRATCH_U32 u32 of LDS. Block-uniform arguments; every thread calls.
constexpr int KTH_BUF = 1024;
constexpr int KTH_SCRATCH_U32 = 128 + KTH_BUF + 256 + 8;
// section timers of block_kth_u32 (PROF builds: -DSHODH_PROF; one line from thread 0 of workgroup 37)
#ifdef SHODH_PROF
#define KTH_PROF_DECL long long kp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, kq_ = clock64();
#define KTH_PROF_T(i) { const long long t_ = clock64(); kp_[i] += t_ - kq_; kq_ = t_; }
#define KTH_PROF_PRINT(tag) if (tid == 0 && blockIdx.x == 37) printf("kth %s NT %d n %u k %u c %u | tmin %lld gmin %lld reduce %lld gather %lld rank %lld prep %lld passes %lld\n", tag, NT, n, k, c, kp_[0], kp_[1], kp_[2], kp_[3], kp_[4], kp_[5], kp_[6]);
#else
#define KTH_PROF_DECL
#define KTH_PROF_T(i)
#define KTH_PROF_PRINT(tag)
#endif
template <int NT, class KeyFn>
__device__ __forceinline__ uint32_t block_kth_u32(KeyFn key_at, uint32_t n, uint32_t k, uint32_t *scratch, bool *overflow) {
    const uint32_t tid = threadIdx.x;
    uint32_t *gmin = scratch, *buf = scratch + 128, *hist = buf + KTH_BUF, *cnt = hist + 256, *res = cnt + 1, *sel = cnt + 2;
    uint32_t T = 0xFFFFFFFFu;
    KTH_PROF_DECL
    const bool filt = k <= 128 && n >= 4u * k;
    if (tid == 0) { *cnt = 0; *res = 0xFFFFFFFFu; }
    if (filt && tid < k) gmin[tid] = 0xFFFFFFFFu;
    __syncthreads();
    if (filt) {
        // group g = the keys of the threads with tid % k == g: one LDS atomic per thread, none for empty threads
        uint32_t tmin = 0xFFFFFFFFu;
        for (uint32_t i = tid; i < n; i += NT) { const uint32_t key = key_at(i); tmin = key < tmin ? key : tmin; }
        KTH_PROF_T(0)
        if (tmin != 0xFFFFFFFFu) atomicMin(&gmin[tid % k], tmin);
        __syncthreads();
        KTH_PROF_T(1)
        // the largest of the k group minima: a loop in every thread for a small k; from k = 24 up a reduction over the first two waves (k <= 128 <= NT) -- two barriers and six
        // shuffles, ~1 000 cycles whatever k is, where the loop's dependent LDS reads cost ~60 cycles per minimum (section timers: `profiles/r6_kth_phases.txt`)
        static_assert(NT >= 128, "block_kth_u32: the group minima are reduced by the first 128 threads");
        uint32_t m = 0;
        if (k < 24u) {
            for (uint32_t j = 0; j < k; ++j) { const uint32_t v = gmin[j]; m = v > m ? v : m; }
        } else {
            m = tid < k ? gmin[tid] : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off); m = o > m ? o : m; }
            if (tid == 0 || tid == 64) sel[tid >> 6] = m;      // (sel: free until the radix passes, which are behind further barriers)
            __syncthreads();
            m = sel[0] > sel[1] ? sel[0] : sel[1];
        }
        T = m;      // 0xFFFFFFFF if some group is empty: nothing is filtered, still correct
        KTH_PROF_T(2)
    }
    for (uint32_t i = tid; i < n; i += NT) {
        const uint32_t key = key_at(i);
        if (key <= T && key != 0xFFFFFFFFu) { const uint32_t slot = atomicAdd(cnt, 1u); if (slot < (uint32_t)KTH_BUF) buf[slot] = key; }
    }
    __syncthreads();
    const uint32_t c = *cnt;
    KTH_PROF_T(3)
    __syncthreads();              // (see the end of the rank path: nobody may reset the counter before everybody has read it)
    *overflow = false;          // (kept in the signature: an overflowing gather now falls through to the direct radix select)
    if (c < k) return 0xFFFFFFFFu;
    if (c <= 128u) {
        if (tid < c) {
            const uint32_t v = buf[tid];
            uint32_t r = 0, j = 0;
            for (; j + 8 <= c; j += 8) {
                uint32_t w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = buf[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += (w[u] < v) || (w[u] == v && (j + u) < tid);
            }
            for (; j < c; ++j) { const uint32_t w = buf[j]; r += (w < v) || (w == v && j < tid); }
            if (r == k - 1) *res = v;
        }
        __syncthreads();
        const uint32_t kth = *res;
        __syncthreads();          // a second call in the same kernel resets *res / *cnt: not before everybody has read them
        KTH_PROF_T(4)
        KTH_PROF_PRINT("rank")
        return kth;
    }
    // many survivors: radix select over the gathered copy, or -- if even that overflowed (c > KTH_BUF: long runs of
    // near-equal keys, or a large k) -- straight over the source keys that passed the filter
    const bool gathered = c <= (uint32_t)KTH_BUF;
    uint32_t prefix = 0, mask = 0, kk = k;
    KTH_PROF_T(5)
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        if (gathered) {
            for (uint32_t i = tid; i < c; i += NT) {
                const uint32_t v = buf[i];
                if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
            }
        } else {
            for (uint32_t i = tid; i < n; i += NT) {
                const uint32_t v = key_at(i);
                if (v <= T && v != 0xFFFFFFFFu && (v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t sum = h0 + h1 + h2 + h3;
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)tid >= o) incl += t; }
            const uint32_t excl = incl - sum;
            if (excl < kk && kk <= incl) {
                uint32_t r = kk - excl, d = 0;
                if (r > h0) { r -= h0; d = 1; if (r > h1) { r -= h1; d = 2; if (r > h2) { r -= h2; d = 3; } } }
                sel[0] = 4 * tid + d; sel[1] = r;
            }
        }
        __syncthreads();
        prefix |= sel[0] << shift; mask |= 255u << shift; kk = sel[1];
    }
    KTH_PROF_T(6)
    KTH_PROF_PRINT("radix")
    return prefix;
}

}  // namespace shodh

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

__device__ __forceinline__ void topk_push(TopKBuf &b, uint64_t key) {
    if (key < *b.thr) {
        const uint32_t slot = atomicAdd(b.cnt, 1u);
        if (slot < b.cap) b.keys[slot] = key;
    }
}

// k smallest of the keys key_at(0..n) (KEY_NONE entries are skipped). Result: b.keys[0..m) sorted
// ascending, m = min(k, #valid) returned and left in *b.cnt. `mins` is NT u64 of LDS scratch.
// Every thread of the block must call this with the same arguments.
template <int NT, class KeyFn>
__device__ __forceinline__ uint32_t block_select_topk(KeyFn key_at, uint64_t n, TopKBuf &b, uint64_t *mins) {
    const uint32_t tid = threadIdx.x;
    uint64_t T = KEY_NONE;
    if (b.k > 0 && b.k <= NT && n > (uint64_t)b.cap / 2) {
        uint64_t tmin = KEY_NONE;
        for (uint64_t i = tid; i < n; i += NT) { const uint64_t key = key_at(i); tmin = key < tmin ? key : tmin; }
        mins[tid] = tmin;
        __syncthreads();
        bitonic_sort_lds<NT>(mins, NT, NT);
        T = mins[b.k - 1];
        __syncthreads();
    }
    if (tid == 0) { *b.cnt = 0; *b.thr = b.k ? (T == KEY_NONE ? KEY_NONE : T + 1) : 0; }
    __syncthreads();
    const uint64_t n_iter = (n + NT - 1) / NT;
    for (uint64_t it = 0; it < n_iter; ++it) {
        const uint64_t i = it * NT + tid;
        if (i < n) { const uint64_t key = key_at(i); if (key != KEY_NONE) topk_push(b, key); }
        if ((it & 1) == 1 || it + 1 == n_iter) {          // at most 2*NT pushes between checks
            __syncthreads();
            if (*b.cnt + 2 * NT > b.cap) topk_compact<NT>(b);
        }
    }
    __syncthreads();
    topk_compact<NT>(b);
    return *b.cnt;
}

}  // namespace shodh

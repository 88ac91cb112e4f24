// topk.h -- block-level "k smallest 64-bit keys" buffer in LDS.
//
// Keys are make_key(dist, id): ascending key order == the reference's result order
// (dist by f32::total_cmp, then id; vamana.rs:1185, spann.rs:689-690). Keys are unique (ids are),
// so "k smallest" is a well-defined set.  The buffer is a threshold filter: push() appends a key
// only if it beats the current k-th best (thr); when the buffer could overflow it is compacted
// with a block-wide bitonic sort to the k best, which also tightens thr.
#pragma once
#include "common.h"

namespace shodh {

struct TopKBuf {
    uint64_t *keys;   // [cap] in LDS, cap is a power of two
    uint32_t *cnt;    // LDS counter
    uint64_t *thr;    // LDS: current k-th best key (KEY_NONE until k keys are known)
    uint32_t cap;
    uint32_t k;
};

// block-wide ascending bitonic sort of keys[0..cap); entries >= n are treated as KEY_NONE
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, uint32_t cap, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < cap; i += NT)
        if (i >= n) keys[i] = KEY_NONE;
    __syncthreads();
    for (uint32_t size = 2; size <= cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (cap >> 1); t += NT) {
                uint32_t lo = 2 * t - (t & (stride - 1));   // index with bit `stride` cleared
                uint32_t hi = lo + stride;
                bool up = ((lo & size) == 0);
                uint64_t a = keys[lo], b = keys[hi];
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
    bitonic_sort_lds<NT>(b.keys, b.cap, n);
    if (threadIdx.x == 0) {
        uint32_t m = n < b.k ? n : b.k;
        *b.cnt = m;
        *b.thr = (m == b.k && b.k > 0) ? b.keys[b.k - 1] : KEY_NONE;
        if (b.k == 0) *b.thr = 0;   // k == 0: nothing is ever accepted
    }
    __syncthreads();
}

__device__ __forceinline__ void topk_push(TopKBuf &b, uint64_t key) {
    if (key < *b.thr) {
        uint32_t slot = atomicAdd(b.cnt, 1u);
        if (slot < b.cap) b.keys[slot] = key;
    }
}

}  // namespace shodh

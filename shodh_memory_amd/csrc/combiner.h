// combiner.h -- coalescing front for concurrent single-item callers (host only).
//
// The reference serves MANY callers with ONE item each: every `recall` is one query under a read lock (src/handlers/recall.rs:512-513
// spawn_blocking + memory.read(); src/memory/retrieval.rs:912 vector_index.read()), every `remember` one encode() behind Mutex<Session>
// (src/embeddings/minilm.rs:889-897). On this device one query costs a whole pass over the corpus' shadow copy (HBM-bound) and up to 256
// queries cost the same pass, so N callers that each launch their own pass get 1/N of the machine each. This front gathers the calls that
// arrive while a pass is in flight into the NEXT pass (group commit):
//
//   * a caller that finds no open batch opens one and becomes its leader; later callers join the open batch (as long as it has room) and
//     sleep on the batch's `done` word;
//   * a leader waits for the previous batch to finish (`go`), closes its batch, runs ONE device pass for all members, scatters the results
//     to the members' own output pointers, hands `go` to the next batch's leader and wakes its members (one futex wake for all of them);
//   * a caller that is alone (no pass in flight, nobody else arriving) starts at once: its call is the same single-item call as without
//     the front, plus two uncontended mutex operations;
//   * linger: a leader that could start but whose batch is smaller than the previous one waits up to `linger_us` for the members of the
//     batch that just ended to come back (closed-loop callers return within microseconds of each other; without this the callers split
//     into a batch of one -- the leader, who needs no wake-up -- and a batch of everyone else, alternating). A lone caller never lingers:
//     its previous batch had one member.
//
// The front changes WHEN work runs, never WHAT is computed: the executor must produce, for every member, the bytes the member's own call
// would have produced (searches: rows of a batched pass are bit-identical to single-query passes, tests/test_concurrent_gpu.py; encodes:
// quant_scope PER_TEXT makes a batch N x encode() by construction).
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <climits>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace shodh {

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

static inline uint64_t mono_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// sleep until *word != 0 (a short poll first: a wake-up through the kernel costs 5-10 us, a pass 150-400 us)
static inline void futex_wait_nonzero(std::atomic<uint32_t> *word, int spin = 64) {
    for (int i = 0; i < spin; ++i) {
        if (word->load(std::memory_order_acquire) != 0) return;
        cpu_relax();
    }
    while (word->load(std::memory_order_acquire) == 0) syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAIT_PRIVATE, 0u, nullptr, nullptr, 0);
}
static inline void futex_set_and_wake_all(std::atomic<uint32_t> *word) {
    word->store(1u, std::memory_order_release);
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

struct CoBatch {
    std::vector<void *> reqs;              // members' requests in arrival order (the leader's first); guarded by Combiner::m until closed
    std::atomic<uint32_t> members{0};
    uint32_t units = 0;                    // queries / texts held
    bool closed = false;
    std::atomic<uint32_t> go{0};           // the leader may start (previous batch done)
    std::atomic<uint32_t> done{0};         // results are in the members' buffers
    int rc = 0;
    std::string err;
};

struct CombinerStats { uint64_t batches = 0, members = 0, max_members = 0, lingered = 0; };

class Combiner {
public:
    uint32_t linger_us = 30;

    // exec(reqs) runs ONE device pass for all requests, writes every member's outputs and returns the status shared by all of them
    // (`err_of_leader` is copied to the members when it is not SHODH_OK). Returns the batch status; *led = this caller ran the pass.
    template <class Exec, class ErrFn>
    int submit(void *req, uint32_t units, uint32_t max_units, Exec &&exec, ErrFn &&last_error, std::string *err_out) {
        std::shared_ptr<CoBatch> b;
        bool leader = false;
        {
            std::lock_guard<std::mutex> g(m);
            if (!open.empty() && !open.back()->closed && open.back()->units + units <= max_units) {
                b = open.back();
            } else {
                b = std::make_shared<CoBatch>();
                leader = true;
                if (!running && open.empty()) b->go.store(1u, std::memory_order_relaxed);
                open.push_back(b);
            }
            b->reqs.push_back(req);
            b->units += units;
            b->members.fetch_add(1u, std::memory_order_release);
        }
        if (!leader) {
            futex_wait_nonzero(&b->done);
            if (b->rc != 0 && err_out) *err_out = b->err;
            return b->rc;
        }
        futex_wait_nonzero(&b->go);
        // linger for the members of the batch that just ended (see the header)
        const uint32_t want = prev_members.load(std::memory_order_relaxed);
        if (linger_us && b->members.load(std::memory_order_acquire) < want) {
            const uint64_t t_end = mono_ns() + (uint64_t)linger_us * 1000ull;
            while (b->members.load(std::memory_order_acquire) < want && mono_ns() < t_end) cpu_relax();
            n_lingered.fetch_add(1u, std::memory_order_relaxed);
        }
        std::vector<void *> reqs;
        {
            std::lock_guard<std::mutex> g(m);
            b->closed = true;
            reqs = b->reqs;                 // (copied: the pass runs without the lock)
            open.pop_front();               // b is the front: batches start in the order they were opened
            running = true;
        }
        b->rc = exec(reqs);
        if (b->rc != 0) { b->err = last_error(); if (err_out) *err_out = b->err; }
        {
            std::lock_guard<std::mutex> g(m);
            running = false;
            prev_members.store((uint32_t)reqs.size(), std::memory_order_relaxed);
            st.batches++; st.members += reqs.size(); if (reqs.size() > st.max_members) st.max_members = reqs.size();
            st.lingered = n_lingered.load(std::memory_order_relaxed);
            if (!open.empty()) futex_set_and_wake_all(&open.front()->go);
        }
        if (reqs.size() > 1) futex_set_and_wake_all(&b->done);
        else b->done.store(1u, std::memory_order_release);
        return b->rc;
    }

    CombinerStats stats() {
        std::lock_guard<std::mutex> g(m);
        return st;
    }
    void reset_stats() {
        std::lock_guard<std::mutex> g(m);
        st = CombinerStats();
        n_lingered.store(0);
    }

private:
    std::mutex m;
    std::deque<std::shared_ptr<CoBatch>> open;     // batches not yet started, oldest first (the front one may hold `go`)
    bool running = false;
    std::atomic<uint32_t> prev_members{1};
    std::atomic<uint32_t> n_lingered{0};
    CombinerStats st;
};

}  // namespace shodh

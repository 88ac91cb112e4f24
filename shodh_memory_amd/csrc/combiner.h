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
//   * linger: a leader that may start waits up to `linger_us` (30 us against a 150-400 us pass; a quarter of the last pass if that is more)
//     while its batch is smaller than what it can expect: (a) if it had to wait for a running pass, the callers of that pass are on their
//     way back -- target = their number + the members that were waiting with it when that pass ended (the closed-loop estimate); (b) the
//     largest of the last four passes (the first caller back after a shared pass waits for the others). Without (a) two callers alternate
//     alone for ever (each finds the other's pass running and never a partner); without (b) the callers split into the leader -- who needs
//     no wake-up -- and everyone else. A caller that is alone never lingers: it never waits for a pass and its last four passes had one
//     member; after 2 ms without a pass nobody lingers (sporadic callers).
//
// The front changes WHEN work runs, never WHAT is computed: the executor must produce, for every member, the bytes the member's own call
// would have produced (searches: rows of a batched pass are bit-identical to single-query passes, tests/test_concurrent_gpu.py; encodes:
// quant_scope PER_TEXT makes a batch N x encode() by construction).
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
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
    std::atomic<uint32_t> members{0}, units_now{0}, expect{0};
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
            b->units_now.store(b->units, std::memory_order_release);
            b->members.fetch_add(1u, std::memory_order_release);
        }
        if (!leader) {
            futex_wait_nonzero(&b->done);
            if (b->rc != 0 && err_out) *err_out = b->err;
            return b->rc;
        }
        futex_wait_nonzero(&b->go);
        // linger (see the header)
        if (linger_us) {
            uint32_t target = std::max<uint32_t>(want_members.load(std::memory_order_relaxed), b->expect.load(std::memory_order_relaxed));
            const uint32_t here = b->units_now.load(std::memory_order_acquire);
            if (target > max_units) target = max_units;
            const uint64_t now = mono_ns();
            if (now - last_end_ns.load(std::memory_order_relaxed) > 2000000ull) target = 0;
            if (here < target) {
                // at most linger_us, or a quarter of the last pass if that is more: what merging two half-size passes saves is a whole pass, and a
                // sleeper's way back through the kernel takes 5-10 us on bare metal but 50+ us inside a VM
                const uint64_t cap_ns = std::max<uint64_t>((uint64_t)linger_us * 1000ull, last_pass_ns.load(std::memory_order_relaxed) / 4);
                const uint64_t t_end = now + cap_ns;
                while (b->units_now.load(std::memory_order_acquire) < target && mono_ns() < t_end) cpu_relax();
                n_lingered.fetch_add(1u, std::memory_order_relaxed);
            }
        }
        std::vector<void *> reqs;
        uint32_t units_run = 0;
        {
            std::lock_guard<std::mutex> g(m);
            b->closed = true;
            reqs = b->reqs;                 // (copied: the pass runs without the lock)
            units_run = b->units;
            open.pop_front();               // b is the front: batches start in the order they were opened
            running = true;
        }
        const uint64_t t_pass0 = mono_ns();
        b->rc = exec(reqs);
        last_pass_ns.store(mono_ns() - t_pass0, std::memory_order_relaxed);
        if (b->rc != 0) { b->err = last_error(); if (err_out) *err_out = b->err; }
        // the members first (their way back through the kernel is the longest part of the cycle), then the next leader
        if (reqs.size() > 1) futex_set_and_wake_all(&b->done);
        else b->done.store(1u, std::memory_order_release);
        std::shared_ptr<CoBatch> next;
        {
            std::lock_guard<std::mutex> g(m);
            running = false;
            // what later leaders wait for: (b) the largest of the last four passes; (a) for the batch that waited behind this pass, this pass's callers
            // plus the ones waiting in it now
            recent[recent_at++ & 3u] = (uint32_t)units_run;
            want_members.store(std::max(std::max(recent[0], recent[1]), std::max(recent[2], recent[3])), std::memory_order_relaxed);
            if (!open.empty()) open.front()->expect.store((uint32_t)units_run + open.front()->units, std::memory_order_relaxed);
            last_end_ns.store(mono_ns(), std::memory_order_relaxed);
            st.batches++; st.members += reqs.size(); if (reqs.size() > st.max_members) st.max_members = reqs.size();
            st.lingered = n_lingered.load(std::memory_order_relaxed);
            if (!open.empty()) next = open.front();
        }
        if (next) futex_set_and_wake_all(&next->go);      // (outside the lock: the wake-up is a system call, and the woken leader wants the lock)
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
    std::atomic<uint32_t> want_members{1};
    uint32_t recent[4] = {1, 1, 1, 1}, recent_at = 0;      // units of the last four passes (guarded by m)
    std::atomic<uint64_t> last_end_ns{0}, last_pass_ns{0};
    std::atomic<uint32_t> n_lingered{0};
    CombinerStats st;
};

}  // namespace shodh

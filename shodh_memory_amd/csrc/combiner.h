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
//     to the members' own output pointers, wakes its members (a tree over eight futex words, see CoBatch::done) and hands `go` to the next batch's leader;
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
#include <sched.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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

// sleep until *word != 0 (a short poll first: a wake-up through the kernel costs 5-10 us, a pass 150-400 us).
// Waking is a TREE: the finishing leader wakes WAKE_ROOT sleepers and every sleeper that was woken wakes WAKE_FAN more before it goes on -- one thread
// waking 63 sleepers with a single FUTEX_WAKE spends ~1.5 us per sleeper inside that system call, so the last caller of a 64-caller pass used to
// come back ~90 us after the first (measured: the next leader lingered 87 us to collect 62.8 of 64); with the tree the last one is ~5 levels away.
constexpr int WAKE_ROOT = 1, WAKE_FAN = 2;      // (per shard of CoBatch::done)
static inline void futex_wait_nonzero(std::atomic<uint32_t> *word, int spin = 64, bool relay = false) {
    for (int i = 0; i < spin; ++i) {
        if (word->load(std::memory_order_acquire) != 0) return;
        cpu_relax();
    }
    bool slept = false;
    while (word->load(std::memory_order_acquire) == 0) { syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAIT_PRIVATE, 0u, nullptr, nullptr, 0); slept = true; }
    if (slept && relay) syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE_PRIVATE, WAKE_FAN, nullptr, nullptr, 0);
}
static inline void futex_set_and_wake_all(std::atomic<uint32_t> *word) {
    word->store(1u, std::memory_order_release);
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
// (sleepers use futex_wait_nonzero(.., relay = true))
static inline void futex_set_and_wake_tree(std::atomic<uint32_t> *word) {
    word->store(1u, std::memory_order_release);
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE_PRIVATE, WAKE_ROOT, nullptr, nullptr, 0);
}

// The front's own lock. Its critical sections are ~50-100 ns (append a pointer, pop a batch), but 60 callers come back from a pass within a few
// microseconds of each other. Behind a pthread mutex the losers sleep in the kernel and are woken one per unlock -- a convoy that took ~80 us to
// let 57 callers rejoin. A test-and-test-and-set lock lets them through in ~0.5-1 us each and does not care when a waiter is descheduled. Measured
// alternatives (64 callers, 1M x 384, k = 120): an MCS queue lock hands over fastest (all 63 rejoined within 40 us instead of 60) but with 256 callers
// on 256 hardware threads one preempted waiter stalled the queue behind it (p99 38 ms); exponential back-off made the rejoin SLOWER (75 us).
class SpinLock {
public:
    void lock() {
        for (uint32_t spins = 0;;) {
            if (!f.load(std::memory_order_relaxed) && !f.exchange(true, std::memory_order_acquire)) return;
            cpu_relax();
            if (++spins > 4096) { sched_yield(); spins = 0; }      // (an oversubscribed machine: the holder is not running)
        }
    }
    void unlock() { f.store(false, std::memory_order_release); }
private:
    std::atomic<bool> f{false};
};
struct LockGuard {
    SpinLock &l;
    explicit LockGuard(SpinLock &lk) : l(lk) { l.lock(); }
    ~LockGuard() { l.unlock(); }
};

struct CoBatch {
    std::vector<void *> reqs;              // members' requests in arrival order (the leader's first); guarded by Combiner::m until closed
    std::atomic<uint32_t> members{0}, units_now{0}, expect{0};
    uint32_t units = 0;                    // queries / texts held
    bool closed = false;
    std::atomic<uint32_t> go{0};           // the leader may start (previous batch done)
    // results are in the members' buffers. SHARDED: member i sleeps on word i % DONE_SHARDS -- all sleepers of one futex word are queued behind one
    // kernel hash-bucket lock and leave it one after the other (~1.1 us each, measured: the 63 members of a pass rejoined evenly spread over 70 us
    // whether they were woken by one call, by a tree, or polled part of the time); eight words are eight independent queues
    static constexpr uint32_t DONE_SHARDS = 8;
    struct alignas(64) DoneWord { std::atomic<uint32_t> v{0}; };
    DoneWord done[DONE_SHARDS];
    std::vector<uint64_t> arrive_ns;       // SHODH_COALESCE_TRACE: when each member joined
    int rc = 0;
    std::string err;
};

struct CombinerStats { uint64_t batches = 0, members = 0, max_members = 0, lingered = 0, exec_ns = 0, linger_ns = 0; };

class Combiner {
public:
    // (atomics: *_set_coalesce may change them while leaders read them)
    std::atomic<uint32_t> linger_us{30}, quiet_us{12};
    std::atomic<bool> trace{false};       // SHODH_COALESCE_TRACE=1: print when each member of a pass joined
    // (A predictive wait -- members wake before the expected end of their pass and poll -- was built and measured in round 5: no gain at 64 callers, -40 % at 256
    // callers on 256 hardware threads, ~10 % of a core per waiting caller. Removed)

    // exec(reqs) runs ONE device pass for all requests (reqs[0] is the leader's own), writes every member's outputs and returns the status shared by all
    // of them; last_error() is the leader thread's message, copied to every member through *err_out when the status is not 0. Returns the batch status.
    // A request must be valid on its own (the callers screen their inputs before they submit): one member's bad input must not fail the others.
    template <class Exec, class ErrFn>
    int submit(void *req, uint32_t units, uint32_t max_units, Exec &&exec, ErrFn &&last_error, std::string *err_out) {
        std::shared_ptr<CoBatch> b;
        bool leader = false;
        uint32_t my_index = 0;
        // a batch object in case this caller has to open one: allocated OUTSIDE the lock (the hint is only a hint; the common case under load is to join)
        std::shared_ptr<CoBatch> fresh;
        if (!joinable_hint.load(std::memory_order_relaxed)) { fresh = std::make_shared<CoBatch>(); fresh->reqs.reserve(64); }
        {
            LockGuard g(m);
            if (!open.empty() && !open.back()->closed && open.back()->units + units <= max_units) {
                b = open.back();
            } else {
                if (!fresh) { fresh = std::make_shared<CoBatch>(); fresh->reqs.reserve(64); }       // (stale hint: rare)
                b = fresh;
                leader = true;
                if (!running && open.empty()) b->go.store(1u, std::memory_order_relaxed);
                open.push_back(b);
                joinable_hint.store(true, std::memory_order_relaxed);
            }
            my_index = (uint32_t)b->reqs.size();
            b->reqs.push_back(req);
            if (trace.load(std::memory_order_relaxed)) b->arrive_ns.push_back(mono_ns());
            b->units += units;
            b->units_now.store(b->units, std::memory_order_release);
            b->members.fetch_add(1u, std::memory_order_release);
        }
        if (!leader) {
            futex_wait_nonzero(&b->done[my_index % CoBatch::DONE_SHARDS].v, 64, true);
            if (b->rc != 0 && err_out) *err_out = b->err;
            return b->rc;
        }
        futex_wait_nonzero(&b->go);
        // linger (see the header)
        const uint32_t linger = linger_us.load(std::memory_order_relaxed), quiet = quiet_us.load(std::memory_order_relaxed);
        if (linger) {
            uint32_t target = std::max<uint32_t>(want_members.load(std::memory_order_relaxed), b->expect.load(std::memory_order_relaxed));
            const uint32_t here = b->units_now.load(std::memory_order_acquire);
            if (target > max_units) target = max_units;
            const uint64_t now = mono_ns();
            if (now - last_end_ns.load(std::memory_order_relaxed) > 2000000ull) target = 0;
            if (here < target) {
                // at most linger_us, or a quarter of the last pass if that is more: what merging two half-size passes saves is a whole pass, and a
                // sleeper's way back through the kernel takes 5-10 us on bare metal but 50+ us inside a VM ...
                const uint64_t cap_ns = std::max<uint64_t>((uint64_t)linger * 1000ull, last_pass_ns.load(std::memory_order_relaxed) / 4);
                const uint64_t t_end = now + cap_ns;
                // ... but the quarter of a pass only once somebody HAS arrived: with nobody else around (a lone caller within 2 ms of a burst on an index whose
                // passes take milliseconds) the wait ends after linger_us -- "a caller that is alone is not delayed" (it used to spin for pass / 4)
                const uint64_t t_end_alone = now + (uint64_t)linger * 1000ull;
                // ... and once callers have started to arrive, no longer than quiet_us after the last arrival: the stragglers of a big pass (a thread
                // that was descheduled, a caller that went away) are not worth a quarter of a pass
                uint32_t seen = here;
                uint64_t t_last = 0;
                for (;;) {
                    const uint32_t u = b->units_now.load(std::memory_order_acquire);
                    const uint64_t t = mono_ns();
                    if (u >= target || t >= t_end) break;
                    if (u != seen) { seen = u; t_last = t; }
                    else if (!t_last && t >= t_end_alone) break;
                    else if (quiet && t_last && t - t_last > (uint64_t)quiet * 1000ull) break;
                    cpu_relax();
                }
                n_lingered.fetch_add(1u, std::memory_order_relaxed);
                linger_ns.fetch_add(mono_ns() - now, std::memory_order_relaxed);
            }
        }
        std::vector<void *> reqs;
        uint32_t units_run = 0;
        {
            LockGuard g(m);
            b->closed = true;
            reqs = b->reqs;                 // (copied: the pass runs without the lock)
            if (b->arrive_ns.size() > 1) {
                // diagnostics: when the members joined, relative to the end of the previous pass (us): first, median, last; then when the pass started
                std::vector<uint64_t> a = b->arrive_ns;
                std::sort(a.begin(), a.end());
                const uint64_t e0 = last_end_ns.load(std::memory_order_relaxed);
                auto rel = [&](uint64_t t) { return ((double)t - (double)e0) / 1e3; };
                fprintf(stderr, "[combiner] pass of %zu: joined %+.1f / %+.1f / %+.1f us after the previous pass ended, started at %+.1f (previous pass %.1f us)\n", a.size(), rel(a.front()),
                        rel(a[a.size() / 2]), rel(a.back()), rel(mono_ns()), (double)last_pass_ns.load(std::memory_order_relaxed) / 1e3);
            }
            units_run = b->units;
            open.pop_front();               // b is the front: batches start in the order they were opened
            joinable_hint.store(!open.empty(), std::memory_order_relaxed);
            running = true;
        }
        const uint64_t t_pass0 = mono_ns();
        b->rc = exec(reqs);
        const uint64_t pass_ns = mono_ns() - t_pass0;
        last_pass_ns.store(pass_ns, std::memory_order_relaxed);
        if (b->rc != 0) { b->err = last_error(); if (err_out) *err_out = b->err; }
        // the members first (their way back through the kernel is the longest part of the cycle), then the next leader
        for (uint32_t sh = 0; sh < CoBatch::DONE_SHARDS; ++sh) b->done[sh].v.store(1u, std::memory_order_release);
        // one sleeper per shard from here, the rest through the sleepers themselves (member 0 is this thread: shard i has sleepers if the pass had more than i + (i == 0) members)
        for (uint32_t sh = 0; sh < CoBatch::DONE_SHARDS; ++sh) {
            const uint32_t first_sleeper = sh == 0 ? CoBatch::DONE_SHARDS : sh;       // member indices of shard sh: sh, sh + 8, ... (member 0 is this thread)
            if (first_sleeper < reqs.size()) syscall(SYS_futex, reinterpret_cast<uint32_t *>(&b->done[sh].v), FUTEX_WAKE_PRIVATE, WAKE_ROOT, nullptr, nullptr, 0);
        }
        std::shared_ptr<CoBatch> next;
        {
            LockGuard g(m);
            running = false;
            // what later leaders wait for: (b) the largest of the last four passes; (a) for the batch that waited behind this pass, this pass's callers
            // plus the ones waiting in it now
            recent[recent_at++ & 3u] = (uint32_t)units_run;
            want_members.store(std::max(std::max(recent[0], recent[1]), std::max(recent[2], recent[3])), std::memory_order_relaxed);
            if (!open.empty()) open.front()->expect.store((uint32_t)units_run + open.front()->units, std::memory_order_relaxed);
            last_end_ns.store(mono_ns(), std::memory_order_relaxed);
            st.batches++; st.members += reqs.size(); if (reqs.size() > st.max_members) st.max_members = reqs.size();
            st.lingered = n_lingered.load(std::memory_order_relaxed);
            st.exec_ns += pass_ns; st.linger_ns = linger_ns.load(std::memory_order_relaxed);
            if (!open.empty()) next = open.front();
        }
        if (next) futex_set_and_wake_all(&next->go);      // (outside the lock: the wake-up is a system call, and the woken leader wants the lock)
        return b->rc;
    }

    CombinerStats stats() {
        LockGuard g(m);
        return st;
    }
    void reset_stats() {
        LockGuard g(m);
        st = CombinerStats();
        n_lingered.store(0); linger_ns.store(0);
    }

private:
    SpinLock m;
    std::deque<std::shared_ptr<CoBatch>> open;     // batches not yet started, oldest first (the front one may hold `go`)
    bool running = false;
    std::atomic<uint32_t> want_members{1};
    std::atomic<bool> joinable_hint{false};
    uint32_t recent[4] = {1, 1, 1, 1}, recent_at = 0;      // units of the last four passes (guarded by m)
    std::atomic<uint64_t> last_end_ns{0}, last_pass_ns{0};
    std::atomic<uint32_t> n_lingered{0};
    std::atomic<uint64_t> linger_ns{0};
    CombinerStats st;
};

}  // namespace shodh

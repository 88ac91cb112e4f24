// Host-only test of the coalescing front (shodh_memory_amd/csrc/combiner.h): built with g++ by tests/test_combiner_cpu.py.
// A fake "device pass" takes PASS_US microseconds whatever the number of requests, squares every request's input into its own output.
// Checks: every caller gets ITS result; a lone caller is never batched or delayed; T closed-loop callers converge to passes of ~T members;
// a failing pass hands its status and message to every member; batches never exceed max_units.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "combiner.h"

using namespace shodh;

struct Req { int in; int out; };
static std::atomic<int> g_passes{0}, g_fail_next{0};
static std::atomic<uint32_t> g_largest{0};

static int run_pass(const std::vector<void *> &reqs, int pass_us) {
    g_passes.fetch_add(1);
    uint32_t n = (uint32_t)reqs.size(), cur = g_largest.load();
    while (n > cur && !g_largest.compare_exchange_weak(cur, n)) {}
    std::this_thread::sleep_for(std::chrono::microseconds(pass_us));       // (a sleep like a device wait; a busy wait makes this VM park the other threads for milliseconds)
    if (g_fail_next.exchange(0)) return -3;
    for (void *r : reqs) { Req *q = static_cast<Req *>(r); q->out = q->in * q->in; }
    return 0;
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16, CALLS = argc > 2 ? atoi(argv[2]) : 200, PASS_US = argc > 3 ? atoi(argv[3]) : 200;
    const uint32_t MAXU = argc > 4 ? (uint32_t)atoi(argv[4]) : 256u;
    Combiner co;
    std::atomic<int> wrong{0}, failed{0};
    // 1. a lone caller: one pass per call, never lingers
    {
        const uint64_t t0 = mono_ns();
        for (int i = 0; i < 50; ++i) {
            Req r{i, -1};
            std::string err;
            int rc = co.submit(&r, 1, MAXU, [&](const std::vector<void *> &v) { return run_pass(v, 50); }, []() { return std::string("x"); }, &err);
            if (rc != 0 || r.out != i * i) wrong++;
        }
        const double us = (mono_ns() - t0) / 1e3 / 50;
        CombinerStats s = co.stats();
        printf("solo: %.1f us per call, passes %llu calls %llu largest %llu lingered %llu\n", us, (unsigned long long)s.batches, (unsigned long long)s.members,
               (unsigned long long)s.max_members, (unsigned long long)s.lingered);
        if (s.batches != 50 || s.max_members != 1 || s.lingered != 0) { printf("FAIL solo\n"); return 1; }
        co.reset_stats();
        g_passes = 0; g_largest = 0;
    }
    // 2. T closed-loop callers
    std::vector<std::thread> th;
    const uint64_t t0 = mono_ns();
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (int i = 0; i < CALLS; ++i) {
                Req r{t * 100000 + i, -1};
                std::string err;
                int rc = co.submit(&r, 1, MAXU, [&](const std::vector<void *> &v) { return run_pass(v, PASS_US); }, []() { return std::string("device lost"); }, &err);
                if (rc != 0) { failed++; if (err != "device lost") wrong++; continue; }
                if (r.out != r.in * r.in) wrong++;
            }
        });
    // one failing pass somewhere in the middle
    std::this_thread::sleep_for(std::chrono::microseconds(PASS_US * 5));
    g_fail_next = 1;
    for (auto &t : th) t.join();
    const double wall = (mono_ns() - t0) / 1e9;
    CombinerStats s = co.stats();
    printf("threads %d calls %d pass_us %d: wall %.3f s, %.0f calls/s, passes %llu, mean members %.2f, largest %llu, lingered %llu, failed calls %d\n", T, CALLS, PASS_US, wall,
           T * CALLS / wall, (unsigned long long)s.batches, (double)s.members / (double)s.batches, (unsigned long long)s.max_members, (unsigned long long)s.lingered, failed.load());
    if (wrong.load()) { printf("FAIL wrong results %d\n", wrong.load()); return 1; }
    if (s.members != (uint64_t)T * CALLS) { printf("FAIL calls served %llu\n", (unsigned long long)s.members); return 1; }
    if (s.max_members > MAXU) { printf("FAIL pass larger than max_units\n"); return 1; }
    if (failed.load() < 1) { printf("FAIL the failing pass reached nobody\n"); return 1; }
    printf("mean_members %.3f\n", (double)s.members / (double)s.batches);
    printf("OK\n");
    return 0;
}

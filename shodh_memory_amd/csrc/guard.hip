// guard.hip -- the allocation front of the library (guard.h): plain hipMalloc / hipHostMalloc, or, under SHODH_GUARD, one fenced
// virtual-memory mapping per allocation. Nothing here launches a kernel.
#include "guard.h"

#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../include/shodh_hip.h"

namespace shodh {

namespace {

struct Block {
    char *va = nullptr;        // start of the reserved range (first fence page)
    size_t reserved = 0;       // bytes reserved (fences included)
    char *map = nullptr;       // start of the mapped part
    size_t mapped = 0;
    size_t bytes = 0;          // what the caller asked for
    bool host = false;
    int device = 0;
};

std::mutex g_mu;
std::map<uintptr_t, Block> g_live;          // user pointer -> block
std::atomic<int> g_mode{-1};
FILE *g_log = nullptr;
std::atomic<uint64_t> g_allocs{0}, g_frees{0}, g_live_bytes{0};

int mode_now() {
    int m = g_mode.load(std::memory_order_acquire);
    if (m >= 0) return m;
    std::lock_guard<std::mutex> g(g_mu);
    m = g_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char *e = getenv("SHODH_GUARD");
    m = e ? atoi(e) : 0;
    if (m < 0 || m > 3) m = 0;
    if (m) {
        if (const char *lp = getenv("SHODH_GUARD_LOG")) g_log = fopen(lp, "a");
        fprintf(stderr, "[shodh guard] mode %d: every allocation is a fenced mapping (%s); diagnostic build of the allocator, not for measurements\n", m,
                m == 2 ? "start-aligned: underruns fault" : m == 3 ? "end-aligned to 256 B: overruns fault" : "end-aligned to 16 B: overruns fault");
        fflush(stderr);
    }
    g_mode.store(m, std::memory_order_release);
    return m;
}

void log_line(const char *what, const Block &b, const void *user, const char *file, int line) {
    if (!g_log) return;
    const char *base = file ? strrchr(file, '/') : nullptr;
    fprintf(g_log, "%s %s user=%p bytes=%zu mapped=[%p,%p) reserved=[%p,%p) at %s:%d\n", what, b.host ? "host" : "dev", user, b.bytes, (void *)b.map,
            (void *)(b.map + b.mapped), (void *)b.va, (void *)(b.va + b.reserved), base ? base + 1 : (file ? file : "?"), line);
    fflush(g_log);
}

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

char *place(const Block &b, int mode) {
    if (mode == 2) return b.map;
    const size_t a = mode == 3 ? 256 : 16;
    return b.map + b.mapped - round_up(b.bytes ? b.bytes : 1, a);
}

hipError_t guarded_dev_alloc(void **p, size_t bytes, const char *file, int line, int mode) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran < 4096) gran = 4096;
    Block b;
    b.bytes = bytes;
    b.device = dev;
    b.mapped = round_up(bytes ? bytes : 1, gran);
    b.reserved = b.mapped + 2 * gran;
    void *va = nullptr;
    e = hipMemAddressReserve(&va, b.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    b.va = (char *)va;
    b.map = b.va + gran;
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, b.mapped, &prop, 0);
    if (e != hipSuccess) { hipMemAddressFree(va, b.reserved); return e; }
    e = hipMemMap(b.map, b.mapped, 0, h, 0);
    if (e != hipSuccess) { hipMemRelease(h); hipMemAddressFree(va, b.reserved); return e; }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(b.map, b.mapped, &acc, 1);
    hipMemRelease(h);                      // (the mapping keeps the memory alive until it is unmapped)
    if (e != hipSuccess) { hipMemUnmap(b.map, b.mapped); hipMemAddressFree(va, b.reserved); return e; }
    e = hipMemset(b.map, 0xCB, b.mapped);  // never-written memory reads as 0xCBCB...: -15.6 in fp16, -2.7e7 in f32, 3419130827 as a count
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipMemUnmap(b.map, b.mapped); hipMemAddressFree(va, b.reserved); return e; }
    char *user = place(b, mode);
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_live[(uintptr_t)user] = b;
        log_line("A", b, user, file, line);
    }
    g_allocs++;
    g_live_bytes += b.mapped;
    *p = user;
    return hipSuccess;
}

hipError_t guarded_pin_alloc(void **p, size_t bytes, const char *file, int line, int mode) {
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    Block b;
    b.host = true;
    b.bytes = bytes;
    b.mapped = round_up(bytes ? bytes : 1, page);
    b.reserved = b.mapped + 2 * page;
    void *va = mmap(nullptr, b.reserved, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (va == MAP_FAILED) return hipErrorOutOfMemory;
    b.va = (char *)va;
    b.map = b.va + page;
    if (mprotect(b.map, b.mapped, PROT_READ | PROT_WRITE) != 0) { munmap(va, b.reserved); return hipErrorOutOfMemory; }
    memset(b.map, 0xCB, b.mapped);
    hipError_t e = hipHostRegister(b.map, b.mapped, hipHostRegisterMapped);
    if (e != hipSuccess) { munmap(va, b.reserved); return e; }
    void *dp = nullptr;
    e = hipHostGetDevicePointer(&dp, b.map, 0);
    if (e != hipSuccess || dp != (void *)b.map) {
        // the kernels are handed the HOST address (as with hipHostMalloc): a registration that lives at another device address cannot stand in for it
        fprintf(stderr, "[shodh guard] registered host memory is not device-visible at its own address (%p -> %p, %s): pinned allocations are NOT fenced\n",
                (void *)b.map, dp, hipGetErrorString(e));
        hipHostUnregister(b.map);
        munmap(va, b.reserved);
        return hipHostMalloc(p, bytes, hipHostMallocDefault);
    }
    char *user = place(b, mode);
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_live[(uintptr_t)user] = b;
        log_line("A", b, user, file, line);
    }
    g_allocs++;
    *p = user;
    return hipSuccess;
}

// true: p was a guarded block and has been released
bool guarded_free(void *p, hipError_t *err) {
    Block b;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_live.find((uintptr_t)p);
        if (it == g_live.end()) return false;
        b = it->second;
        g_live.erase(it);
        log_line("F", b, p, nullptr, 0);
    }
    // hipFree / hipHostFree wait for the device before they release: so does this (a lifetime bug that the product's hipFree would hide is not one)
    int cur = 0;
    hipGetDevice(&cur);
    if (!b.host && cur != b.device) hipSetDevice(b.device);
    hipError_t e = hipDeviceSynchronize();
    if (b.host) {
        hipError_t e2 = hipHostUnregister(b.map);
        if (e == hipSuccess) e = e2;
        // the range stays reserved (PROT_NONE, no memory): a stale pointer faults, it never lands in a newer allocation
        mmap(b.va, b.reserved, PROT_NONE, MAP_FIXED | MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    } else {
        hipError_t e2 = hipMemUnmap(b.map, b.mapped);      // the address range is never freed: stale device pointers fault
        if (e == hipSuccess) e = e2;
        g_live_bytes -= b.mapped;
    }
    if (!b.host && cur != b.device) hipSetDevice(cur);
    g_frees++;
    *err = e;
    return true;
}

}  // namespace

int guard_mode() { return mode_now(); }

hipError_t dev_alloc_raw(void **p, size_t bytes, const char *file, int line) {
    const int m = mode_now();
    if (!m) return hipMalloc(p, bytes);
    return guarded_dev_alloc(p, bytes, file, line, m);
}

hipError_t dev_free_raw(void *p) {
    if (!p) return hipSuccess;
    if (mode_now()) {
        hipError_t e = hipSuccess;
        if (guarded_free(p, &e)) return e;
    }
    return hipFree(p);
}

hipError_t pin_alloc_raw(void **p, size_t bytes, const char *file, int line) {
    const int m = mode_now();
    if (!m) return hipHostMalloc(p, bytes, hipHostMallocDefault);
    return guarded_pin_alloc(p, bytes, file, line, m);
}

hipError_t pin_free_raw(void *p) {
    if (!p) return hipSuccess;
    if (mode_now()) {
        hipError_t e = hipSuccess;
        if (guarded_free(p, &e)) return e;
    }
    return hipHostFree(p);
}

}  // namespace shodh

extern "C" {

int shodh_guard_mode(void) { return shodh::guard_mode(); }

int shodh_guard_stats(uint64_t *allocations, uint64_t *frees, uint64_t *live_device_bytes) {
    if (allocations) *allocations = shodh::g_allocs.load();
    if (frees) *frees = shodh::g_frees.load();
    if (live_device_bytes) *live_device_bytes = shodh::g_live_bytes.load();
    return SHODH_OK;
}

// torch.cuda.memory.CUDAPluggableAllocator entry points: under SHODH_GUARD the TESTS' device tensors (queries, rows, result buffers handed to the
// *_device entry points) are fenced mappings too, so that a kernel over-reading a caller's buffer faults like one over-reading the library's own.
void *shodh_guard_torch_alloc(int64_t bytes, int device, void *stream) {
    (void)stream;
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != device) hipSetDevice(device);
    void *p = nullptr;
    hipError_t e = shodh::dev_alloc_raw(&p, bytes > 0 ? (size_t)bytes : 0, "torch", 0);
    if (cur != device) hipSetDevice(cur);
    return e == hipSuccess ? p : nullptr;
}

void shodh_guard_torch_free(void *p, int64_t bytes, int device, void *stream) {
    (void)bytes; (void)device; (void)stream;
    shodh::dev_free_raw(p);
}

}  // extern "C"

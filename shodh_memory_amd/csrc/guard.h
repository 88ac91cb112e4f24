// guard.h -- every device / pinned-host allocation of the library goes through these four calls.
//
// Normally they ARE hipMalloc / hipFree / hipHostMalloc / hipHostFree. With SHODH_GUARD=1|2|3 in the environment (read once, at the first
// allocation) every allocation becomes its own virtual-memory mapping with UNMAPPED pages on both sides, placed so that its last byte is the
// last byte of the mapping (mode 1: 16-byte granularity; mode 3: 256-byte) or its first byte the first (mode 2): a kernel that reads or
// writes past the end (or before the start) of ANY buffer takes a GPU page fault at once, on every run, instead of silently touching a
// neighbour -- or, once in thirty runs, an unmapped page ("Memory access fault by GPU node", cause unknown). Freed ranges are
// unmapped and their addresses never handed out again, so a use after free faults as well; fresh memory is filled with 0xCB so that a read
// of never-written memory shows up as a wrong result. SHODH_GUARD_LOG=<file> appends one line per allocation / free (address range, bytes,
// file:line) for matching a fault address after the process has died (tools/guard_lookup.py).
//
// Diagnostic mode only: an allocation costs a few hundred microseconds and at least one 2 MiB (device) / 4 KiB (host) granule.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace shodh {

hipError_t dev_alloc_raw(void **p, size_t bytes, const char *file, int line);
hipError_t dev_free_raw(void *p);
hipError_t pin_alloc_raw(void **p, size_t bytes, const char *file, int line);
hipError_t pin_free_raw(void *p);
int guard_mode();      // 0 = off

template <class T>
inline hipError_t dev_alloc(T **p, size_t bytes, const char *file = __builtin_FILE(), int line = __builtin_LINE()) {
    return dev_alloc_raw((void **)p, bytes, file, line);
}
inline hipError_t dev_free(void *p) { return dev_free_raw(p); }
template <class T>
inline hipError_t pin_alloc(T **p, size_t bytes, const char *file = __builtin_FILE(), int line = __builtin_LINE()) {
    return pin_alloc_raw((void **)p, bytes, file, line);
}
inline hipError_t pin_free(void *p) { return pin_free_raw(p); }

}  // namespace shodh

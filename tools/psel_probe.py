"""One IVF-PQ search of 1024 queries against a 4096-partition table with a small posting set: the probe-selection kernels alone matter (phase timers with
tools/build_variant.sh prof -DSHODH_PROF; rocprofv3 --kernel-trace for the kernel times)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import shodh_memory_amd as S
f32 = np.float32
rng = np.random.default_rng(2)
P, n = 4096, 60000
def unit(x): return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(f32)
centres = unit(rng.standard_normal((64, 384)))
c = unit(centres[rng.integers(0, 64, P)] + 0.8 * rng.standard_normal((P, 384)).astype(f32))
sizes = rng.multinomial(n, np.ones(P) / P)
off = np.zeros(P + 1, np.uint64); off[1:] = np.cumsum(sizes)
idx = S.SpannIndex(384, num_probes=32)
idx.set_trained_state(c, (rng.standard_normal((48, 256, 8)) * 0.05).astype(f32), off, rng.permutation(n).astype(np.uint32), rng.integers(0, 256, (n, 48), dtype=np.uint8))
q = unit(centres[rng.integers(0, 64, 1024)] + 0.8 * rng.standard_normal((1024, 384)).astype(f32))
import torch
qd = torch.from_numpy(q).cuda()
it = int(os.environ.get("ITERS", 3))
if it <= 3:
    for _ in range(it):
        t = time.perf_counter(); idx.search_batch(q, 10); print("search %.1f us" % ((time.perf_counter() - t) * 1e6), flush=True)
else:      # device pointers, back to back: the stream's time per search
    for _ in range(5): idx.search_batch_device(qd, 10)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): idx.search_batch_device(qd, 10)
    torch.cuda.synchronize(); print("search (device pointers, %d back to back): %.1f us each" % (it, (time.perf_counter() - t) / it * 1e6), flush=True)

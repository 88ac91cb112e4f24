"""Round 6 probe: the contract step issued round-robin on S streams (independent batches in flight) -- what overlapping the small kernels of one step with the
scan of another is worth. usage: python tools/r6_streams_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import shodh_memory_amd as S
dev = torch.device("cuda", 0)
n, k, nq = 1_000_000, int(os.environ.get("K", 10)), 256
rows = bench.synth_rows(torch, n, 384, 1, dev)
qs = [bench.synth_rows(torch, nq, 384, 2 + i, dev) for i in range(8)]
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, scan_mode=2, reserve_rows=n))
idx.build(rows)
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq,), dtype=torch.int32, device=dev)) for _ in range(ns)]
    def step(i):
        with torch.cuda.stream(streams[i % ns]):
            idx.search_batch_device(qs[i % 8], k, out=outs[i % ns])
    for i in range(300): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    N = 2000
    for i in range(N): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / N
    print("streams %d: %.1f us per step, %.0f queries/s" % (ns, dt * 1e6, nq / dt))

import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import shodh_memory_amd as S
dev = torch.device("cuda", 0)
n = int(os.environ.get("ROWS", 1_000_000)); k = int(os.environ.get("K", 10)); nq = int(os.environ.get("NQ", 256)); dim = int(os.environ.get("DIM", 384))
rows = bench.synth_rows(torch, n, dim, 1, dev)
q = bench.synth_rows(torch, nq, dim, 2, dev)
idx = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=2, reserve_rows=n))
idx.build(rows)
for _ in range(3): idx.search_batch_device(q, k)
torch.cuda.synchronize(); idx.kernel_timing(True)
t = time.perf_counter()
ITERS = int(os.environ.get('ITERS', 30))
for _ in range(ITERS): idx.search_batch_device(q, k)
t_enq = (time.perf_counter() - t) / ITERS
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / ITERS
print("host enqueue per step %.1f us" % (t_enq * 1e6))
qh = q.cpu().numpy()
idx.search_batch(qh, k)
print("step %.1f us | emit kernel mean/min us: %s | stats %s" % (dt * 1e6, idx.kernel_timing(True)[:2], idx.scan_stats()))

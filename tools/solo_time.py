"""Single-query calls on the contract corpus (1M x 384, 5 % tombstoned): scan-kernel event times and host-call percentiles. K=10|120 REPS=400"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import shodh_memory_amd as S  # noqa: E402
dev = torch.device("cuda:0")
n = int(os.environ.get("ROWS", 1_000_000)); k = int(os.environ.get("K", 10)); reps = int(os.environ.get("REPS", 400))
q = B.synth_rows(torch, 256, 384, B.SEED + 1, dev)
rows = B.synth_rows(torch, n, 384, B.SEED, dev, adversarial_queries=q)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=n))
idx.build(rows)
B.tombstone(torch, idx, n, 0.05, B.SEED + 2, dev)
idx.set_coalesce(False)
qh = q.cpu().numpy()
for i in range(20): idx.search_batch(qh[i:i + 1], k)
torch.cuda.synchronize(); idx.kernel_timing(True)
ts = []
for i in range(reps):
    a = time.perf_counter(); idx.search_batch(qh[i % 256:i % 256 + 1], k); ts.append(time.perf_counter() - a)
ts.sort()
m, mn, c = idx.kernel_timing(True)
print("k %3d: scan kernel mean %.1f min %.1f us over %d | host call p50 %.1f p10 %.1f min %.1f us" % (k, m, mn, c, ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6, ts[0] * 1e6))

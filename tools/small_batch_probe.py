"""How long ONE host-pointer search of nq queries takes on the contract corpus (1M x 384, 5 % tombstoned), nq = 1 ... 256, k = 10 / 120: the pass a
coalesced batch of nq callers runs. usage: python tools/small_batch_probe.py [only_nq only_k reps]   (with arguments: just that shape, for rocprofv3)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import shodh_memory_amd as S  # noqa: E402

dev = torch.device("cuda:0")
n = int(os.environ.get("ROWS", 1_000_000))
q = B.synth_rows(torch, 256, 384, B.SEED + 1, dev)
rows = B.synth_rows(torch, n, 384, B.SEED, dev, adversarial_queries=q)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=n))
idx.build(rows)
B.tombstone(torch, idx, n, 0.05, B.SEED + 2, dev)
idx.set_coalesce(False)
qh = q.cpu().numpy()
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(nq, k) for k in (120, 10) for nq in (1, 2, 4, 8, 16, 32, 64, 128, 256)]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for nq, k in shapes:
    for _ in range(5):
        idx.search_batch(qh[:nq], k)
    ts = []
    for i in range(reps):
        a = time.perf_counter(); idx.search_batch(qh[:nq], k); ts.append(time.perf_counter() - a)
    ts.sort()
    st = idx.stage_timings_us()
    print("nq %3d k %3d: host call p50 %.1f us (min %.1f) | device stages scan %.1f select %.1f other %.1f total %.1f | %s" % (
        nq, k, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, st["scan"], st["select"], st["other"], st["total"], {kk: int(v) for kk, v in idx.scan_stats().items()}), flush=True)

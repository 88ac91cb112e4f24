"""Kernel time of the single-query scan at 1M x 384: back-to-back device-pointer calls (the GPU never idles) against host-pointer calls
(idle between calls), from the event pair attached to the scan kernel."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import shodh_memory_amd as S  # noqa: E402
from tests import synth  # noqa: E402

n = 1_000_000
q = synth.queries(8)
rows = synth.corpus(n, queries=q)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384))
idx.build(rows)
dq = torch.from_numpy(q).cuda()
for i in range(10):
    idx.search_batch(dq[:1], 10)
torch.cuda.synchronize()
idx.kernel_timing()
for i in range(200):
    idx.search_batch(dq[i % 8:i % 8 + 1], 10)
torch.cuda.synchronize()
m, mn, c = idx.kernel_timing()
out = {"back_to_back_us_mean": round(m, 2), "min": round(mn, 2), "count": c}
for i in range(100):
    idx.search_batch(q[i % 8:i % 8 + 1], 10)
m, mn, c = idx.kernel_timing()
out.update({"host_calls_us_mean": round(m, 2), "host_min": round(mn, 2)})
print(json.dumps(out), flush=True)

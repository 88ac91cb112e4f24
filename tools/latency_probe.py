"""Single-query latency of the flat index at 1M x 384 (host-pointer API, wall clock around shodh_index_search): p50 / p95 over 300 calls,
for k = 10 and k = 120, plus the device-side stage timings of the last call. SHODH_SOLO=0 switches the single-pass scan off."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import shodh_memory_amd as S  # noqa: E402
from tests import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
q = synth.queries(64)
rows = synth.corpus(n, queries=q)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384))
idx.build(rows)
idx.mark_deleted_many(np.nonzero(synth.tombstones(n))[0].astype(np.uint32))
for k in (10, 120):
    for i in range(20):
        idx.search_batch(q[i % 64:i % 64 + 1], k)
    ts = []
    for i in range(300):
        t0 = time.perf_counter()
        idx.search_batch(q[i % 64:i % 64 + 1], k)
        ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e3
    print(json.dumps({"rows": n, "k": k, "p50_ms": round(float(ts[150]), 4), "p95_ms": round(float(ts[285]), 4), "min_ms": round(float(ts[0]), 4),
                      "stage_us": idx.stage_timings_us(), "stats": idx.scan_stats()}), flush=True)

"""diagnostics: graph-mode insert and ANN search rates (python tools/graph_probe.py [n])"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import shodh_memory_amd as S
from shodh_memory_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dim = 384
rng = np.random.default_rng(0)
x = rng.standard_normal((n, dim)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
idx = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=L.SCAN_GRAPH, reserve_rows=n))
idx.add_vectors(x[:64])
t = time.time(); idx.add_vectors(x[64:]); dt = time.time() - t
q = rng.standard_normal((256, dim)).astype(np.float32)
idx.search_batch(q, 10)
t = time.time()
for _ in range(5): idx.search_batch(q, 10)
ds = (time.time() - t) / 5
t = time.time()
for i in range(50): idx.search(q[i], 10)
d1 = (time.time() - t) / 50
print(json.dumps({"n": n, "insert_us_each": round(dt / (n - 64) * 1e6, 1), "search_b256_ms": round(ds * 1e3, 3), "search_qps_b256": round(256 / ds, 1), "search_single_us": round(d1 * 1e6, 1)}))

"""diagnostics: ms per search of 1M x DIM rows at several batch sizes (python tools/bigdim_probe.py 1024 64 128 256)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
import shodh_memory_amd as S
dev = torch.device("cuda", 0)
dim = int(sys.argv[1]); n = 1_000_000
rows = bench.synth_rows(torch, n, dim, 77, dev)
idx = S.VamanaIndex(S.VamanaConfig(dimension=dim, reserve_rows=n)); idx.build(rows)
for b in [int(x) for x in sys.argv[2:]]:
    q = bench.synth_rows(torch, b, dim, 78, dev)
    out = (torch.empty((b, 10), dtype=torch.int32, device=dev), torch.empty((b, 10), dtype=torch.float32, device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
    idx.kernel_timing(reset=True)
    dt = bench.timed_steps(torch, lambda i: idx.search_batch_device(q, 10, out=out), 10, 3)
    km, kmin, kn = idx.kernel_timing(reset=True)
    print(json.dumps({"dim": dim, "batch": b, "ms": round(dt * 1e3, 4), "scan_kernel_us": round(km, 1)}))

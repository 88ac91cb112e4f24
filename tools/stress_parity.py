"""Randomised stress: many recall batches of random shapes through the MFMA pre-scan path, compared bit for bit with
the exact-order scan path (itself checked against the CPU oracle by tests/) on the same device.
usage: python tools/stress_parity.py [rounds] [rows] [dims, e.g. 384,128,768,1024]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import bench
import shodh_memory_amd as S
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
rng = np.random.default_rng(123)
bad = 0
dims = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [384, 128]
for dim in dims:
    rows = bench.synth_rows(torch, n, dim, 7, dev)
    rows[1000:1040] = rows[5]                                    # duplicates
    a = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=2, reserve_rows=n)); a.build(rows)
    b = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=1, reserve_rows=n)); b.build(rows)
    for i in rng.choice(n, 500, replace=False):
        a.mark_deleted(int(i)); b.mark_deleted(int(i))
    for r in range(rounds):
        nq = int(rng.choice([1, 3, 32, 200, 256, 257, 600]))
        k = int(rng.choice([1, 10, 10, 10, 50, 120]))
        q = bench.synth_rows(torch, nq, dim, 1000 + r, dev)
        if r % 7 == 0:
            q[0] = rows[int(rng.integers(n))]                    # a query equal to a row
        ia, da, ca = a.search_batch_device(q, k)
        ib, db, cb = b.search_batch_device(q, k)
        torch.cuda.synchronize()
        if not (torch.equal(ia, ib) and torch.equal(da.view(torch.int32), db.view(torch.int32)) and torch.equal(ca, cb)):
            bad += 1
            print("MISMATCH dim %d round %d nq %d k %d" % (dim, r, nq, k), flush=True)
print("stress: %d rounds x dims %s, mismatches: %d" % (rounds, dims, bad))
sys.exit(1 if bad else 0)

"""Randomised stress of the HOST-pointer search paths (round 5): batches of 1 ... 200 queries through shodh_index_search -- the single-query zero-copy path, the
few-query zero-copy pass (queries read from / rows written into pinned host memory, statistics mirrored by the final stage's last workgroup), the copy path above
128 queries -- compared bit for bit with the exact-order scan through device pointers on the same device (itself checked against the CPU oracle by tests/).
usage: python tools/stress_host_parity.py [rounds] [rows]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import bench
import shodh_memory_amd as S
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
rng = np.random.default_rng(321)
bad = 0
for dim in (384, 768):
    rows = bench.synth_rows(torch, n, dim, 7, dev)
    rows[1000:1040] = rows[5]
    a = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=2, reserve_rows=n)); a.build(rows)
    b = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=1, reserve_rows=n)); b.build(rows)
    dead = rng.choice(n, 500, replace=False).astype(np.uint32)
    a.mark_deleted_many(dead); b.mark_deleted_many(dead)
    for r in range(rounds):
        nq = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 33, 64, 100, 128, 129, 200]))
        k = int(rng.choice([1, 10, 10, 50, 120, 300]))
        q = bench.synth_rows(torch, nq, dim, 5000 + r, dev)
        if r % 5 == 0:
            q[0] = rows[int(rng.integers(n))]
        ia, da, ca = a.search_batch(q.cpu().numpy(), k)
        ib, db, cb = b.search_batch_device(q, k)
        torch.cuda.synchronize()
        if not (np.array_equal(ia, ib.cpu().numpy().view(np.uint32)) and da.tobytes() == db.cpu().numpy().tobytes() and np.array_equal(ca, cb.cpu().numpy().view(np.uint32))):
            bad += 1
            print("MISMATCH dim %d round %d nq %d k %d" % (dim, r, nq, k), flush=True)
    a.close(); b.close()
print("host-pointer stress: %d rounds x 2 dims, mismatches %d" % (rounds, bad))
sys.exit(1 if bad else 0)

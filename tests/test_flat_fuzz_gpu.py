"""Randomised flat-index parity: random dimension (incl. sizes no fast path is written for), corpus size, k, accumulation order, scan mode,
tombstones, duplicates -- ids and distances against `oracle.brute_force_search`, bit for bit. SHODH_FUZZ_ROUNDS (default 12) scales it."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def S():
    import shodh_memory_amd as s
    return s


def test_random_flat_workloads(S, oracle):
    from shodh_memory_amd import _lib as L
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "12"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "777")))
    for rnd in range(rounds):
        dim = int(rng.choice([4, 12, 20, 32, 100, 128, 256, 384, 388, 512, 768, 1000, 1024]))
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 5000, 16383, 16384, 20000, 33333]))
        order = int(rng.integers(0, 2))
        mode = int(rng.choice([L.SCAN_AUTO, L.SCAN_EXACT, L.SCAN_MFMA]))
        k = min(int(rng.choice([1, 2, 10, 37, 120, 300, n, n + 5])), 7936)           # 7936: the largest k the selection buffers hold (shodh_hip.h)
        nq = int(rng.choice([1, 2, 9]))
        print("flat fuzz round %d: dim %d n %d order %d mode %d k %d nq %d" % (rnd, dim, n, order, mode, k, nq), flush=True)
        q = synth.queries(nq, dim, seed=3000 + rnd)
        rows = synth.corpus(n, dim, seed=2000 + rnd, queries=q)
        if n > 10 and rng.random() < 0.5:
            rows[rng.integers(0, n, 6)] = rows[1]
        idx = S.VamanaIndex(S.VamanaConfig(dimension=dim, order=order, scan_mode=mode))
        at = 0
        while at < n:                                                  # grown in random pieces: ids stay dense and sequential
            b = int(min(n - at, rng.choice([1, 7, 500, 20000])))
            assert idx.add_vectors(rows[at:at + b]) == at
            at += b
        deleted = None
        if n > 3 and rng.random() < 0.6:
            deleted = np.zeros(n, np.uint8)
            deleted[rng.choice(n, int(rng.integers(1, max(2, n // 2))), replace=False)] = 1
            idx.mark_deleted_many(np.nonzero(deleted)[0].astype(np.uint32))
        ids, dist, counts = idx.search_batch(q, k)
        for i in range(nq):
            e_ids, e_dist = oracle.brute_force_search(rows, q[i], k, deleted, order=order, select=True)
            m = int(counts[i])
            assert m == len(e_ids), (rnd, i, m, len(e_ids))
            assert ids[i, :m].tolist() == e_ids.tolist(), (rnd, i)
            assert dist[i, :m].tobytes() == e_dist.tobytes(), (rnd, i)
        assert idx.extract_all_vectors().tobytes() == rows.tobytes()
        idx.close()

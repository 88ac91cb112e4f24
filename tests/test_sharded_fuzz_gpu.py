"""Randomised parity of the multi-GPU index behind the C ABI (several shards on the one visible GPU, device-copy exchange; RCCL where the
layout allows it): random shard counts, block sizes, dimensions, growth in pieces, tombstones -- against the oracle's brute force over the
same rows with the same GLOBAL ids. SHODH_FUZZ_ROUNDS (default 6) scales it."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    import shodh_memory_amd.distributed as d
    return d


def test_random_sharded_workloads(D, oracle):
    from shodh_memory_amd import _lib as L
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "6"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "99")))
    for rnd in range(rounds):
        shards = int(rng.choice([1, 2, 3, 5, 8]))
        block_log2 = int(rng.choice([6, 7, 10, 16]))                      # (the library accepts 6 .. 26)
        dim = int(rng.choice([64, 128, 384, 768]))
        n = int(rng.choice([1, 50, 1000, 20000, 40000]))
        k = int(rng.choice([1, 10, 120]))
        nq = int(rng.choice([1, 5, 33]))
        order = int(rng.integers(0, 2))
        print("sharded fuzz round %d: shards %d block 2^%d dim %d n %d k %d nq %d order %d" % (rnd, shards, block_log2, dim, n, k, nq, order), flush=True)
        q = synth.queries(nq, dim, seed=8000 + rnd)
        rows = synth.corpus(n, dim, seed=9000 + rnd, queries=q)
        exch = L.EXCHANGE_RCCL if shards == 1 else L.EXCHANGE_COPY
        idx = D.MultiGpuIndex([0] * shards, dim=dim, order=order, block_log2=block_log2, exchange=exch)
        at = 0
        first = True
        while at < n:
            b = int(min(n - at, rng.choice([1, 13, 700, 30000])))
            if first and rng.random() < 0.5:
                idx.build(rows[at:at + b])
            else:
                assert idx.add_vectors(rows[at:at + b]) == at
            first = False
            at += b
        assert idx.len() == n and idx.extract_all_vectors().tobytes() == rows.tobytes()
        deleted = None
        if n > 3 and rng.random() < 0.6:
            deleted = np.zeros(n, np.uint8)
            dead = rng.choice(n, int(rng.integers(1, max(2, n // 2))), replace=False).astype(np.uint32)
            deleted[dead] = 1
            assert idx.mark_deleted_many(dead) == len(dead)
        ids, dist, counts = idx.search_batch(q, k)
        for i in range(nq):
            e_ids, e_dist = oracle.brute_force_search(rows, q[i], k, deleted, order=order, select=True)
            m = int(counts[i])
            assert m == len(e_ids), (rnd, i, m, len(e_ids))
            assert ids[i, :m].tolist() == e_ids.tolist(), (rnd, i)
            assert dist[i, :m].tobytes() == e_dist.tobytes(), (rnd, i)
        idx.close()

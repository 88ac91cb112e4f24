"""Randomised IVF-PQ parity given a trained state: random dimension (8 x sub-quantisers), partitions, probes, k, batch -- device encode
(nearest centroid + PQ codes) and search against the oracle, bit for bit. SHODH_FUZZ_ROUNDS (default 8) scales it."""
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


def test_random_ivfpq_workloads(S, oracle):
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "8"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "4242")))
    for rnd in range(rounds):
        dim = int(rng.choice([8, 16, 64, 128, 384, 392, 512, 768]))
        M = dim // 8
        P = int(rng.choice([1, 2, 17, 64, 200, 600]))
        n = int(max(P, rng.choice([300, 1000, 3000])))
        nprobe = int(rng.choice([1, 5, 20, P, 1000]))                  # (the library probes at most 1024 lists per query)
        k = int(rng.choice([1, 10, 200]))
        nq = int(rng.choice([1, 3, 40]))
        ncent = int(rng.choice([256, 256, 16]))
        print("ivfpq fuzz round %d: dim %d P %d n %d nprobe %d k %d nq %d ncent %d" % (rnd, dim, P, n, nprobe, k, nq, ncent), flush=True)
        rows = synth.corpus(n, dim, seed=6000 + rnd, adversarial=False)
        centroids = rows[rng.choice(n, P, replace=False)].copy()
        codebook = np.stack([rows[rng.choice(n, ncent, replace=False), m * 8:(m + 1) * 8] for m in range(M)]).astype(f32)
        idx = S.SpannIndex(dim, num_probes=nprobe)
        idx.set_trained_state(centroids, codebook, np.zeros(P + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, M), np.uint8))
        assign, codes = idx.encode(rows)
        for i in rng.choice(n, min(n, 60), replace=False):
            assert assign[i] == oracle.spann_find_nearest_centroid(rows[i], centroids), (rnd, i)
            assert codes[i].tolist() == oracle.pq_encode(codebook, rows[i]).tolist(), (rnd, i)
        order = np.argsort(assign, kind="stable")
        off = np.zeros(P + 1, np.uint64); off[1:] = np.cumsum(np.bincount(assign, minlength=P))
        idx.set_trained_state(centroids, codebook, off, order.astype(np.uint32), codes[order])
        q = synth.queries(nq, dim, seed=7000 + rnd)
        ids, dist, counts = idx.search_batch(q, k)
        for i in range(nq):
            e_ids, e_dist = oracle.spann_search(centroids, off, order.astype(np.uint32), codes[order], codebook, nprobe, q[i], k, 0)
            m = int(counts[i])
            assert m == len(e_ids), (rnd, i, m, len(e_ids))
            assert ids[i, :m].tolist() == e_ids.tolist(), (rnd, i)
            assert dist[i, :m].tobytes() == e_dist.tobytes(), (rnd, i)
        idx.close()

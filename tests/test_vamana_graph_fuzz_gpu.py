"""Randomised graph-mode parity: random dimension / max_degree / order / batch sizes / tombstones, device graph and answers against
`oracle.VamanaGraph` after every step. SHODH_FUZZ_ROUNDS (default 6) scales it; SHODH_FUZZ_SEED picks the sequence."""
import os

import numpy as np
import pytest

from tests.test_vamana_graph_gpu import assert_graph_equal, assert_search_equal, gpu_index, unit_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import shodh_memory_amd as s
    return s


def test_random_graph_workloads(S, oracle):
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "6"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "20240927")))
    for rnd in range(rounds):
        dim = int(rng.choice([8, 16, 40, 64, 120, 256, 384, 392, 512, 520, 768]))
        R = int(rng.choice([2, 3, 7, 16, 31, 32, 33, 48, 63, 64, 100]))
        order = int(rng.integers(0, 2))
        Ls = int(rng.choice([R + 1, 20, 75, 120]))
        n = int(rng.integers(60, 500))
        clusters = int(rng.choice([0, 3, 12]))
        print("fuzz round %d: dim %d R %d order %d L %d n %d clusters %d" % (rnd, dim, R, order, Ls, n, clusters), flush=True)
        rows = unit_rows(n, dim, 1000 + rnd, clusters=clusters)
        if rng.random() < 0.5:
            rows[rng.integers(0, n, 5)] = rows[0]                      # exact duplicates: ties decided by id
        idx = gpu_index(S, dim, R, Ls, order)
        g = oracle.VamanaGraph(dim, R=R, L=Ls, order=order, capacity=n)
        at = 0
        while at < n:
            b = int(min(n - at, rng.choice([1, 2, 17, 64, 200])))
            assert idx.add_vectors(rows[at:at + b]) == at
            for r in rows[at:at + b]:
                g.add_vector(r)
            at += b
        assert_graph_equal(idx, g)
        q = np.concatenate([unit_rows(10, dim, 5000 + rnd, clusters=clusters), rows[:3]])
        deleted = None
        if rng.random() < 0.6:
            deleted = np.zeros(n, np.uint8)
            deleted[rng.choice(n, int(rng.integers(1, max(2, n // 3))), replace=False)] = 1
            idx.mark_deleted_many(np.nonzero(deleted)[0].astype(np.uint32))
        for k in (1, int(rng.integers(2, 40))):
            assert_search_equal(idx, g, q, k, deleted=deleted)
        if n >= 120 and rng.random() < 0.5:                             # a repair of the tail (the host mirror's threshold does not apply to the raw call)
            import ctypes as C
            from shodh_memory_amd import _lib as L
            cnt = int(rng.integers(10, n // 2))
            r = C.c_uint32()
            L.check(L.lib().shodh_index_incremental_repair(idx.handle, n - cnt, cnt, C.byref(r)))
            assert int(r.value) == g.incremental_repair(n - cnt)
            assert_graph_equal(idx, g)
            assert_search_equal(idx, g, q[:6], 5, deleted=deleted)
        idx.close()


@pytest.mark.parametrize("dim,R", [(192, 63), (136, 62), (232, 63), (200, 48)])
def test_backedge_rerank_with_wide_lists(S, oracle, dim, R):
    """with max_degree 62 / 63 and 136 <= dim <= 232 (scalar-4 order) the fourth wave's private buffers of the parallel back-edge
    re-rank reached into the neighbour list all four waves were reading. Dense clusters make back-edge lists overflow R early."""
    n = 700
    rows = unit_rows(n, dim, 77, clusters=2)
    idx = gpu_index(S, dim, R, 75, 0)
    g = oracle.VamanaGraph(dim, R=R, L=75, order=0, capacity=n)
    for at in range(0, n, 100):
        assert idx.add_vectors(rows[at:at + 100]) == at
        for r in rows[at:at + 100]:
            g.add_vector(r)
    assert_graph_equal(idx, g)
    assert_search_equal(idx, g, rows[:8], 10)
    idx.close()

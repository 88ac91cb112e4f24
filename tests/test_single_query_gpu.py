"""The single-query scan (`solo_scan_kernel`, scan_mfma.hip): `recall` asks the index ONE query at a time (mod.rs:2875-2904), and for one
query the pre-scan is a single pass with workgroup-local thresholds instead of the sampled threshold of the batch pipeline. Ids and
distances against `oracle.brute_force_search`, bit for bit, on every dimension with a pre-scan, both accumulation orders, tombstones,
crowded and degenerate corpora, the slice cap, and interleaved with batch calls (the path keeps one counter alive between calls)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def S():
    import shodh_memory_amd as s
    return s


def make_index(S, dim=384, order=0, scan_mode=2, **kw):
    return S.VamanaIndex(S.VamanaConfig(dimension=dim, order=order, scan_mode=scan_mode, **kw))


def check_one(oracle, idx, rows, q, k, order, deleted=None):
    ids, dist, counts = idx.search_batch(q[None, :], k)
    e_ids, e_dist = oracle.brute_force_search(rows, q, k, deleted, order=order, select=True)
    n = int(counts[0])
    assert n == len(e_ids), (n, len(e_ids))
    assert ids[0, :n].tolist() == e_ids.tolist(), (ids[0, :n][:12], e_ids[:12])
    assert dist[0, :n].tobytes() == e_dist.tobytes()
    assert (ids[0, n:] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("dim", [128, 256, 384, 512, 768, 1024])
def test_single_query_matches_oracle(S, oracle, dim, order):
    n = 20011
    q = synth.queries(5, dim, seed=50 + dim)
    rows = synth.corpus(n, dim, seed=60 + dim, queries=q)
    deleted = synth.tombstones(n, 0.05, seed=70 + dim)
    idx = make_index(S, dim=dim, order=order)
    idx.build(rows)
    idx.mark_deleted_many(np.nonzero(deleted)[0].astype(np.uint32))
    for i, k in enumerate((10, 1, 32, 5, 10)):
        check_one(oracle, idx, rows, q[i], k, order, deleted)
        st = idx.scan_stats()
        assert st["sampled_rows"] == 0 and st["emitted"] >= k and st["overflowed"] == 0, st      # the single pass ran, nothing went to the exact scan
    check_one(oracle, idx, rows, rows[n // 3], 10, order, deleted)                              # a stored row: itself (or its duplicates) first
    t = idx.stage_timings_us()
    assert t["total"] > 0 and t["scan"] > 0
    # k beyond the local-threshold limit (32): global threshold from the wave maxima (scan + emit + final stage); about k rows are
    # handed over instead of 256 x k
    for i, k in enumerate((33, 120, 300)):
        check_one(oracle, idx, rows, q[i], k, order, deleted)
        st = idx.scan_stats()
        assert st["sampled_rows"] == 0 and k <= st["emitted"] <= 3 * k + 64 and st["overflowed"] == 0, st
    idx.close()


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("n", [65536, 70001, 131103])
def test_streamed_scan_matches_oracle(S, oracle, n, order):
    """From 256 workgroups x 8 tiles on, the 384-d single-query scan reads its rows through an LDS ring of 32-row tiles dealt round-robin over the
    workgroups (solo_scan_kernel, `stream`): a corpus that ends inside a tile, tombstones in the first and the last tile of the corpus and of single
    workgroups, local thresholds (k <= 32) and the global one (k = 33, 120, 300), a stored row as the query."""
    dim = 384
    q = synth.queries(6, dim, seed=150 + n % 97)
    rows = synth.corpus(n, dim, seed=160 + n % 89, queries=q)
    deleted = synth.tombstones(n, 0.05, seed=170)
    deleted[:40] = True; deleted[n - 45:n - 3] = True; deleted[256 * 32:256 * 32 + 32] = True; deleted[n // 2] = False
    idx = make_index(S, dim=dim, order=order)
    idx.build(rows)
    idx.mark_deleted_many(np.nonzero(deleted)[0].astype(np.uint32))
    for i, k in enumerate((10, 1, 32, 33, 120, 300)):
        check_one(oracle, idx, rows, q[i], k, order, deleted)
        st = idx.scan_stats()
        assert st["sampled_rows"] == 0 and st["emitted"] >= k and st["overflowed"] == 0, st
    check_one(oracle, idx, rows, rows[n // 2], 10, order, deleted)
    check_one(oracle, idx, rows, rows[n - 2], 120, order, deleted)        # the corpus' last rows: the partial tile's keys reach the threshold kernel
    idx.close()


def test_streamed_scan_crowded_cone(S, oracle):
    """... and a corpus inside a narrow cone at that size: hundreds of hand-overs per workgroup (the staged list overflows into the shared one), level 2."""
    rng = np.random.default_rng(77)
    base = rng.standard_normal(384).astype(f32); base /= np.linalg.norm(base)
    rows = base[None, :] + f32(0.006) * rng.standard_normal((70000, 384)).astype(f32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    rows = np.ascontiguousarray(rows.astype(f32))
    idx = make_index(S)
    idx.build(rows)
    for qi, k in ((5, 10), (40001, 32), (69999, 120)):
        check_one(oracle, idx, rows, rows[qi], k, 0)
    idx.close()


def test_negative_scores_few_live_rows_and_ties(S, oracle):
    n = 20000
    rows = synth.corpus(n)
    idx = make_index(S)
    idx.build(rows)
    # every score negative: the batch pipeline's sampled threshold is unusable for such a query (exact fallback), a local k-th best is not
    qn = -rows[:4000].mean(axis=0).astype(f32)
    qn /= np.linalg.norm(qn)
    check_one(oracle, idx, rows, np.ascontiguousarray(qn), 10, 0)
    assert idx.scan_stats()["overflowed"] == 0
    # fewer live rows than k in the whole corpus
    deleted = np.ones(n, np.uint8)
    deleted[[5, 77, 10001, 19999]] = 0
    idx.mark_deleted_many(np.nonzero(deleted)[0].astype(np.uint32))
    q = synth.queries(2)
    ids, dist, counts = idx.search_batch(q[:1], 10)
    assert int(counts[0]) == 4
    check_one(oracle, idx, rows, q[0], 10, 0, deleted)
    idx.close()
    # every row identical: all candidates tie, the window holds the whole corpus -> exact scan, decided on the host after the call's
    # own synchronisation
    base = synth.queries(1)[0]
    same = np.tile(base, (n, 1)).astype(f32)
    idx2 = make_index(S)
    idx2.build(same)
    for qq in (base, -base, synth.queries(2)[1]):
        ids, dist, counts = idx2.search_batch(np.ascontiguousarray(qq[None, :]), 10)
        e_ids, e_dist = oracle.brute_force_search(same, qq, 10, select=True)
        assert ids[0].tolist() == e_ids.tolist() == list(range(10)) and dist[0].tobytes() == e_dist.tobytes()
        assert idx2.scan_stats()["overflowed"] == 1
        ids, dist, counts = idx2.search_batch(np.ascontiguousarray(qq[None, :]), 120)      # the same through the global-threshold mode
        e_ids, e_dist = oracle.brute_force_search(same, qq, 120, select=True)
        assert ids[0].tolist() == e_ids.tolist() and dist[0].tobytes() == e_dist.tobytes()
        assert idx2.scan_stats()["overflowed"] == 1
    idx2.close()


@pytest.mark.parametrize("order", [0, 1])
def test_crowded_corpus_and_interleaved_batches(S, oracle, order):
    """40k rows in a narrow cone: half the corpus is inside the fp16 window of the cone axis (level-2 filter), every slice hands over
    hundreds of rows through the shared list; batch calls in between use the same workspace"""
    rng = np.random.default_rng(5)
    base = synth.queries(1)[0]
    rows = base[None, :] + f32(0.0045) * rng.standard_normal((40000, 384)).astype(f32)     # (0.006 until round 6: the window follows the measured rounding residuals now)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    rows = np.ascontiguousarray(rows.astype(f32))
    idx = make_index(S, order=order)
    idx.build(rows)
    qs = np.ascontiguousarray(np.stack([base, rows[5], rows[39999], synth.queries(2)[1]]).astype(f32))
    for rep in range(2):
        for i in range(len(qs)):
            check_one(oracle, idx, rows, qs[i], 10, order)
            assert idx.scan_stats()["overflowed"] == 0
        ids, dist, counts = idx.search_batch(qs, 10)                 # batch pipeline on the same handle
        e_ids, e_dist = oracle.brute_force_batch(rows, qs, 10, order=order)
        assert np.array_equal(ids, e_ids) and dist.tobytes() == e_dist.tobytes()
    check_one(oracle, idx, rows, base, 10, order)
    assert idx.scan_stats()["level2"] == 1
    for i in range(len(qs)):                                             # global-threshold mode on the crowded corpus
        check_one(oracle, idx, rows, qs[i], 120, order)
        assert idx.scan_stats()["overflowed"] == 0
    check_one(oracle, idx, rows, qs[0], 10, order)
    idx.close()


def test_device_pointer_single_query(S, oracle):
    """search_device with one query: same pass, the exact fallback enqueued behind it (no host look at the statistics)"""
    import torch
    q = synth.queries(4)
    rows = synth.corpus(30000, queries=q)
    idx = make_index(S)
    idx.build(torch.from_numpy(rows).cuda())
    for i in range(4):
        ids, dist, counts = idx.search_batch(torch.from_numpy(q[i:i + 1]).cuda(), 10)
        torch.cuda.synchronize()
        e_ids, e_dist = oracle.brute_force_search(rows, q[i], 10, select=True)
        assert ids.cpu().numpy().view(np.uint32)[0].tolist() == e_ids.tolist() and dist.cpu().numpy()[0].tobytes() == e_dist.tobytes()
    ids, dist, counts = idx.search_batch(torch.from_numpy(q[2:3]).cuda(), 120)
    torch.cuda.synchronize()
    e_ids, e_dist = oracle.brute_force_search(rows, q[2], 120, select=True)
    assert ids.cpu().numpy().view(np.uint32)[0].tolist() == e_ids.tolist() and dist.cpu().numpy()[0].tobytes() == e_dist.tobytes()
    # a query the fp16 pre-scan cannot take (component beyond the fp16 range after scaling): device-side exact fallback
    big = (q[0] * f32(3000.0)).astype(f32)
    ids, dist, counts = idx.search_batch(torch.from_numpy(big[None, :]).cuda(), 10)
    torch.cuda.synchronize()
    e_ids, e_dist = oracle.brute_force_search(rows, big, 10, select=True)
    assert ids.cpu().numpy().view(np.uint32)[0].tolist() == e_ids.tolist() and dist.cpu().numpy()[0].tobytes() == e_dist.tobytes()
    # the same through host pointers (fallback decided after the synchronisation)
    check_one(oracle, idx, rows, big, 10, 0)
    assert idx.scan_stats()["overflowed"] == 1
    check_one(oracle, idx, rows, q[1], 10, 0)
    assert idx.scan_stats()["overflowed"] == 0
    idx.close()


def test_slice_cap_many_workgroups(S, oracle):
    """4.3M rows x 128-d: more rows than 256 slices of 16384 hold, so the launch has more workgroups than CUs and the final stage reads
    their slot table past its LDS staging"""
    n, dim = 4_300_000, 128
    rng = np.random.default_rng(9)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    q = rows[[17, n - 1]] + f32(0.1) * rng.standard_normal((2, dim)).astype(f32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = np.ascontiguousarray(q.astype(f32))
    idx = make_index(S, dim=dim)
    idx.build(rows)
    for i in range(2):
        check_one(oracle, idx, rows, q[i], 10, 0)
        st = idx.scan_stats()
        assert st["sampled_rows"] == 0 and st["overflowed"] == 0, st
        check_one(oracle, idx, rows, q[i], 120, 0)                       # no slice cap in the global-threshold mode: 256 slices of 16 797 rows
        st = idx.scan_stats()
        assert st["sampled_rows"] == 0 and st["overflowed"] == 0 and st["emitted"] < 1000, st
    idx.close()

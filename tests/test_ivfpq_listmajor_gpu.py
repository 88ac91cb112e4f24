"""The list-major batch scan of the IVF-PQ path (adc_bound_kernel: a bound from the head of every query's nearest lists; adc_list_kernel: pairs
grouped by list, codes in registers, two queries' tables interleaved in LDS, keys at or under the bound -> candidates; redo of overflowed queries;
lm_merge_kernel) against the oracle's SpannIndex::search (spann.rs:574-693, pq.rs:358-368) and against the query-major kernel: same ids
and distances, bit for bit, whichever kernel the batch shape selects."""
import numpy as np
import pytest

from tests import synth
from tests.test_ivfpq_gpu import check

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


@pytest.fixture(scope="module")
def state(oracle):
    rng = np.random.default_rng(5)
    rows = synth.corpus(2000, adversarial=False)
    P = oracle.spann_compute_partitions(2000)
    st = oracle.spann_build(rows, P, rng.permutation(2000).astype(np.uint32),
                            [rng.permutation(2000).astype(np.uint32) for _ in range(48)], kmeans_iterations=8)
    return rows, st


@pytest.fixture(autouse=True, params=["7", "60", None], ids=["bound=7-postings", "bound=60-postings", "bound=default"])
def bound_postings(request, monkeypatch):
    """how many postings of the nearest lists stand behind the bound: the default (3072) covers the small fixtures entirely, so every test
    also runs with a bound drawn from 60 postings (loose: many candidates) and from 7 (fewer than k = 10: no bound at all)"""
    if request.param is None:
        monkeypatch.delenv("SHODH_ADC_LM_BOUND_POSTINGS", raising=False)
    else:
        monkeypatch.setenv("SHODH_ADC_LM_BOUND_POSTINGS", request.param)


def _index(S, st, nprobe):
    idx = S.SpannIndex(384, num_probes=nprobe)
    idx.set_trained_state(st["centroids"], st["codebook"], st["list_off"], st["ids"], st["codes"])
    return idx


def test_list_major_matches_oracle_and_query_major(S, oracle, state, monkeypatch):
    rows, st = state
    q = np.concatenate([rows[:3], synth.queries(37)])
    for nprobe in (1, 7, 45):
        idx = _index(S, st, nprobe)
        for k in (1, 10, 16, 17, 40, 64, 120, 600):
            monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "0")
            a = idx.search_batch(q, k)
            monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "1")
            b = idx.search_batch(q, k)
            assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b)), (nprobe, k)
        monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "1")
        check(oracle, idx, st, q[:12], 10, nprobe)
        check(oracle, idx, st, q[:1], 10, nprobe)            # one query: blocks of one (its own pair partner)
        check(oracle, idx, st, q[:5], 33, nprobe)


def test_list_major_is_the_default_for_every_batch_size(S, oracle, state, monkeypatch):
    """no environment: the list-major scan, whatever the batch size -- 300 queries and one query give the same answers, equal to the query-major scan's"""
    rows, st = state
    monkeypatch.delenv("SHODH_ADC_LIST_MAJOR", raising=False)
    idx = _index(S, st, 20)
    q = synth.queries(300)
    ids, dist, counts = idx.search_batch(q, 10)
    for i in (0, 17, 299):
        one = idx.search_batch(q[i:i + 1], 10)
        assert one[0][0].tolist() == ids[i].tolist() and one[1][0].tobytes() == dist[i].tobytes()
    monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "0")
    qm = idx.search_batch(q, 10)
    assert qm[0].tobytes() == ids.tobytes() and qm[1].tobytes() == dist.tobytes()
    check(oracle, idx, st, q[:6], 10, 20)


def test_list_major_slow_selection_path(S, oracle, state, monkeypatch):
    """a candidate list of 4 keys overflows for (nearly) every query: those are recomputed from scratch by the query-major kernel"""
    rows, st = state
    monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "1")
    idx = _index(S, st, 9)
    q = synth.queries(21)
    ref = idx.search_batch(q, 10)
    monkeypatch.setenv("SHODH_ADC_LM_CAP", "4")
    got = idx.search_batch(q, 10)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(ref, got))
    check(oracle, idx, st, q[:4], 25, 9)


def test_list_major_long_lists_in_chunks(S, oracle, monkeypatch):
    """24 000 rows in 3 lists: 8 000 postings per list = three chunks of the 3 072-posting register tile, each its own work item (and more
    than one segment of the nearest-list scan). Also with the overflow path, and with an empty list among the probed ones."""
    rng = np.random.default_rng(3)
    n, P = 24000, 4
    rows = synth.corpus(n, adversarial=False)
    centroids = rows[rng.choice(n, P, replace=False)].copy()
    centroids[3] = -centroids[0]                             # nobody's nearest: an empty list that every query still probes
    codebook = np.stack([rows[rng.choice(n, 256, replace=False), m * 8:(m + 1) * 8] for m in range(48)]).astype(f32)
    idx = S.SpannIndex(384, num_probes=4)
    idx.set_trained_state(centroids, codebook, np.zeros(P + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, 48), np.uint8))
    assign, codes = idx.encode(rows)
    order = np.argsort(assign, kind="stable")
    off = np.zeros(P + 1, np.uint64); off[1:] = np.cumsum(np.bincount(assign, minlength=P))
    assert (np.diff(off.astype(np.int64)) > 3072).sum() >= 2
    st = dict(centroids=centroids, codebook=codebook, list_off=off, ids=order.astype(np.uint32), codes=codes[order])
    idx.set_trained_state(st["centroids"], st["codebook"], st["list_off"], st["ids"], st["codes"])
    q = np.concatenate([rows[:2], synth.queries(9)])
    for cap in ("0", "8"):
        monkeypatch.setenv("SHODH_ADC_LM_CAP", cap)
        monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "0")
        a = idx.search_batch(q, 10)
        monkeypatch.setenv("SHODH_ADC_LIST_MAJOR", "1")
        b = idx.search_batch(q, 10)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b)), cap
        check(oracle, idx, st, q[:3], 10, 4)
        check(oracle, idx, st, q[:2], 64, 4)

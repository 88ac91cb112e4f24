"""IVF probe selection on the centroid table (scan_mfma.hip: score pre-scan + probe_select_kernel, one wave per query) against the oracle's
SpannIndex::search (spann.rs:595-607: the num_probes nearest partitions by (compute_distance, index)) and against the general pipeline it replaces.
Every partition of these fixtures holds ONE posting whose id is the partition's index, and k = num_probes: the ids a search returns ARE the probed set."""
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


def one_posting_state(centroids, seed=3):
    rng = np.random.default_rng(seed)
    P = len(centroids)
    return dict(centroids=np.ascontiguousarray(centroids, f32), codebook=rng.standard_normal((48, 256, 8)).astype(f32) * f32(0.05),
                list_off=np.arange(P + 1, dtype=np.uint64), ids=np.arange(P, dtype=np.uint32), codes=rng.integers(0, 256, (P, 48), dtype=np.uint8))


def unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(f32)


def table(P, seed):
    """unit rows in a few clusters, with near-duplicates (1e-4 apart: inside the 2-eps band of each other) and exact duplicates (ties: the smaller index wins)"""
    rng = np.random.default_rng(seed)
    centres = unit(rng.standard_normal((8, 384)))
    c = unit(centres[rng.integers(0, 8, P)] + 0.6 * rng.standard_normal((P, 384)).astype(f32))
    for i in range(0, P - 8, 37):
        c[i + 1] = unit((c[i] + f32(1e-4) * rng.standard_normal(384).astype(f32))[None])[0]
        c[i + 5] = c[i]
    return c


def index_of(S, st, nprobe):
    idx = S.SpannIndex(384, num_probes=nprobe)
    idx.set_trained_state(st["centroids"], st["codebook"], st["list_off"], st["ids"], st["codes"])
    return idx


@pytest.mark.parametrize("P", [576, 1000, 4096, 5000])
def test_probe_sets_match_the_oracle(S, oracle, P, monkeypatch):
    monkeypatch.delenv("SHODH_PROBE_SELECT", raising=False)
    c = table(P, P)
    st = one_posting_state(c)
    rng = np.random.default_rng(P + 1)
    q = np.concatenate([c[:3], c[37:39], unit(rng.standard_normal((9, 384))), (c[7:8] * f32(1000.0)), np.zeros((1, 384), f32)])      # (x 1000: outside the fp16 range -> every row scored exactly)
    for nprobe in (1, 8, 32, 64):
        idx = index_of(S, st, nprobe)
        check(oracle, idx, st, q, nprobe, nprobe)


def test_dense_and_tied_tables(S, oracle, monkeypatch):
    """every row within the band (a table of near-equal rows: rounds of 64 exact sums, the best kept across rounds) and every row EQUAL (the k-th best is a tie of
    hundreds: bisection; the smallest indices win)"""
    monkeypatch.delenv("SHODH_PROBE_SELECT", raising=False)
    rng = np.random.default_rng(11)
    base = unit(rng.standard_normal((1, 384)))
    near = unit(base + f32(2e-5) * rng.standard_normal((640, 384)).astype(f32))
    q = np.concatenate([base, unit(rng.standard_normal((3, 384)))])
    for tab in (near, np.repeat(base, 576, axis=0)):
        st = one_posting_state(tab)
        for nprobe in (1, 8, 33):
            check(oracle, index_of(S, st, nprobe), st, q, nprobe, nprobe)


def test_equal_to_the_general_pipeline_and_nearest_centroid(S, oracle, monkeypatch):
    """whole searches over a real posting layout: byte-equal with the general pipeline (SHODH_PROBE_SELECT=0); encode()'s assignment (k = 1) equals the oracle's"""
    rng = np.random.default_rng(2)
    P, n = 4096, 60000
    c = table(P, 77)
    sizes = rng.multinomial(n, np.ones(P) / P)
    off = np.zeros(P + 1, np.uint64); off[1:] = np.cumsum(sizes)
    st = dict(centroids=c, codebook=(rng.standard_normal((48, 256, 8)) * 0.05).astype(f32), list_off=off, ids=rng.permutation(n).astype(np.uint32),
              codes=rng.integers(0, 256, (n, 48), dtype=np.uint8))
    q = np.concatenate([c[:40], unit(rng.standard_normal((1000, 384)))])
    for nprobe, k in ((32, 10), (20, 120), (64, 10)):
        idx = index_of(S, st, nprobe)
        monkeypatch.setenv("SHODH_PROBE_SELECT", "0")
        a = idx.search_batch(q, k)
        monkeypatch.setenv("SHODH_PROBE_SELECT", "1")
        b = idx.search_batch(q, k)
        assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b)), (nprobe, k)
        check(oracle, idx, st, q[:4], k, nprobe)
    idx = index_of(S, st, 8)
    new = np.concatenate([c[100:110], unit(rng.standard_normal((200, 384)))])
    assign, _ = idx.encode(new)
    for i in range(0, len(new), 7):
        assert assign[i] == oracle.spann_find_nearest_centroid(new[i], c)


def test_band_full_of_rows_takes_the_f32_level(S, oracle, monkeypatch):
    """a trained table's shape: hundreds of rows inside the fp16 band around the k-th best (configs[3]: 50 - 90 per query), a handful inside the f32 band --
    the rows in between get an f32 score first and only the ones it cannot settle are summed in the reference's order"""
    monkeypatch.delenv("SHODH_PROBE_SELECT", raising=False)
    rng = np.random.default_rng(21)
    base = unit(rng.standard_normal((2, 384)))
    tab = np.concatenate([unit(base[0] + f32(0.01) * rng.standard_normal((300, 384)).astype(f32)), unit(rng.standard_normal((600, 384))),
                          unit(base[1] + f32(0.004) * rng.standard_normal((124, 384)).astype(f32))])
    tab = tab[rng.permutation(len(tab))]
    st = one_posting_state(tab)
    q = np.concatenate([base, unit(base + f32(0.02) * rng.standard_normal((2, 384)).astype(f32)), unit(rng.standard_normal((4, 384)))])
    for nprobe in (1, 10, 32, 64):
        check(oracle, index_of(S, st, nprobe), st, q, nprobe, nprobe)


@pytest.mark.parametrize("dim", [128, 512])
def test_other_dimensions(S, oracle, dim, monkeypatch):
    """the score pre-scan's other instantiations (8 and 32 k-steps)"""
    monkeypatch.delenv("SHODH_PROBE_SELECT", raising=False)
    rng = np.random.default_rng(dim)
    P = 1536
    c = unit(rng.standard_normal((P, dim)))
    c[5] = c[3]
    c[40] = unit((c[41] + f32(1e-4) * rng.standard_normal(dim).astype(f32))[None])[0]
    st = dict(centroids=c, codebook=(rng.standard_normal((dim // 8, 256, 8)) * 0.05).astype(f32), list_off=np.arange(P + 1, dtype=np.uint64),
              ids=np.arange(P, dtype=np.uint32), codes=rng.integers(0, 256, (P, dim // 8), dtype=np.uint8))
    q = np.concatenate([c[3:6], c[40:42], unit(rng.standard_normal((6, dim)))])
    for nprobe in (1, 20, 64):
        idx = S.SpannIndex(dim, num_probes=nprobe)
        idx.set_trained_state(st["centroids"], st["codebook"], st["list_off"], st["ids"], st["codes"])
        check(oracle, idx, st, q, nprobe, nprobe)

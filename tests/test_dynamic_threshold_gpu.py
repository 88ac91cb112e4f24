"""The emit threshold that tightens DURING the scan (csrc/scan_mfma.hip, MfmaArgs::dcnt): results must stay the exact scan's, ids and distance
bytes, on the corpora where the mechanism has the most room to go wrong -- big enough for many publication rounds, scores piled up at the top (duplicates of
the query, dense cones), heavy tombstoning, several passes of 256 queries, fewer queries than waves, and every k from 1 to 1000. The comparator is the
exact-order scan of the same index (itself bit-equal to the oracle: tests/test_flat_gpu.py), so a million rows cost seconds, not minutes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from . import synth
from .conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def _both(S, rows, dim=384, dead=None):
    a = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=2))      # MFMA pre-scan (dynamic threshold) + exact re-score
    b = S.VamanaIndex(S.VamanaConfig(dimension=dim, scan_mode=1))      # the exact-order scan
    a.build(rows); b.build(rows)
    if dead is not None:
        for i in dead:
            a.mark_deleted(int(i)); b.mark_deleted(int(i))
    return a, b


def _same(a, b, q, k):
    i1, d1, c1 = a.search_batch(q, k)
    i2, d2, c2 = b.search_batch(q, k)
    assert np.array_equal(c1, c2)
    assert np.array_equal(i1, i2), np.argwhere(i1 != i2)[:5]
    assert d1.tobytes() == d2.tobytes()
    return a.scan_stats()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_dynamic_threshold_on_a_large_corpus_every_k_and_batch_shape(S):
    q = synth.queries(300)
    rows = synth.corpus(600_000, queries=q)
    a, b = _both(S, rows)
    for nq, k in ((256, 10), (256, 120), (300, 10), (64, 120), (5, 10), (33, 1), (200, 1000), (2, 300)):
        st = _same(a, b, q[:nq], k)
        assert st["overflowed"] == 0, (nq, k, st)
    a.close(); b.close()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
@pytest.mark.parametrize("dim", [128, 256])
def test_dynamic_threshold_at_other_dimensions(S, dim):
    q = synth.queries(256, dim=dim)
    rows = synth.corpus(300_000, dim=dim, queries=q)
    a, b = _both(S, rows, dim=dim)
    for k in (10, 120):
        _same(a, b, q, k)
    a.close(); b.close()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_dynamic_threshold_with_scores_piled_up_at_the_top(S):
    """hundreds of exact copies of every query (score 1.0, far above every level), a dense cone around some of them, and half the corpus tombstoned --
    including many of the copies: a count must never include a row that is not there"""
    rng = np.random.default_rng(9)
    q = synth.queries(64)
    rows = synth.corpus(400_000, queries=q)
    for j in range(64):                                          # 300 copies of each query, spread over the corpus
        rows[rng.integers(0, len(rows), 300)] = q[j]
    cone = q[3] + f32(0.02) * rng.standard_normal((5000, 384)).astype(f32)
    cone /= np.linalg.norm(cone, axis=1, keepdims=True)
    rows[rng.integers(0, len(rows), 5000)] = cone
    dead = rng.choice(len(rows), len(rows) // 2, replace=False)
    a, b = _both(S, rows, dead=dead)
    for k in (10, 120, 500):
        _same(a, b, q, k)
    a.close(); b.close()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_the_bound_really_moves_and_the_switch_turns_it_off(S):
    """SHODH_DYN_THR=0 (read once per process: a child) leaves the sampled bound alone; with the mechanism on, fewer rows survive the pre-scan and the answers
    are the same bytes"""
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
import shodh_memory_amd as S
from tests import synth
q = synth.queries(256)
rows = synth.corpus(500_000, queries=q)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, scan_mode=2)); idx.build(rows)
ids, dist, _ = idx.search_batch(q, 120)
print('emitted', idx.scan_stats()['emitted'], 'digest', int(ids.astype(np.uint64).sum()), dist.tobytes().hex()[:64])
""" % ROOT
    outs = {}
    for dyn in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SHODH_DYN_THR=dyn), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("emitted")]
        assert p.returncode == 0 and line, p.stdout.decode()[-2000:]
        outs[dyn] = line[0].split()
    assert outs["0"][3:] == outs["1"][3:]                                  # the same ids and distance bytes
    assert int(outs["1"][1]) < 0.75 * int(outs["0"][1]), (outs["0"][1], outs["1"][1])      # ... from noticeably fewer survivors

"""The allocation guard (csrc/guard.h) itself: under SHODH_GUARD=1 a kernel that reads 16 bytes past the end of a buffer dies with a GPU page
fault, deterministically, and the same call with a buffer of the right size answers. The faulting half is run only on request
(SHODH_GUARD_SELFTEST=1): it kills a child process with a GPU memory access fault ON PURPOSE."""
import os
import subprocess
import sys

import pytest

from .conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
import shodh_memory_amd as S
from shodh_memory_amd import _lib as L
lib = L.lib()
assert lib.shodh_guard_mode() == int(os.environ['SHODH_GUARD'])
short = int(sys.argv[1])
rng = np.random.default_rng(5)
rows = rng.standard_normal((20000, 384)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, scan_mode=2, device=0))
idx.build(rows)
nq, k = 4, 10
qbytes = nq * 384 * 4 - short                      # `short` bytes missing at the end of the caller's query buffer
d_q = lib.shodh_guard_torch_alloc(qbytes, 0, None)
d_ids = lib.shodh_guard_torch_alloc(nq * k * 4, 0, None); d_dist = lib.shodh_guard_torch_alloc(nq * k * 4, 0, None); d_cnt = lib.shodh_guard_torch_alloc(nq * 4, 0, None)
hip = C.CDLL('libamdhip64.so')
q = np.ascontiguousarray(rows[:nq])
hip.hipMemcpy(C.c_void_p(d_q), q.ctypes.data_as(C.c_void_p), C.c_size_t(qbytes), 1)
rc = lib.shodh_index_search_device(idx.handle, C.c_void_p(d_q), nq, k, C.c_void_p(d_ids), C.c_void_p(d_dist), C.c_void_p(d_cnt), None)
hip.hipDeviceSynchronize()
ids = np.zeros((nq, k), np.uint32)
hip.hipMemcpy(ids.ctypes.data_as(C.c_void_p), C.c_void_p(d_ids), C.c_size_t(ids.nbytes), 2)
assert rc == 0 and ids[:, 0].tolist() == [0, 1, 2, 3], (rc, ids[:, 0])
a, f, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
lib.shodh_guard_stats(C.byref(a), C.byref(f), C.byref(b))
print('answered; guard allocations', a.value, 'frees', f.value, 'live device bytes', b.value)
"""


def _run(short, mode="1"):
    env = dict(os.environ, SHODH_GUARD=mode, SHODH_COALESCE="0")
    return subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(short)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_guarded_allocations_serve_a_search(mode):
    p = _run(0, mode)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert "answered; guard allocations" in out and "[shodh guard] mode %s" % mode in out
    assert "NOT fenced" not in out, out[-2000:]            # pinned host memory is fenced as well on this platform


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
@pytest.mark.skipif(os.environ.get("SHODH_GUARD_SELFTEST", "0") in ("", "0"), reason="faults the GPU on purpose: SHODH_GUARD_SELFTEST=1")
def test_an_overread_of_sixteen_bytes_is_a_page_fault():
    p = _run(16)
    out = p.stdout.decode()
    assert p.returncode != 0 and "Memory access fault" in out, out[-3000:]


ENC_CHILD = r"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from shodh_memory_amd import _lib as L
from shodh_memory_amd.embedder import MiniLMEmbedder
assert L.lib().shodh_guard_mode() == 1

def batch(lengths, seed):
    rng = np.random.default_rng(seed)
    ids = np.zeros((len(lengths), 256), np.int32); mask = np.zeros((len(lengths), 256), np.uint8)
    for i, n in enumerate(lengths):
        ids[i, :n] = rng.integers(1000, 30521, n); ids[i, 0] = 101; ids[i, n - 1] = 102; mask[i, :n] = 1
    return ids, mask

for dtype, padded in ((L.DTYPE_BF16, 0), (L.DTYPE_FP32, 0), (L.DTYPE_INT8, 1), (L.DTYPE_INT8, 0)):
    e = MiniLMEmbedder(synthetic_seed=3, dtype=dtype, device=0, compute_padded=padded)
    e.set_coalesce(False)
    ref = {}
    # the round-5 fault: a first forward of a few tokens sizes the scratch (capacity = n + n/4 + 256 token rows), later forwards FILL it -- their last
    # 64- / 128- / 256-row tile used to reach past the end of the allocation. Token totals around every capacity the growth formula produces.
    for n_tok in (16, 270, 276, 277, 300, 601, 631, 632, 1000, 1506, 1507, 130, 64, 65, 1):
        lengths = [128] * (n_tok // 128) + ([n_tok %% 128] if n_tok %% 128 else [])
        lengths = [max(l, 2) for l in lengths] if n_tok > 1 else [2]
        ids, mask = batch(lengths, n_tok)
        out = e.encode_ids(ids, mask)
        assert np.isfinite(out).all() and np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-3), (dtype, n_tok)
        one = e.encode_ids(ids[:1], mask[:1])                  # (and a one-text forward right after a large one)
        assert np.isfinite(one).all()
    e.close()
print('encoder forwards at every scratch capacity: no fault')
"""


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_encoder_scratch_filled_to_its_capacity_under_the_guard():
    """Regression for the round-5 'Memory access fault by GPU node' (one bench run in thirty): the encoder's per-token scratch had no tile slack beyond its
    capacity, so a forward that filled it read up to 255 token rows past the allocation. Under SHODH_GUARD=1 that is a fault on every run."""
    env = dict(os.environ, SHODH_GUARD="1")
    p = subprocess.run([sys.executable, "-c", ENC_CHILD % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "no fault" in out, out[-3000:]

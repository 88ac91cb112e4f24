"""src/similarity.rs over the C ABI: `cosine_similarity` (:10-24) and `top_k_similar` (:27-48), same names and semantics."""
import ctypes as C

import numpy as np

from . import _lib as L


def cosine_similarity(a, b, order=L.ORDER_SCALAR4, device=0):
    """clamp(dot / (|a| |b|), -1, 1) in dot_product_inline's order; 0.0 on a length mismatch or a zero norm"""
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    b = np.ascontiguousarray(b, np.float32).reshape(-1)
    if a.size != b.size:
        return 0.0
    if a.size == 0:
        return 0.0                       # both norms are 0
    out = np.zeros(1, np.float32)
    L.check(L.lib().shodh_cosine_similarity_batch(device, a.ctypes.data, b.ctypes.data, 1, a.size, order, out.ctypes.data))
    return float(out[0])


def top_k_similar(query, candidates, k, order=L.ORDER_SCALAR4, device=0):
    """candidates: sequence of (vector, item). -> [(score, item)], stable-sorted by score descending, at most k."""
    if len(candidates) == 0:
        return []
    q = np.ascontiguousarray(query, np.float32).reshape(-1)
    dims = {len(v) for v, _ in candidates}
    if len(dims) != 1:
        # ragged candidate lists: the reference scores each pair on its own (a mismatched one gets 0.0); group by length
        scores = [cosine_similarity(q, v, order, device) for v, _ in candidates]
        idx = sorted(range(len(scores)), key=lambda i: _desc_key(scores[i]))          # sorted() is stable
        return [(scores[i], candidates[i][1]) for i in idx[:k]]
    dim = dims.pop()
    c = np.ascontiguousarray([v for v, _ in candidates], np.float32).reshape(len(candidates), dim)
    kk = min(int(k), len(candidates))
    sc = np.zeros(max(kk, 1), np.float32)
    ix = np.zeros(max(kk, 1), np.uint32)
    cnt = C.c_uint64()
    L.check(L.lib().shodh_top_k_similar(device, q.ctypes.data, q.size, c.ctypes.data, len(candidates), dim, kk, order,
                                        sc.ctypes.data, ix.ctypes.data, C.byref(cnt)))
    return [(float(sc[i]), candidates[int(ix[i])][1]) for i in range(cnt.value)]


def _desc_key(s):
    # OrderedFloat descending: NaN first, then +inf .. -inf; -0.0 == +0.0
    return (0, 0.0) if s != s else (1, -(s + 0.0))

"""ctypes front-end of the CPU oracle (oracle/shodh_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py. The product package (shodh_memory_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libshodh_oracle.so")

ORDER_SCALAR4, ORDER_AVX2 = 0, 1
METRIC_NDP, METRIC_EUCLIDEAN, METRIC_COSINE = 0, 1, 2


def build(force=False):
    """Compile the oracle with gcc (-ffp-contract=off). Idempotent."""
    src = os.path.join(_HERE, "shodh_oracle.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(src)
            and os.path.getmtime(_SO) >= os.path.getmtime(os.path.join(_HERE, "vamana_oracle.c"))
            and os.path.getmtime(_SO) >= os.path.getmtime(os.path.join(_HERE, "shodh_oracle.h"))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libshodh_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None


class Weights(C.Structure):
    _fields_ = [("semantic", C.c_float), ("entity", C.c_float), ("tag", C.c_float),
                ("importance", C.c_float), ("momentum", C.c_float),
                ("access_count", C.c_float), ("graph_strength", C.c_float),
                ("update_count", C.c_uint32)]

    def as_tuple(self):
        return (self.semantic, self.entity, self.tag, self.importance, self.momentum,
                self.access_count, self.graph_strength)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    fp = C.POINTER(C.c_float)
    u8p = C.POINTER(C.c_uint8)
    u32p = C.POINTER(C.c_uint32)
    u64p = C.POINTER(C.c_uint64)
    i64p = C.POINTER(C.c_int64)
    sz = C.c_size_t
    sig = {
        "so_dot_scalar4": (C.c_float, [fp, fp, sz]),
        "so_dot_avx2": (C.c_float, [fp, fp, sz]),
        "so_dot_avx2_native": (C.c_float, [fp, fp, sz]),
        "so_dot": (C.c_float, [fp, fp, sz, C.c_int]),
        "so_l2sq": (C.c_float, [fp, fp, sz, C.c_int]),
        "so_normsq": (C.c_float, [fp, sz, C.c_int]),
        "so_l2_norm": (C.c_float, [fp, sz, C.c_int]),
        "so_cosine_similarity_inline": (C.c_float, [fp, fp, sz, C.c_int]),
        "so_cosine_distance_inline": (C.c_float, [fp, fp, sz, C.c_int]),
        "so_normalized_distance": (C.c_float, [fp, fp, sz, C.c_int]),
        "so_is_normalized": (C.c_int, [fp, sz, C.c_float, C.c_int]),
        "so_normalize_inplace": (None, [fp, sz, C.c_int]),
        "so_metric_distance": (C.c_float, [fp, fp, sz, C.c_int, C.c_int]),
        "so_cosine_similarity": (C.c_float, [fp, sz, fp, sz, C.c_int]),
        "so_top_k_similar": (sz, [fp, fp, sz, sz, sz, C.c_int, fp, u32p]),
        "so_total_cmp": (C.c_int, [C.c_float, C.c_float]),
        "so_total_order_key": (C.c_uint32, [C.c_float]),
        "so_brute_force_search": (sz, [fp, sz, sz, u8p, fp, sz, C.c_int, C.c_int, u32p, fp]),
        "so_brute_force_search_select": (sz, [fp, sz, sz, u8p, fp, sz, C.c_int, C.c_int, u32p, fp]),
        "so_search_ids_postprocess": (sz, [u32p, fp, sz, u8p, sz, sz, u8p, fp]),
        "so_squared_l2": (C.c_float, [fp, fp, sz]),
        "so_pq_encode": (None, [fp, sz, sz, sz, fp, u8p]),
        "so_pq_decode": (None, [fp, sz, sz, sz, u8p, fp]),
        "so_pq_build_distance_table": (None, [fp, sz, sz, sz, fp, fp]),
        "so_pq_distance_with_table": (C.c_float, [fp, sz, sz, u8p, sz]),
        "so_pq_asymmetric_distance": (C.c_float, [fp, sz, sz, sz, fp, u8p]),
        "so_pq_kmeans": (None, [fp, sz, sz, sz, sz, u32p, fp]),
        "so_spann_compute_distance": (C.c_float, [fp, fp, sz, C.c_int]),
        "so_spann_find_nearest_centroid": (sz, [fp, fp, sz, sz, C.c_int]),
        "so_spann_compute_partitions": (sz, [sz]),
        "so_spann_kmeans": (sz, [fp, sz, sz, sz, sz, C.c_int, u32p, fp]),
        "so_spann_search": (sz, [fp, sz, sz, C.c_int, u64p, u32p, u8p, fp, sz, sz, sz, sz, fp, sz, u32p, fp]),
        "so_mean_pool_finalize": (None, [fp, i64p, sz, sz, fp]),
        "so_siphash13_str": (C.c_uint64, [u8p, sz]),
        "so_hash_embed": (None, [C.c_char_p, sz, sz, fp]),
        "so_weights_default": (None, [C.POINTER(Weights)]),
        "so_weights_normalize": (None, [C.POINTER(Weights)]),
        "so_weights_apply_feedback": (None, [C.POINTER(Weights), C.c_int, C.c_int, C.c_int, C.c_int]),
        "so_calibrate_score": (C.c_float, [C.c_float]),
        "so_fuse_scores_full": (C.c_float, [C.POINTER(Weights)] + [C.c_float] * 5 + [C.c_uint32, C.c_float]),
        "so_fuse_scores": (C.c_float, [C.POINTER(Weights)] + [C.c_float] * 4),
        "so_fuse_scores_with_momentum": (C.c_float, [C.POINTER(Weights)] + [C.c_float] * 5),
        "so_calculate_tag_score": (C.c_float, [C.c_char_p, C.POINTER(C.c_char_p), sz]),
        "so_apply_recency_boost": (C.c_float, [C.c_float, C.c_int64, C.c_uint64, C.c_float]),
        "so_rank_surfaced": (sz, [C.POINTER(Weights), C.c_float, C.c_uint64, C.c_float, C.c_float, sz, sz, fp, fp, fp, fp, fp, u32p, fp,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64), u8p, u32p, fp, u8p]),
        "so_rrf_fuse": (sz, [C.c_float, fp, sz, u8p, C.POINTER(sz), u8p, fp, sz]),
        "so_density_weights": (None, [C.c_float, fp]),
        "so_leg_fusion_weights": (None, [C.c_int, C.c_float, C.c_float, C.c_float, fp, fp]),
        "so_fuse_legs": (sz, [C.POINTER(C.c_int), fp, u8p, fp, fp, sz, u8p, fp, sz, sz, u8p, fp, sz, fp]),
        "so_fnv1a64": (C.c_uint64, [u8p, sz]),
        "so_finalize_pooled": (sz, [fp, sz, C.c_int, sz, fp]),
        "so_vamana_greedy_search": (sz, [fp, sz, sz, u32p, u32p, sz, fp, sz, C.c_uint32, C.c_int, u32p, fp]),
        "so_vamana_search": (sz, [fp, sz, sz, u32p, u32p, sz, C.c_uint32, u8p, fp, sz, C.c_int, u32p, fp]),
        "so_vamana_add_vector": (None, [fp, sz, sz, u32p, u32p, sz, sz, C.c_uint32, C.c_int]),
        "so_vamana_robust_prune": (sz, [fp, sz, C.c_uint32, u32p, fp, sz, sz, C.c_float, C.c_int, u32p]),
        "so_vamana_find_medoid": (C.c_uint32, [fp, sz, sz, C.c_int]),
        "so_vamana_incremental_repair": (sz, [fp, sz, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), sz, sz, sz, C.c_float, C.c_uint32, C.c_uint32, C.c_int]),
        "so_vamana_build": (C.c_uint32, [fp, sz, sz, u32p, u32p, sz, sz, sz, C.c_float, C.c_int]),
        "so_bench_brute_force": (C.c_double, [fp, sz, sz, fp, sz, sz, C.c_int, C.c_int, C.c_int, u32p, fp]),
        "so_bench_brute_force_del": (C.c_double, [fp, sz, sz, u8p, fp, sz, sz, C.c_int, C.c_int, C.c_int, u32p, fp]),
        "so_interleaved_copy": (C.c_void_p, [fp, sz, C.c_int]),
        "so_free": (None, [C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


# ---- distance_inline / similarity ----------------------------------------------------
def dot(a, b, order=ORDER_SCALAR4):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_dot(pa, pb, a.size, order))


def dot_avx2_emulated(a, b):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_dot_avx2(pa, pb, a.size))


def l2sq(a, b, order=ORDER_SCALAR4):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_l2sq(pa, pb, a.size, order))


def l2_norm(a, order=ORDER_SCALAR4):
    a, pa = _f(a)
    return np.float32(lib().so_l2_norm(pa, a.size, order))


def cosine_similarity_inline(a, b, order=ORDER_SCALAR4):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_cosine_similarity_inline(pa, pb, a.size, order))


def normalized_distance(a, b, order=ORDER_SCALAR4):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_normalized_distance(pa, pb, a.size, order))


def is_normalized(a, eps, order=ORDER_SCALAR4):
    a, pa = _f(a)
    return bool(lib().so_is_normalized(pa, a.size, eps, order))


def normalize_inplace(a, order=ORDER_SCALAR4):
    a = np.array(a, dtype=np.float32)
    lib().so_normalize_inplace(_p(a, C.c_float), a.size, order)
    return a


def cosine_similarity(a, b, order=ORDER_SCALAR4):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_cosine_similarity(pa, a.size, pb, b.size, order))


def top_k_similar(query, cands, k, order=ORDER_SCALAR4):
    q, pq = _f(query)
    c = np.ascontiguousarray(cands, dtype=np.float32).reshape(-1, q.size) if len(cands) else np.zeros((0, q.size), np.float32)
    n = c.shape[0]
    sc = np.zeros(max(min(k, n), 1), np.float32)
    ix = np.zeros(max(min(k, n), 1), np.uint32)
    m = lib().so_top_k_similar(pq, _p(c, C.c_float), n, q.size, k, order, _p(sc, C.c_float), _p(ix, C.c_uint32))
    return sc[:m], ix[:m]


def total_order_key(x):
    return int(lib().so_total_order_key(np.float32(x)))


# ---- brute force -----------------------------------------------------------------------
def brute_force_search(rows, q, k, deleted=None, metric=METRIC_NDP, order=ORDER_SCALAR4, select=False):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n, dim = (rows.shape if rows.ndim == 2 else (0, np.asarray(q).size))
    q, pq = _f(q)
    ids = np.zeros(max(k, 1), np.uint32)
    dist = np.zeros(max(k, 1), np.float32)
    dp = None
    if deleted is not None:
        deleted = np.ascontiguousarray(deleted, dtype=np.uint8)
        dp = _p(deleted, C.c_uint8)
    fn = lib().so_brute_force_search_select if select else lib().so_brute_force_search
    m = fn(_p(rows, C.c_float), n, dim, dp, pq, k, metric, order, _p(ids, C.c_uint32), _p(dist, C.c_float))
    return ids[:m].copy(), dist[:m].copy()


def brute_force_batch(rows, queries, k, deleted=None, order=ORDER_SCALAR4, threads=None):
    """top-k for many queries (select variant, threaded) -> ids [nq,k], dist [nq,k]; rows must be >= k live."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    nq = queries.shape[0]
    if deleted is not None:
        out_i = np.zeros((nq, k), np.uint32); out_d = np.zeros((nq, k), np.float32)
        for i in range(nq):
            a, b = brute_force_search(rows, queries[i], k, deleted, order=order, select=True)
            out_i[i, :len(a)] = a; out_d[i, :len(b)] = b
        return out_i, out_d
    ids = np.zeros((nq, k), np.uint32)
    dist = np.zeros((nq, k), np.float32)
    threads = threads or os.cpu_count() or 1
    lib().so_bench_brute_force(_p(rows, C.c_float), rows.shape[0], rows.shape[1], _p(queries, C.c_float),
                               nq, k, order, 0, threads, _p(ids, C.c_uint32), _p(dist, C.c_float))
    return ids, dist


def bench_brute_force(rows, queries, k, order, full_sort, threads, deleted=None):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    nq = queries.shape[0]
    ids = np.zeros((nq, k), np.uint32)
    dist = np.zeros((nq, k), np.float32)
    dp = None
    if deleted is not None:
        deleted = np.ascontiguousarray(deleted, dtype=np.uint8)
        dp = _p(deleted, C.c_uint8)
    secs = lib().so_bench_brute_force_del(_p(rows, C.c_float), rows.shape[0], rows.shape[1], dp, _p(queries, C.c_float),
                                          nq, k, order, int(full_sort), threads, _p(ids, C.c_uint32), _p(dist, C.c_float))
    return secs, ids, dist


class InterleavedCopy:
    """A copy of `rows` whose pages were first touched by `threads` threads in 2 MiB stripes (so_interleaved_copy): on a
    multi-socket host it is spread over the NUMA nodes instead of sitting where the producing thread ran. `.array` is a
    numpy view; the memory is released when the object goes away."""

    def __init__(self, rows, threads):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self._ptr = lib().so_interleaved_copy(_p(rows, C.c_float), rows.size, int(threads))
        if not self._ptr:
            raise MemoryError("so_interleaved_copy")
        self.array = np.ctypeslib.as_array((C.c_float * rows.size).from_address(self._ptr)).reshape(rows.shape)

    def close(self):
        if self._ptr:
            self.array = None
            lib().so_free(self._ptr)
            self._ptr = None

    def __del__(self):
        self.close()


def search_ids_postprocess(vec_ids, dists, vector_to_memory, limit):
    vec_ids = np.ascontiguousarray(vec_ids, dtype=np.uint32)
    dists = np.ascontiguousarray(dists, dtype=np.float32)
    v2m = np.ascontiguousarray(vector_to_memory, dtype=np.uint8).reshape(-1, 16)
    ou = np.zeros((max(limit, 1), 16), np.uint8)
    os_ = np.zeros(max(limit, 1), np.float32)
    m = lib().so_search_ids_postprocess(_p(vec_ids, C.c_uint32), _p(dists, C.c_float), vec_ids.size,
                                        _p(v2m, C.c_uint8), v2m.shape[0], limit, _p(ou, C.c_uint8), _p(os_, C.c_float))
    return ou[:m].copy(), os_[:m].copy()


# ---- PQ / SPANN ------------------------------------------------------------------------
def pq_encode(codebook, v):
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, nc, sub = cb.shape
    v, pv = _f(v)
    codes = np.zeros(M, np.uint8)
    lib().so_pq_encode(_p(cb, C.c_float), M, nc, sub, pv, _p(codes, C.c_uint8))
    return codes


def pq_decode(codebook, codes):
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, nc, sub = cb.shape
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(M * sub, np.float32)
    lib().so_pq_decode(_p(cb, C.c_float), M, nc, sub, _p(codes, C.c_uint8), _p(out, C.c_float))
    return out


def pq_build_distance_table(codebook, q):
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, nc, sub = cb.shape
    q, pq_ = _f(q)
    t = np.zeros((M, nc), np.float32)
    lib().so_pq_build_distance_table(_p(cb, C.c_float), M, nc, sub, pq_, _p(t, C.c_float))
    return t


def pq_distance_with_table(table, codes):
    t = np.ascontiguousarray(table, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    return np.float32(lib().so_pq_distance_with_table(_p(t, C.c_float), t.shape[0], t.shape[1], _p(codes, C.c_uint8), codes.size))


def pq_asymmetric_distance(codebook, q, codes):
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, nc, sub = cb.shape
    q, pq_ = _f(q)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    return np.float32(lib().so_pq_asymmetric_distance(_p(cb, C.c_float), M, nc, sub, pq_, _p(codes, C.c_uint8)))


def pq_kmeans(vectors, k, iterations, init_perm):
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    perm = np.ascontiguousarray(init_perm, dtype=np.uint32)
    out = np.zeros((k, v.shape[1]), np.float32)
    lib().so_pq_kmeans(_p(v, C.c_float), v.shape[0], v.shape[1], k, iterations, _p(perm, C.c_uint32), _p(out, C.c_float))
    return out


def pq_train(vectors, init_perms, ncent=256, sub=8, iterations=20):
    """ProductQuantizer::fit (pq.rs:114-150): independent k-means per subspace -> [M][ncent][sub]."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = v.shape
    M = dim // sub
    k = min(ncent, n)
    cb = np.zeros((M, k, sub), np.float32)
    for m in range(M):
        cb[m] = pq_kmeans(v[:, m * sub:(m + 1) * sub], k, iterations, init_perms[m])
    return cb


def spann_compute_distance(a, b, metric=METRIC_NDP):
    a, pa = _f(a); b, pb = _f(b)
    return np.float32(lib().so_spann_compute_distance(pa, pb, a.size, metric))


def spann_find_nearest_centroid(v, centroids, metric=METRIC_NDP):
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    v, pv = _f(v)
    return int(lib().so_spann_find_nearest_centroid(pv, _p(c, C.c_float), c.shape[0], c.shape[1], metric))


def spann_compute_partitions(n):
    return int(lib().so_spann_compute_partitions(n))


def spann_kmeans(vectors, k, iterations, init_perm, metric=METRIC_NDP):
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    perm = np.ascontiguousarray(init_perm, dtype=np.uint32)
    out = np.zeros((k, v.shape[1]), np.float32)
    its = lib().so_spann_kmeans(_p(v, C.c_float), v.shape[0], v.shape[1], k, iterations, metric, _p(perm, C.c_uint32), _p(out, C.c_float))
    return out, int(its)


def spann_search(centroids, list_off, ids, codes, codebook, num_probes, q, k, metric=METRIC_NDP):
    c = np.ascontiguousarray(centroids, dtype=np.float32)
    lo = np.ascontiguousarray(list_off, dtype=np.uint64)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, nc, sub = cb.shape
    q, pq_ = _f(q)
    oi = np.zeros(max(k, 1), np.uint32)
    od = np.zeros(max(k, 1), np.float32)
    m = lib().so_spann_search(_p(c, C.c_float), c.shape[0], c.shape[1], metric, _p(lo, C.c_uint64), _p(ids, C.c_uint32),
                              _p(codes, C.c_uint8), _p(cb, C.c_float), M, nc, sub, num_probes, pq_, k,
                              _p(oi, C.c_uint32), _p(od, C.c_float))
    return oi[:m].copy(), od[:m].copy()


def spann_build(vectors, num_partitions, ivf_perm, pq_perms, kmeans_iterations=25, metric=METRIC_NDP):
    """SpannIndex::build (spann.rs:363-463) given the two shuffles. Returns dict of trained state (CSR postings)."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = v.shape
    centroids, _ = spann_kmeans(v, num_partitions, kmeans_iterations, ivf_perm, metric)
    codebook = pq_train(v, pq_perms)
    assign = np.array([spann_find_nearest_centroid(v[i], centroids, metric) for i in range(n)], dtype=np.int64)
    codes_all = np.stack([pq_encode(codebook, v[i]) for i in range(n)]) if n else np.zeros((0, dim // 8), np.uint8)
    order = np.argsort(assign, kind="stable")      # within a partition: insertion (id) order
    counts = np.bincount(assign, minlength=num_partitions)
    list_off = np.zeros(num_partitions + 1, np.uint64)
    list_off[1:] = np.cumsum(counts)
    return dict(centroids=centroids, codebook=codebook, list_off=list_off,
                ids=order.astype(np.uint32), codes=codes_all[order], assign=assign)


# ---- MiniLM glue -----------------------------------------------------------------------
def mean_pool_finalize(hidden, mask):
    h = np.ascontiguousarray(hidden, dtype=np.float32)
    m = np.ascontiguousarray(mask, dtype=np.int64)
    out = np.zeros(h.shape[1], np.float32)
    lib().so_mean_pool_finalize(_p(h, C.c_float), _p(m, C.c_int64), h.shape[0], h.shape[1], _p(out, C.c_float))
    return out


def finalize_pooled(pooled, apply_prenorm=False, out_dim=None):
    p, pp = _f(pooled)
    out = np.zeros(max(p.size, 1), np.float32)
    m = lib().so_finalize_pooled(pp, p.size, int(apply_prenorm), p.size if out_dim is None else int(out_dim), _p(out, C.c_float))
    return out[:m].copy()


# ---- Vamana graph (vamana_oracle.c) -------------------------------------------------------------------------------------
class VamanaGraph:
    """deg [n] + nbr [n, cap] (cap = R + 1) + medoid over rows [n, dim]: the graph side of VamanaIndex, restated."""

    def __init__(self, dim, R=32, L=100, alpha=1.2, order=ORDER_SCALAR4, capacity=1024):
        self.dim, self.R, self.L, self.alpha, self.order = dim, R, L, np.float32(alpha), order
        self.cap = R + 1
        self.rows = np.zeros((capacity, dim), np.float32)
        self.deg = np.zeros(capacity, np.uint32)
        self.nbr = np.zeros((capacity, self.cap), np.uint32)
        self.n = 0
        self.medoid = 0

    def _grow(self, need):
        if need <= self.rows.shape[0]:
            return
        c = max(need, self.rows.shape[0] * 2)
        for name in ("rows", "deg", "nbr"):
            a = getattr(self, name)
            b = np.zeros((c,) + a.shape[1:], a.dtype)
            b[:a.shape[0]] = a
            setattr(self, name, b)

    def add_vector(self, v):
        """VamanaIndex::add_vector (vamana.rs:853-974)"""
        self._grow(self.n + 1)
        self.rows[self.n] = np.asarray(v, np.float32)
        lib().so_vamana_add_vector(_p(self.rows, C.c_float), self.n, self.dim, _p(self.deg, C.c_uint32), _p(self.nbr, C.c_uint32),
                                   self.cap, self.R, self.medoid, self.order)
        self.n += 1
        return self.n - 1

    def build(self, rows, init_deg, init_nbr):
        """VamanaIndex::build (vamana.rs:200-284) GIVEN the initial graph (init_nbr [n, <= R])"""
        rows = np.ascontiguousarray(rows, np.float32)
        n = rows.shape[0]
        self._grow(n)
        self.rows[:n] = rows
        self.n = n
        self.deg[:n] = init_deg
        self.nbr[:n] = 0
        self.nbr[:n, :init_nbr.shape[1]] = init_nbr
        self.medoid = int(lib().so_vamana_build(_p(self.rows, C.c_float), n, self.dim, _p(self.deg, C.c_uint32), _p(self.nbr, C.c_uint32),
                                                self.cap, self.R, self.L, self.alpha, self.order))
        return self.medoid

    def incremental_repair(self, start):
        """VamanaIndex::incremental_repair (vamana.rs:1033-1115) for nodes [start, n) -> nodes whose list changed"""
        return int(lib().so_vamana_incremental_repair(_p(self.rows, C.c_float), self.n, self.dim, _p(self.deg, C.c_uint32), _p(self.nbr, C.c_uint32),
                                                      self.cap, self.R, self.L, self.alpha, self.medoid, start, self.order))

    def greedy_search(self, q, k, entry=None):
        q, pq = _f(q)
        ids = np.zeros(max(k, 1), np.uint32); dist = np.zeros(max(k, 1), np.float32)
        m = lib().so_vamana_greedy_search(_p(self.rows, C.c_float), self.n, self.dim, _p(self.deg, C.c_uint32), _p(self.nbr, C.c_uint32), self.cap,
                                          pq, k, self.medoid if entry is None else entry, self.order, _p(ids, C.c_uint32), _p(dist, C.c_float))
        return ids[:m].copy(), dist[:m].copy()

    def search(self, q, k, deleted=None):
        """VamanaIndex::search without SHODH_VECTOR_EXACT (vamana.rs:764-808)"""
        q, pq = _f(q)
        ids = np.zeros(max(k, 1), np.uint32); dist = np.zeros(max(k, 1), np.float32)
        dp = None
        if deleted is not None:
            deleted = np.ascontiguousarray(deleted, np.uint8)
            dp = _p(deleted, C.c_uint8)
        m = lib().so_vamana_search(_p(self.rows, C.c_float), self.n, self.dim, _p(self.deg, C.c_uint32), _p(self.nbr, C.c_uint32), self.cap,
                                   self.medoid, dp, pq, k, self.order, _p(ids, C.c_uint32), _p(dist, C.c_float))
        return ids[:m].copy(), dist[:m].copy()


def siphash13_str(b: bytes):
    arr = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, np.uint8)
    return int(lib().so_siphash13_str(_p(np.ascontiguousarray(arr), C.c_uint8), len(b)))


def hash_embed(text: str, dim=384):
    b = text.encode("utf-8")
    out = np.zeros(dim, np.float32)
    lib().so_hash_embed(b, len(b), dim, _p(out, C.c_float))
    return out


# ---- relevance -------------------------------------------------------------------------
def weights_default():
    w = Weights()
    lib().so_weights_default(C.byref(w))
    return w


def weights_normalize(w):
    lib().so_weights_normalize(C.byref(w))
    return w


def weights_apply_feedback(w, sem, ent, tag, helpful):
    lib().so_weights_apply_feedback(C.byref(w), int(sem), int(ent), int(tag), int(helpful))
    return w


def calibrate_score(s):
    return np.float32(lib().so_calibrate_score(np.float32(s)))


def fuse_scores_full(w, sem, ent, tag, imp, mom, acc, gs):
    return np.float32(lib().so_fuse_scores_full(C.byref(w), sem, ent, tag, imp, mom, int(acc), gs))


def fuse_scores(w, sem, ent, tag, imp):
    return np.float32(lib().so_fuse_scores(C.byref(w), sem, ent, tag, imp))


def fuse_scores_with_momentum(w, sem, ent, tag, imp, mom):
    return np.float32(lib().so_fuse_scores_with_momentum(C.byref(w), sem, ent, tag, imp, mom))


def calculate_tag_score(content: str, tags):
    arr = (C.c_char_p * max(len(tags), 1))(*[t.encode() for t in tags])
    return np.float32(lib().so_calculate_tag_score(content.encode(), arr, len(tags)))


def apply_recency_boost(base, age_hours, boost_hours, mult):
    return np.float32(lib().so_apply_recency_boost(base, int(age_hours), int(boost_hours), mult))


def rank_surfaced(w, cands, min_importance=0.3, recency_boost_hours=24, recency_boost_multiplier=1.2, graph_boost_multiplier=1.15, max_results=5):
    """cands: list of dicts(semantic, entity, tag, importance, momentum, access_count, graph_strength, age_hours, created_at_ns, uuid)
    -> [(index, score, reason)]"""
    n = len(cands)
    f = lambda key: np.array([c[key] for c in cands] or [0], np.float32)
    sem, ent, tag, imp, mom, gs = (f(k_) for k_ in ("semantic", "entity", "tag", "importance", "momentum", "graph_strength"))
    acc = np.array([c["access_count"] for c in cands] or [0], np.uint32)
    age = np.array([c["age_hours"] for c in cands] or [0], np.int64)
    cre = np.array([c["created_at_ns"] for c in cands] or [0], np.int64)
    uu = np.frombuffer(b"".join(c["uuid"] for c in cands), np.uint8).copy() if n else np.zeros(16, np.uint8)
    oi = np.zeros(max(max_results, 1), np.uint32); os_ = np.zeros(max(max_results, 1), np.float32); orr = np.zeros(max(max_results, 1), np.uint8)
    m = lib().so_rank_surfaced(C.byref(w), min_importance, recency_boost_hours, recency_boost_multiplier, graph_boost_multiplier, max_results, n,
                               _p(sem, C.c_float), _p(ent, C.c_float), _p(tag, C.c_float), _p(imp, C.c_float), _p(mom, C.c_float), _p(acc, C.c_uint32),
                               _p(gs, C.c_float), age.ctypes.data_as(C.POINTER(C.c_int64)), cre.ctypes.data_as(C.POINTER(C.c_int64)), _p(uu, C.c_uint8),
                               _p(oi, C.c_uint32), _p(os_, C.c_float), _p(orr, C.c_uint8))
    return [(int(oi[i]), np.float32(os_[i]), int(orr[i])) for i in range(m)]


def rrf_fuse(k, weights, lists):
    """lists: list of lists of 16-byte uuids. Returns (uuids, scores) sorted (score desc, uuid asc)."""
    w = np.ascontiguousarray(weights, dtype=np.float32)
    flat = np.frombuffer(b"".join(b"".join(l) for l in lists), dtype=np.uint8) if any(len(l) for l in lists) else np.zeros(16, np.uint8)
    flat = np.ascontiguousarray(flat)
    lens = (C.c_size_t * len(lists))(*[len(l) for l in lists])
    cap = max(sum(len(l) for l in lists), 1)
    ou = np.zeros((cap, 16), np.uint8)
    os_ = np.zeros(cap, np.float32)
    m = lib().so_rrf_fuse(k, _p(w, C.c_float), len(lists), _p(flat, C.c_uint8), lens, _p(ou, C.c_uint8), _p(os_, C.c_float), cap)
    return [bytes(ou[i]) for i in range(m)], os_[:m].copy()


def density_weights(density):
    out = np.zeros(3, np.float32)
    lib().so_density_weights(density, _p(out, C.c_float))
    return out


def leg_fusion_weights(density=None, graph_weight_override=None, graph_w_floor=None):
    g, h = C.c_float(), C.c_float()
    nan = float("nan")
    lib().so_leg_fusion_weights(int(density is not None), 0.0 if density is None else density,
                                nan if graph_weight_override is None else graph_weight_override,
                                nan if graph_w_floor is None else graph_w_floor, C.byref(g), C.byref(h))
    return np.float32(g.value), np.float32(h.value)


LEG_SW_DEFAULT = dict(fusion_v2=0, fusion_flat=0, fusion_sum=0, fusion_rrf=0, isolate_leg=0, flat_adaptive=1, adapt_feature=0, adapt_symmetric=1)
LEG_FP_DEFAULT = dict(graph_w=0.3, hybrid_w=float(np.float32(0.6) + np.float32(0.1)), rrf_k=30.0, flat_consensus=0.3, adapt_trust_max=2.0,
                      fw_graph=0.3, fw_vec=0.6, fw_bm25=0.4, agree_k=10.0, agree_lo=0.1, agree_hi=0.5, peak_lo=2.0, peak_hi=6.0)


def fuse_legs(hybrid, graph, query_len, **cfg):
    """hybrid: [(uuid16, bm25, vec)] in hybrid rank order; graph: [(uuid16, activation)] in rank order.
    cfg: any key of LEG_SW_DEFAULT / LEG_FP_DEFAULT. Returns (uuids, scores, effective_vec_trust)."""
    sw = dict(LEG_SW_DEFAULT); fpv = dict(LEG_FP_DEFAULT)
    for k_, v in cfg.items():
        if k_ in sw: sw[k_] = int(v)
        elif k_ in fpv: fpv[k_] = float(v)
        else: raise KeyError(k_)
    swa = (C.c_int * 8)(*[sw[k_] for k_ in LEG_SW_DEFAULT])
    fpa = np.array([fpv[k_] for k_ in LEG_FP_DEFAULT], np.float32)
    hu = np.frombuffer(b"".join(h[0] for h in hybrid), np.uint8).copy() if hybrid else np.zeros(16, np.uint8)
    hb = np.array([h[1] for h in hybrid] or [0], np.float32); hv = np.array([h[2] for h in hybrid] or [0], np.float32)
    gu = np.frombuffer(b"".join(g[0] for g in graph), np.uint8).copy() if graph else np.zeros(16, np.uint8)
    ga = np.array([g[1] for g in graph] or [0], np.float32)
    cap = len(hybrid) + len(graph) + 1
    ou = np.zeros((cap, 16), np.uint8); os_ = np.zeros(cap, np.float32); tr = C.c_float()
    m = lib().so_fuse_legs(swa, _p(fpa, C.c_float), _p(hu, C.c_uint8), _p(hb, C.c_float), _p(hv, C.c_float), len(hybrid),
                           _p(gu, C.c_uint8), _p(ga, C.c_float), len(graph), int(query_len), _p(ou, C.c_uint8), _p(os_, C.c_float), cap, C.byref(tr))
    return [bytes(ou[i]) for i in range(m)], os_[:m].copy(), np.float32(tr.value)


def fnv1a64(b: bytes):
    arr = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, np.uint8)
    return int(lib().so_fnv1a64(_p(np.ascontiguousarray(arr), C.c_uint8), len(b)))

"""On-disk index formats of the reference (VAMA v1, SPAN v1) -> GPU indexes and back.

`vamana_persist.rs:290-391` (VamanaIndex::load_from_file) and `spann.rs:879-1003` (SpannIndex::load_from_file)
are mirrored by `load_vamana` / `load_spann`; the parsing itself is C++ behind the C ABI (csrc/persist.hip).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .index import DistanceMetric, SpannIndex, VamanaConfig, VamanaIndex

_METRIC = {0: DistanceMetric.NormalizedDotProduct, 1: DistanceMetric.Euclidean, 2: DistanceMetric.Cosine}


def vama_info(path):
    o = L.VamaInfo()
    L.check(L.lib().shodh_vama_info_read(str(path).encode(), C.byref(o)))
    return {f: getattr(o, f) for f, _ in o._fields_}


def verify_index_file(path):
    """VamanaIndex::verify_index_file (vamana_persist.rs:410-424): False on a checksum mismatch, raises on a bad header."""
    o = L.VamaInfo()
    rc = L.lib().shodh_vama_info_read(str(path).encode(), C.byref(o))
    if rc == L.OK:
        return True
    msg = L.lib().shodh_last_error().decode()
    if "checksum mismatch" in msg:
        return False
    raise L.ShodhError(rc, msg)


def read_vamana(path, with_graph=False):
    """-> dict(info, vectors [n,dim] f32, deleted [d] u32[, degree [n] u16, neighbors [edges] u32])"""
    info = vama_info(path)
    n, d = info["num_vectors"], info["dimension"]
    vec = np.empty((n, d), np.float32)
    dele = np.empty(info["deleted_count"], np.uint32)
    deg = np.empty(n, np.uint16) if with_graph else None
    nb = np.empty(info["graph_edges"], np.uint32) if with_graph else None
    L.check(L.lib().shodh_vama_load(str(path).encode(), vec.ctypes.data, dele.ctypes.data,
                                    deg.ctypes.data if with_graph else None, nb.ctypes.data if with_graph else None))
    out = dict(info=info, vectors=vec, deleted=dele)
    if with_graph:
        out.update(degree=deg, neighbors=nb)
    return out


def write_vamana(path, vectors, max_degree=32, medoid=0, metric=0, deleted=(), incremental_inserts=0, degree=None, neighbors=None):
    v = np.ascontiguousarray(vectors, np.float32)
    dele = np.ascontiguousarray(np.asarray(list(deleted), np.uint32))
    deg = None if degree is None else np.ascontiguousarray(degree, np.uint16)
    nb = None if neighbors is None else np.ascontiguousarray(neighbors, np.uint32)
    L.check(L.lib().shodh_vama_save(str(path).encode(), v.ctypes.data, v.shape[0], v.shape[1] if v.ndim == 2 else 0, max_degree, medoid, metric,
                                    dele.ctypes.data if dele.size else None, dele.size, incremental_inserts,
                                    deg.ctypes.data if deg is not None else None, nb.ctypes.data if nb is not None else None))


def load_vamana(path, device=0, scan_mode=L.SCAN_AUTO):
    """A persisted Vamana index -> flat GPU index: same vectors, same ids, same tombstones. The graph is not needed to
    answer (this library runs the exact scan, the reference's own `brute_force_search` semantics) but it is KEPT on the
    index object, so that `save_to_file` of an index whose rows did not change writes the reference-built graph back."""
    f = read_vamana(path, with_graph=True)
    info = f["info"]
    if info["distance_metric"] != 0:
        raise L.ShodhError(L.ERR_UNSUPPORTED, "only NormalizedDotProduct indexes are served (retrieval.rs:188-193)")
    idx = VamanaIndex(VamanaConfig(dimension=info["dimension"], max_degree=info["max_degree"], device=device, scan_mode=scan_mode,
                                   reserve_rows=max(int(info["num_vectors"]), 1)))
    if info["num_vectors"]:
        if scan_mode == L.SCAN_GRAPH:
            # the file's graph IS the index: store the rows, attach the adjacency lists and the medoid as written
            idx.set_graph(f["degree"].astype(np.uint32), f["neighbors"], int(info["medoid"]), vectors=f["vectors"])
        else:
            idx.build(f["vectors"])
    for i in f["deleted"]:
        idx.mark_deleted(int(i))
    # A file WITHOUT edges carries `incremental_inserts >= REBUILD_THRESHOLD` as a message to a graph-walking reader ("rebuild me", see
    # save_vamana); an exact-scan index has no graph to rebuild, so it must not inherit that sentinel -- it would make needs_rebuild()
    # true forever and the next auto_rebuild_if_needed() would renumber ids under the caller.
    from .index import REBUILD_THRESHOLD
    sentinel = int(info["graph_edges"]) == 0 and int(info["num_vectors"]) > 1 and int(info["incremental_inserts"]) >= REBUILD_THRESHOLD
    idx._incremental = 0 if (sentinel and scan_mode != L.SCAN_GRAPH) else int(info["incremental_inserts"])
    idx._graph = dict(n=int(info["num_vectors"]), medoid=int(info["medoid"]), degree=f["degree"], neighbors=f["neighbors"])
    return idx


def save_vamana(index, path):
    """VamanaIndex::save_to_file (vamana_persist.rs:175-284) for an index of this library.

    Three cases:
    * graph mode (`SHODH_SCAN_GRAPH`): the index holds the reference's graph (grown by `add_vector`, built by `vamana_build` or loaded);
      degree / neighbour arrays and medoid are read back from the device and written;
    * exact-scan modes, index loaded from a reference-written file and its rows unchanged (`index._graph`): the loaded degree /
      neighbour arrays and medoid are written back, so the file round-trips with its graph;
    * otherwise every node gets an EMPTY adjacency list, and `incremental_inserts` is raised to REBUILD_THRESHOLD in the
      header: the reference's default (non-exact) `search` would walk a graph without edges and return one hit per query
      (vamana.rs:780-806), so the file says "needs_rebuild()" (vamana.rs:985-993) and the reference's maintenance path
      (`auto_rebuild_if_needed`, retrieval.rs:1633-1656) rebuilds the graph from the vectors. Under SHODH_VECTOR_EXACT the
      file is served as is.
    Tombstones are collected by GLOBAL id (`id_base + row`): a shard of a row-sharded corpus keeps its own."""
    from .index import REBUILD_THRESHOLD
    vec = index.extract_all_vectors()
    vec = np.ascontiguousarray(np.asarray(vec, np.float32)).reshape(-1, index.config.dimension)
    base = int(index.config.id_base)
    deleted = [i for i in range(vec.shape[0]) if index.is_deleted(base + i)] if index.deleted_count() else []
    g = getattr(index, "_graph", None)
    if getattr(index, "graph_mode", False) and vec.shape[0]:
        deg, nb, medoid = index.get_graph()
        flat = nb[np.arange(nb.shape[1])[None, :] < deg[:, None]]
        write_vamana(path, vec, max_degree=index.config.max_degree, medoid=medoid, metric=0, deleted=deleted,
                     incremental_inserts=index.incremental_insert_count(), degree=deg.astype(np.uint16), neighbors=np.ascontiguousarray(flat, np.uint32))
    elif g is not None and g["n"] == vec.shape[0] and (len(g["neighbors"]) > 0 or vec.shape[0] <= 1):      # a loaded file WITH edges round-trips with them
        write_vamana(path, vec, max_degree=index.config.max_degree, medoid=g["medoid"], metric=0, deleted=deleted,
                     incremental_inserts=index.incremental_insert_count(), degree=g["degree"], neighbors=g["neighbors"])
    else:
        write_vamana(path, vec, max_degree=index.config.max_degree, medoid=0, metric=0, deleted=deleted,
                     incremental_inserts=max(index.incremental_insert_count(), REBUILD_THRESHOLD if vec.shape[0] > 1 else 0))


def span_info(path):
    o = L.SpanInfo()
    L.check(L.lib().shodh_span_info_read(str(path).encode(), C.byref(o)))
    return {f: getattr(o, f) for f, _ in o._fields_}


def read_spann(path):
    info = span_info(path)
    P, D, M, T = info["num_partitions"], info["dimension"], info["pq_subvectors"], info["total_postings"]
    pq = info["pq_enabled"] == 1
    cent = np.empty((P, D), np.float32)
    cb = np.empty((M, info["pq_num_centroids"], info["pq_subvec_dim"]), np.float32) if pq else None
    off = np.empty(P + 1, np.uint64)
    ids = np.empty(T, np.uint32)
    codes = np.empty((T, M), np.uint8) if pq else None
    L.check(L.lib().shodh_span_load(str(path).encode(), cent.ctypes.data, cb.ctypes.data if pq else None, off.ctypes.data, ids.ctypes.data,
                                    codes.ctypes.data if pq else None))
    return dict(info=info, centroids=cent, codebook=cb, list_off=off, ids=ids, codes=codes)


def write_spann(path, num_vectors, centroids, codebook, list_off, ids, codes, metric=0):
    cent = np.ascontiguousarray(centroids, np.float32)
    off = np.ascontiguousarray(list_off, np.uint64)
    ids = np.ascontiguousarray(ids, np.uint32)
    cb = None if codebook is None else np.ascontiguousarray(codebook, np.float32)
    cd = None if codes is None else np.ascontiguousarray(codes, np.uint8)
    M = 0 if cb is None else cb.shape[0]
    L.check(L.lib().shodh_span_save(str(path).encode(), num_vectors, cent.shape[0], cent.shape[1], M, metric, cent.ctypes.data,
                                    cb.ctypes.data if cb is not None else None, off.ctypes.data, ids.ctypes.data if ids.size else None,
                                    cd.ctypes.data if cd is not None else None))


def load_spann(path, num_probes=20, device=0):
    """A persisted SPANN index -> GPU IVF-PQ index (centroids, codebook and postings exactly as stored)."""
    f = read_spann(path)
    info = f["info"]
    if info["pq_enabled"] != 1:
        raise L.ShodhError(L.ERR_UNSUPPORTED, "PQ-less SPANN files carry no codes to scan (SpannIndex::build rejects use_pq=false, spann.rs:386-392)")
    idx = SpannIndex(info["dimension"], num_probes=num_probes, distance_metric=_METRIC.get(info["distance_metric"], DistanceMetric.NormalizedDotProduct), device=device)
    idx.set_trained_state(f["centroids"], f["codebook"], f["list_off"], f["ids"], f["codes"])
    return idx

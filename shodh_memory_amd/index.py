"""Host-side mirror of the reference's vector-index API over the C ABI.

Mirrors `VamanaIndex` (src/vector_db/vamana.rs:168-1645) and the `VectorIndexBackend` facade
(src/vector_db/mod.rs:98-266): same method names, argument meaning and error behaviour, so the
parity tests read like the reference's own tests. Results follow the reference's exact path
(`SHODH_VECTOR_EXACT`, vamana.rs:770-777): `search` returns `[(id, distance)]`, distance =
-dot ascending, ties by id.

numpy arrays go through the host entry points; torch CUDA tensors go through the `*_device`
entry points without leaving HBM.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass
from enum import Enum

import numpy as np

from . import _lib as L

# vamana.rs:103-117
MIN_ACCEPTABLE_RECALL = 0.85            # vamana.rs:115
REBUILD_THRESHOLD = 10_000
REPAIR_THRESHOLD = 1_000
DELETION_RATIO_THRESHOLD = 0.30
# vector_db/mod.rs:53
SPANN_AUTO_THRESHOLD = 100_000


class DistanceMetric(Enum):
    NormalizedDotProduct = 0
    Euclidean = 1
    Cosine = 2


class BackendType(Enum):
    Vamana = 0
    Spann = 1


@dataclass
class VamanaConfig:
    """vamana.rs:56-90 VamanaConfig. The graph parameters matter under `scan_mode=SCAN_GRAPH` (the reference's default ANN
    `search`, reproduced on the device); the exact scan modes -- SHODH_VECTOR_EXACT semantics, vamana.rs:1167-1188 -- ignore them."""
    dimension: int = 384
    max_degree: int = 32
    search_list_size: int = 75
    alpha: float = 1.2
    use_mmap: bool = False
    distance_metric: DistanceMetric = DistanceMetric.NormalizedDotProduct
    # device-side knobs (not in the reference)
    device: int = 0
    order: int = L.ORDER_SCALAR4
    scan_mode: int = L.SCAN_AUTO
    reserve_rows: int = 0
    id_base: int = 0


def env_devices():
    """SHODH_HIP_DEVICES="0,2,3" -> [0, 2, 3] (the HIP devices a process may use); unset -> [0]"""
    v = os.environ.get("SHODH_HIP_DEVICES", "").strip()
    return [int(x) for x in v.split(",") if x.strip() != ""] if v else [0]


def env_dimension(default=384):
    """SHODH_TEXT_DIM (minilm.rs:313-322): one of 128 / 256 / 384 / 512 / 768 / 1024, anything else falls back to the default"""
    try:
        d = int(os.environ.get("SHODH_TEXT_DIM", ""))
    except ValueError:
        return default
    return d if d in (128, 256, 384, 512, 768, 1024) else default


def vamana_config_from_env(**overrides):
    """VamanaConfig the way a deployment configures the reference -- by environment:
    SHODH_TEXT_DIM -> dimension; SHODH_HIP_DEVICES -> device (the first one listed);
    SHODH_VECTOR_EXACT set (vamana.rs:770-777) -> the exact scan, which is ALSO this library's default when nothing is set;
    SHODH_HIP_GRAPH_WALK=1 -> the reference's graph walk on the device (its behaviour WITHOUT SHODH_VECTOR_EXACT), unless
    SHODH_VECTOR_EXACT is set as well (the reference's switch wins)."""
    graph = os.environ.get("SHODH_HIP_GRAPH_WALK", "") not in ("", "0") and "SHODH_VECTOR_EXACT" not in os.environ
    kw = dict(dimension=env_dimension(), device=env_devices()[0], scan_mode=L.SCAN_GRAPH if graph else L.SCAN_AUTO)
    kw.update(overrides)
    return VamanaConfig(**kw)


@dataclass
class BackendConfig:
    """vector_db/mod.rs:57-86"""
    dimension: int = 384
    distance_metric: DistanceMetric = DistanceMetric.NormalizedDotProduct
    force_backend: "BackendType | None" = None
    use_pq: bool = True
    spann_probes: int = 20
    vamana_max_degree: int = 32
    vamana_search_list_size: int = 100
    device: int = 0


def _is_torch_cuda(x):
    return hasattr(x, "is_cuda") and x.is_cuda


def _check_device_rows(t, dim):
    """device tensors are handed over as raw pointers: anything but contiguous float32 [n, dim] would be misread"""
    import torch
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != dim or not t.is_contiguous():
        raise L.ShodhError(L.ERR_INVALID, "device rows must be a contiguous float32 [n, %d] tensor (got %s %s)" % (dim, t.dtype, tuple(t.shape)))


def _as_rows(vectors, dim):
    a = np.ascontiguousarray(vectors, dtype=np.float32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.size and a.shape[1] != dim:
        raise L.ShodhError(L.ERR_DIM, "Vector dimension %d doesn't match config %d" % (a.shape[1], dim))
    return a.reshape(-1, dim)


class _Handle:
    def __init__(self, cfg: L.IndexCfg):
        self._h = C.c_void_p()
        L.check(L.lib().shodh_index_create(C.byref(cfg), C.byref(self._h)))
        self.dim = cfg.dim
        self.device = cfg.device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib().shodh_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VamanaIndex:
    """The VamanaIndex method set on one MI355X: an exact scan by default (the reference under SHODH_VECTOR_EXACT), or the
    reference's own graph walk -- same graph, same visits, same answers -- with `VamanaConfig(scan_mode=SCAN_GRAPH)`."""

    def __init__(self, config: VamanaConfig = None):
        self.config = config or VamanaConfig()
        cfg = L.IndexCfg()
        L.lib().shodh_index_cfg_default(C.byref(cfg))
        cfg.dim = self.config.dimension
        cfg.metric = self.config.distance_metric.value
        cfg.kind = L.INDEX_FLAT
        cfg.order = self.config.order
        cfg.device = self.config.device
        cfg.scan_mode = self.config.scan_mode
        cfg.reserve_rows = self.config.reserve_rows
        cfg.id_base = self.config.id_base
        cfg.max_degree = self.config.max_degree
        cfg.search_list_size = self.config.search_list_size
        cfg.alpha = self.config.alpha
        self._hd = _Handle(cfg)
        self._incremental = 0
        self._graph = None            # degree / neighbour arrays of a reference-built file (persist.load_vamana), while the rows are unchanged

    # -- VamanaIndex::new / with_storage_path ----------------------------------------------------
    @classmethod
    def new(cls, config: VamanaConfig):
        return cls(config)

    @classmethod
    def with_storage_path(cls, config: VamanaConfig, storage_path=None):
        """vamana.rs:175-187: the path only tells the reference where to mmap its vectors; rows live in HBM here"""
        idx = cls(config)
        idx.storage_path = storage_path
        return idx

    @property
    def handle(self):
        return self._hd._h

    def close(self):
        self._hd.close()

    # -- len / is_empty (vamana.rs:184-191) --------------------------------------------------------
    def len(self):
        return int(L.lib().shodh_index_len(self.handle))

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    # -- build (vamana.rs:200-284) / rebuild_from_vectors (:1363-1462) -------------------------------
    def build(self, vectors):
        if _is_torch_cuda(vectors):
            _check_device_rows(vectors, self._hd.dim)
            L.check(L.lib().shodh_index_build_device(self.handle, vectors.data_ptr(), vectors.shape[0]))
        else:
            a = _as_rows(vectors, self._hd.dim)
            L.check(L.lib().shodh_index_build(self.handle, a.ctypes.data, a.shape[0]))
        self._incremental = 0
        self._graph = None

    rebuild_from_vectors = build

    # -- the graph itself (SCAN_GRAPH only) ---------------------------------------------------------------
    @property
    def graph_mode(self):
        return self.config.scan_mode == L.SCAN_GRAPH

    def vamana_build(self, seed=0, init_degree=None, init_neighbors=None):
        """VamanaIndex::build's graph construction (vamana.rs:200-284) over the rows already stored. The reference draws its start
        graph from thread_rng (:287-312); here it comes from `seed`, or is handed in as (`init_degree` [n] uint32,
        `init_neighbors` [n, stride] uint32) -- which is how the parity tests pin the construction."""
        if init_degree is not None:
            d = np.ascontiguousarray(init_degree, np.uint32)
            nb = np.ascontiguousarray(init_neighbors, np.uint32).reshape(d.shape[0], -1)
            L.check(L.lib().shodh_index_vamana_build(self.handle, int(seed), d.ctypes.data, nb.ctypes.data, nb.shape[1]))
        else:
            L.check(L.lib().shodh_index_vamana_build(self.handle, int(seed), None, None, 0))
        self._incremental = 0

    def get_graph(self):
        """-> (degree [n] uint32, neighbours [n, max_degree + 1] uint32, medoid); entries past a node's degree are 0"""
        n, st = self.len(), self.config.max_degree + 1
        deg = np.zeros(n, np.uint32); nb = np.zeros((n, st), np.uint32); med = C.c_uint32()
        L.check(L.lib().shodh_index_get_graph(self.handle, deg.ctypes.data, nb.ctypes.data, st, C.byref(med)))
        return deg, nb, int(med.value)

    def set_graph(self, degree, neighbors, medoid, vectors=None):
        """attach a graph built elsewhere (a reference-written VAMA file): `neighbors` [n, stride] or the flat concatenation.
        With `vectors` the index contents are replaced by those rows first (load_from_file)."""
        d = np.ascontiguousarray(degree, np.uint32)
        nb = np.asarray(neighbors, np.uint32)
        if nb.ndim == 1:                                               # flat adjacency lists, VAMA order
            st = max(int(d.max()) if d.size else 0, 1)
            sq = np.zeros((d.shape[0], st), np.uint32)
            off = np.concatenate([[0], np.cumsum(d, dtype=np.int64)])
            cols = np.arange(nb.shape[0], dtype=np.int64) - np.repeat(off[:-1], d)
            sq[np.repeat(np.arange(d.shape[0]), d), cols] = nb
            nb = sq
        nb = np.ascontiguousarray(nb)
        if vectors is not None:
            a = _as_rows(vectors, self._hd.dim)
            L.check(L.lib().shodh_index_build_with_graph(self.handle, a.ctypes.data, a.shape[0], d.ctypes.data, nb.ctypes.data, nb.shape[1], int(medoid)))
            self._incremental = 0
            return
        L.check(L.lib().shodh_index_set_graph(self.handle, d.ctypes.data, nb.ctypes.data, nb.shape[1], int(medoid)))

    # -- add_vector (vamana.rs:853-974): returns the new id -------------------------------------------
    def add_vector(self, vector):
        a = _as_rows(vector, self._hd.dim)
        if a.shape[0] != 1:
            raise L.ShodhError(L.ERR_INVALID, "add_vector takes one vector; use add_vectors")
        first = C.c_uint32()
        L.check(L.lib().shodh_index_add(self.handle, a.ctypes.data, 1, C.byref(first)))
        self._graph = None
        self._incremental += 1 if first.value > self.config.id_base else 0     # the vector that seeds an empty index is not an incremental insert (vamana.rs:888-898)
        return int(first.value)

    def add_vectors(self, vectors):
        """n sequential add_vector calls in one transfer; returns the first id."""
        first = C.c_uint32()
        self._graph = None
        if _is_torch_cuda(vectors):
            _check_device_rows(vectors, self._hd.dim)
            L.check(L.lib().shodh_index_add_device(self.handle, vectors.data_ptr(), vectors.shape[0], C.byref(first)))
            n = int(vectors.shape[0])
        else:
            a = _as_rows(vectors, self._hd.dim)
            L.check(L.lib().shodh_index_add(self.handle, a.ctypes.data, a.shape[0], C.byref(first)))
            n = int(a.shape[0])
        self._incremental += n - (1 if n and first.value == self.config.id_base else 0)      # see add_vector
        return int(first.value)

    def graph_overflowed(self):
        """graph mode: some add since the last build met a walk whose frontier outgrew its array (thousands of equidistant rows). The rows were added
        and the counters above are in step with the device; the graph may differ from the reference's from there on (shodh_index_graph_overflowed)."""
        return bool(L.lib().shodh_index_graph_overflowed(self.handle))

    # -- search (vamana.rs:764-808; exact path :1167-1188) ---------------------------------------------
    def search(self, query, k):
        """-> list[(id, distance)] ascending distance, ties by id; at most k; [] on an empty index."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != self._hd.dim:
            raise L.ShodhError(L.ERR_DIM, "Query dimension %d doesn't match index dimension %d" % (q.size, self._hd.dim))
        ids, dist, counts = self.search_batch(q.reshape(1, -1), k)
        n = int(counts[0])
        return [(int(ids[0, i]), float(dist[0, i])) for i in range(n)]

    def search_batch(self, queries, k):
        """queries [nq, dim] -> (ids [nq,k] uint32, dist [nq,k] float32, counts [nq])."""
        if _is_torch_cuda(queries):
            return self.search_batch_device(queries, k)
        q = _as_rows(queries, self._hd.dim)
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        L.check(L.lib().shodh_index_search(self.handle, q.ctypes.data, nq, k, ids.ctypes.data, dist.ctypes.data, counts.ctypes.data))
        return ids[:, :k], dist[:, :k], counts

    def search_batch_device(self, queries, k, out=None, stream=None):
        """torch CUDA tensors in, torch CUDA tensors out, asynchronous on the current stream."""
        import torch
        _check_device_rows(queries, self._hd.dim)
        nq = queries.shape[0]
        if out is None:
            ids = torch.empty((nq, k), dtype=torch.int32, device=queries.device)     # bit pattern of u32 ids
            dist = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
            counts = torch.empty((nq,), dtype=torch.int32, device=queries.device)
        else:
            ids, dist, counts = out
        st = stream if stream is not None else torch.cuda.current_stream(queries.device).cuda_stream
        L.check(L.lib().shodh_index_search_device(self.handle, queries.data_ptr(), nq, k, ids.data_ptr(), dist.data_ptr(),
                                                  counts.data_ptr(), C.c_void_p(st)))
        if self.graph_mode:
            counts.bitwise_and_(0x7FFFFFFF)        # bit 31 = "this walk's frontier overflowed" (include/shodh_hip.h); the host-pointer API raises instead
        return ids, dist, counts

    # -- tombstones (vamana.rs:813-850) ------------------------------------------------------------------
    def mark_deleted(self, vector_id):
        ok = C.c_int()
        L.check(L.lib().shodh_index_mark_deleted(self.handle, int(vector_id), C.byref(ok)))
        return bool(ok.value)

    def mark_deleted_many(self, vector_ids):
        """n mark_deleted calls in one (returns how many ids were valid and newly tombstoned)"""
        ids = np.ascontiguousarray(vector_ids, np.uint32)
        m = C.c_uint64()
        L.check(L.lib().shodh_index_mark_deleted_batch(self.handle, ids.ctypes.data, ids.size, C.byref(m)))
        return int(m.value)

    def is_deleted(self, vector_id):
        return bool(L.lib().shodh_index_is_deleted(self.handle, int(vector_id)))

    def deleted_count(self):
        return int(L.lib().shodh_index_deleted_count(self.handle))

    def deletion_ratio(self):
        return float(L.lib().shodh_index_deletion_ratio(self.handle))

    def needs_compaction(self):
        return bool(L.lib().shodh_index_needs_compaction(self.handle))

    def clear_deleted(self):
        L.check(L.lib().shodh_index_clear_deleted(self.handle))

    # -- rebuild bookkeeping (vamana.rs:1190-1340): the flat index never degrades -------------------------
    def needs_rebuild(self):
        return self._incremental >= REBUILD_THRESHOLD or self.needs_compaction()

    def incremental_insert_count(self):
        return self._incremental

    def reset_incremental_counter(self):
        self._incremental = 0

    def is_rebuilding(self):
        return False

    def needs_repair(self):
        return REPAIR_THRESHOLD <= self._incremental < REBUILD_THRESHOLD             # vamana.rs:1010-1016

    def incremental_repair(self):
        """vamana.rs:1033-1115: once REPAIR_THRESHOLD inserts have accumulated, the last min(inserts, REPAIR_THRESHOLD) nodes are
        re-pruned (walk + robust_prune + back edges) -- on the device in graph mode; the exact index has no graph to repair (0 nodes).
        The counter moves as the reference's does. Returns the number of nodes whose list changed."""
        n = self.len()
        if n == 0 or self._incremental < REPAIR_THRESHOLD:
            return 0
        repaired = 0
        if self.graph_mode:
            count = min(self._incremental, REPAIR_THRESHOLD, n)
            r = C.c_uint32()
            L.check(L.lib().shodh_index_incremental_repair(self.handle, n - count, count, C.byref(r)))
            repaired = int(r.value)
        self._incremental = max(0, self._incremental - REPAIR_THRESHOLD)
        return repaired

    def brute_force_search_batch(self, queries, k):
        """VamanaIndex::brute_force_search (vamana.rs:1167-1188) whatever the scan mode -> (ids, dist, counts)"""
        q = _as_rows(queries, self._hd.dim)
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        L.check(L.lib().shodh_index_brute_force_search(self.handle, q.ctypes.data, nq, k, ids.ctypes.data, dist.ctypes.data, counts.ctypes.data))
        return ids[:, :k], dist[:, :k], counts

    def estimate_recall(self, sample_size=100, k=10, rng=None):
        """vamana.rs:1128-1165: recall@k of `search` against `brute_force_search` over randomly sampled stored vectors. With the exact
        scan the two are the same scan (1.0 by construction); in graph mode both run on the device and are compared."""
        n = self.len()
        if n < 2 or not self.graph_mode:
            return 1.0
        sample_size = max(min(sample_size, n // 2), 1)
        k = min(k, n - 1)
        rng = rng or np.random.default_rng()
        pick = rng.permutation(n)[:sample_size]                                  # the reference shuffles with thread_rng
        rows = np.empty((sample_size, self._hd.dim), np.float32)
        for j, i in enumerate(pick):
            L.check(L.lib().shodh_index_extract_rows(self.handle, int(i), 1, rows[j].ctypes.data))
        a_ids, _, a_cnt = self.search_batch(rows, k)
        e_ids, _, e_cnt = self.brute_force_search_batch(rows, k)
        total = 0.0
        for j in range(sample_size):
            total += len(set(a_ids[j, :int(a_cnt[j])].tolist()) & set(e_ids[j, :int(e_cnt[j])].tolist())) / float(k)
        return total / sample_size

    def quality_degraded(self):
        """vamana.rs:1194-1210: too small / no incremental inserts -> False; else estimated recall@10 over 50 samples below MIN_ACCEPTABLE_RECALL"""
        if self.len() < 100 or self._incremental == 0:
            return False
        return self.estimate_recall(50, 10) < MIN_ACCEPTABLE_RECALL

    def auto_maintain(self):
        """vamana.rs:1217-1232"""
        if self.needs_rebuild():
            return "full_rebuild" if self.auto_rebuild_if_needed() else "rebuild_skipped"
        if self.needs_repair():
            return "repaired_%d_nodes" % self.incremental_repair()
        return "no_action"

    def auto_rebuild_if_needed(self):
        """Compacts tombstoned rows away when the reference would rebuild. Returns True if it did."""
        if not self.needs_rebuild():
            return False
        live = self.extract_live_vectors()
        self.build(live)
        return True

    # -- extraction (retrieval.rs:2504-2516: rows come back bit-for-bit) -----------------------------------
    def extract_all_vectors(self):
        n = self.len()
        out = np.empty((n, self._hd.dim), np.float32)
        if n:
            L.check(L.lib().shodh_index_extract_rows(self.handle, 0, n, out.ctypes.data))
        return out

    def extract_live_vectors(self):
        n = C.c_uint64()
        L.check(L.lib().shodh_index_extract_live_rows(self.handle, None, None, 0, C.byref(n)))
        out = np.empty((n.value, self._hd.dim), np.float32)
        if n.value:
            L.check(L.lib().shodh_index_extract_live_rows(self.handle, out.ctypes.data, None, n.value, C.byref(n)))
        return out

    # -- persistence (vamana_persist.rs:175-424): VAMA v1 --------------------------------------------------------
    def save_to_file(self, path):
        from . import persist
        persist.save_vamana(self, path)

    @classmethod
    def load_from_file(cls, path, device=0):
        from . import persist
        return persist.load_vamana(path, device=device)

    @staticmethod
    def verify_index_file(path):
        from . import persist
        return persist.verify_index_file(path)

    # -- diagnostics -------------------------------------------------------------------------------------------
    def stage_timings_us(self):
        a = (C.c_float * 4)()
        L.check(L.lib().shodh_index_stage_timings(self.handle, C.byref(a)))
        return dict(scan=a[0], select=a[1], other=a[2], total=a[3])

    def kernel_timing(self, reset=True):
        """(mean_us, min_us, count) of the dominant scan kernel since the last reset; synchronise first."""
        m, mn, c = C.c_float(), C.c_float(), C.c_uint32()
        L.check(L.lib().shodh_index_kernel_timing(self.handle, int(reset), C.byref(m), C.byref(mn), C.byref(c)))
        return m.value, mn.value, c.value

    def set_coalesce(self, enabled, linger_us=30):
        """concurrent host-pointer searches of a few queries share one pass (shodh_index_set_coalesce; on by default)"""
        L.check(L.lib().shodh_index_set_coalesce(self.handle, int(bool(enabled)), int(linger_us)))

    def coalesce_stats(self, reset=False):
        a = (C.c_uint64 * 6)()
        L.check(L.lib().shodh_index_coalesce_stats(self.handle, C.byref(a), int(bool(reset))))
        return dict(passes=int(a[0]), calls=int(a[1]), largest=int(a[2]), lingered=int(a[3]), pass_us=int(a[4]), linger_us=int(a[5]))

    def scan_stats(self):
        a = (C.c_uint64 * 8)()
        L.check(L.lib().shodh_index_scan_stats(self.handle, C.byref(a)))
        return dict(sampled_rows=a[0], emitted=a[1], rescored=a[2], overflowed=a[3], level2=a[4])


class SpannIndex:
    """IVF-PQ index with the SpannIndex method set (src/vector_db/spann.rs), given trained state."""

    def __init__(self, dimension=384, num_probes=10, distance_metric=DistanceMetric.NormalizedDotProduct, device=0):
        cfg = L.IndexCfg()
        L.lib().shodh_index_cfg_default(C.byref(cfg))
        cfg.dim = dimension
        cfg.metric = distance_metric.value
        cfg.kind = L.INDEX_IVFPQ
        cfg.device = device
        cfg.nprobe = num_probes
        self._hd = _Handle(cfg)
        self.num_probes = num_probes
        self._metric = distance_metric.value
        self._state = None
        self._pending = []

    @property
    def handle(self):
        return self._hd._h

    def close(self):
        self._hd.close()

    def len(self):
        return int(L.lib().shodh_index_len(self.handle))       # postings held (SpannIndex::len)

    def is_empty(self):
        return self.len() == 0

    def set_trained_state(self, centroids, codebook, list_off, ids, codes):
        c = np.ascontiguousarray(centroids, np.float32)
        cb = np.ascontiguousarray(codebook, np.float32)
        lo = np.ascontiguousarray(list_off, np.uint64)
        i = np.ascontiguousarray(ids, np.uint32)
        cd = np.ascontiguousarray(codes, np.uint8)
        M, ncent, sub = cb.shape
        assert sub == 8
        L.check(L.lib().shodh_index_set_ivfpq(self.handle, c.ctypes.data, c.shape[0], cb.ctypes.data, M, ncent,
                                              lo.ctypes.data, i.ctypes.data, cd.ctypes.data))
        self._state = dict(centroids=c, codebook=cb, list_off=lo, ids=i, codes=cd.reshape(-1, M))   # host copy, for save_to_file
        self._pending = []

    @staticmethod
    def compute_partitions(num_vectors):
        """SpannConfig::compute_partitions (spann.rs:135-139) with num_partitions unset: ceil(sqrt(n)) as f64, at least 1"""
        return max(1, int(math.ceil(math.sqrt(float(num_vectors)))))

    def num_partitions(self):
        return 0 if self._state is None else int(self._state["centroids"].shape[0])

    def train(self, vectors, num_partitions=None, kmeans_iterations=25, pq_iterations=20, seed=None, ivf_perm=None, pq_perms=None):
        """The two k-means of SpannIndex::build on the device (spann.rs:466-541, pq.rs:152-217). The reference draws its
        initial shuffles from thread_rng; here they come from `seed` (or are passed in), and GIVEN the shuffles the
        result is bit-identical to the reference's arithmetic. -> (centroids [P,dim], codebook [dim/8,256,8])"""
        v = _as_rows(vectors, self._hd.dim)
        n, dim = v.shape
        P = num_partitions or self.compute_partitions(n)
        rng = np.random.default_rng(seed)
        ip = np.ascontiguousarray(rng.permutation(n) if ivf_perm is None else ivf_perm, np.uint32)
        pp = np.ascontiguousarray(np.stack([rng.permutation(n) for _ in range(dim // 8)]) if pq_perms is None else np.stack(pq_perms), np.uint32)
        cent = np.zeros((P, dim), np.float32)
        cb = np.zeros((dim // 8, 256, 8), np.float32)
        L.check(L.lib().shodh_ivfpq_train(self._hd.device, v.ctypes.data, n, dim, P, kmeans_iterations, pq_iterations,
                                          ip.ctypes.data, pp.ctypes.data, cent.ctypes.data, cb.ctypes.data))
        return cent, cb

    def build(self, vectors, num_partitions=None, kmeans_iterations=25, pq_iterations=20, seed=None, ivf_perm=None, pq_perms=None):
        """SpannIndex::build (spann.rs:363-463): train, then assign + PQ-encode every vector into its posting list
        (ids = positions, insertion order inside a list)."""
        v = _as_rows(vectors, self._hd.dim)
        cent, cb = self.train(v, num_partitions, kmeans_iterations, pq_iterations, seed, ivf_perm, pq_perms)
        P = cent.shape[0]
        self.set_trained_state(cent, cb, np.zeros(P + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, v.shape[1] // 8), np.uint8))
        assign, codes = self.encode(v)
        order = np.argsort(assign, kind="stable")
        off = np.zeros(P + 1, np.uint64)
        off[1:] = np.cumsum(np.bincount(assign, minlength=P))
        self.set_trained_state(cent, cb, off, order.astype(np.uint32), codes[order])
        return dict(centroids=cent, codebook=cb, list_off=off, ids=order.astype(np.uint32), codes=codes[order], assign=assign)

    def insert(self, vector_id, vector):
        """SpannIndex::insert (spann.rs:1006-1051): nearest centroid, PQ-encode, append to that posting list"""
        v = _as_rows(vector, self._hd.dim)
        assign, codes = self.encode(v)                    # kept on the host side too, so that save_to_file sees the insert
        L.check(L.lib().shodh_index_ivfpq_insert(self.handle, int(vector_id), v.ctypes.data))
        self._pending.append((int(assign[0]), int(vector_id), codes[0].copy()))

    def _current_state(self):
        """trained state + the inserts since, as CSR (an insert goes to the END of its posting list)"""
        if self._state is None:
            raise L.ShodhError(L.ERR_STATE, "Index not built")
        st = self._state
        if not self._pending:
            return st
        P = st["centroids"].shape[0]
        part = np.concatenate([np.repeat(np.arange(P), np.diff(st["list_off"]).astype(np.int64)), np.array([p[0] for p in self._pending])])
        order = np.argsort(part, kind="stable")
        ids = np.concatenate([st["ids"], np.array([p[1] for p in self._pending], np.uint32)])[order]
        codes = np.concatenate([st["codes"], np.stack([p[2] for p in self._pending])])[order]
        off = np.zeros(P + 1, np.uint64)
        off[1:] = np.cumsum(np.bincount(part, minlength=P))
        self._state = dict(centroids=st["centroids"], codebook=st["codebook"], list_off=off, ids=np.ascontiguousarray(ids), codes=np.ascontiguousarray(codes))
        self._pending = []
        return self._state

    def save_to_file(self, path):
        """SpannIndex::save_to_file (spann.rs:750-876): SPAN v1"""
        from . import persist
        st = self._current_state()
        persist.write_spann(path, self.len(), st["centroids"], st["codebook"], st["list_off"], st["ids"], st["codes"], metric=self._metric)

    @classmethod
    def load_from_file(cls, path, num_probes=10, device=0):
        """SpannIndex::load_from_file (spann.rs:879-1003)"""
        from . import persist
        return persist.load_spann(path, num_probes=num_probes, device=device)

    @staticmethod
    def verify_index_file(path):
        """SpannIndex::verify_index_file: magic, version and checksum"""
        from . import persist
        try:
            persist.span_info(path)
            return True
        except L.ShodhError:
            return False

    def encode(self, vectors):
        v = _as_rows(vectors, self._hd.dim)
        assign = np.zeros(v.shape[0], np.uint32)
        codes = np.zeros((v.shape[0], self._hd.dim // 8), np.uint8)
        L.check(L.lib().shodh_index_ivfpq_encode(self.handle, v.ctypes.data, v.shape[0], assign.ctypes.data, codes.ctypes.data))
        return assign, codes

    def search(self, query, k):
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if q.size != self._hd.dim:
            raise L.ShodhError(L.ERR_DIM, "Query dimension %d doesn't match index dimension %d" % (q.size, self._hd.dim))
        ids, dist, counts = self.search_batch(q.reshape(1, -1), k)
        return [(int(ids[0, i]), float(dist[0, i])) for i in range(int(counts[0]))]

    def search_batch(self, queries, k):
        q = _as_rows(queries, self._hd.dim)
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        L.check(L.lib().shodh_index_search(self.handle, q.ctypes.data, nq, k, ids.ctypes.data, dist.ctypes.data, counts.ctypes.data))
        return ids[:, :k], dist[:, :k], counts


    # (SpannIndex) device-pointer search: torch CUDA tensors in and out, asynchronous on the current stream
    def search_batch_device(self, queries, k, out=None, stream=None):
        import torch
        _check_device_rows(queries, self._hd.dim)
        nq = queries.shape[0]
        if out is None:
            out = (torch.empty((nq, k), dtype=torch.int32, device=queries.device), torch.empty((nq, k), dtype=torch.float32, device=queries.device),
                   torch.empty((nq,), dtype=torch.int32, device=queries.device))
        ids, dist, counts = out
        st = stream if stream is not None else torch.cuda.current_stream(queries.device).cuda_stream
        L.check(L.lib().shodh_index_search_device(self.handle, queries.data_ptr(), nq, k, ids.data_ptr(), dist.data_ptr(),
                                                  counts.data_ptr(), C.c_void_p(st)))
        return ids, dist, counts


    def set_coalesce(self, enabled, linger_us=30):
        """concurrent host-pointer searches of a few queries share one pass (shodh_index_set_coalesce; on by default)"""
        L.check(L.lib().shodh_index_set_coalesce(self.handle, int(bool(enabled)), int(linger_us)))

    def coalesce_stats(self, reset=False):
        a = (C.c_uint64 * 6)()
        L.check(L.lib().shodh_index_coalesce_stats(self.handle, C.byref(a), int(bool(reset))))
        return dict(passes=int(a[0]), calls=int(a[1]), largest=int(a[2]), lingered=int(a[3]), pass_us=int(a[4]), linger_us=int(a[5]))


class VectorIndexBackend:
    """vector_db/mod.rs:98-266 facade."""

    def __init__(self, inner):
        self.inner = inner

    @classmethod
    def auto(cls, config: BackendConfig, expected_vectors: int):
        bt = config.force_backend or (BackendType.Spann if expected_vectors >= SPANN_AUTO_THRESHOLD else BackendType.Vamana)
        return cls.new_vamana(config) if bt == BackendType.Vamana else cls.new_spann(config)

    @classmethod
    def from_env(cls, config: BackendConfig, expected_vectors: int):
        """`auto`, unless SHODH_HIP_INDEX=flat|ivfpq forces one backend"""
        forced = os.environ.get("SHODH_HIP_INDEX", "").strip().lower()
        if forced == "flat":
            return cls.new_vamana(config)
        if forced == "ivfpq":
            return cls.new_spann(config)
        return cls.auto(config, expected_vectors)

    @classmethod
    def new_vamana(cls, config: BackendConfig):
        return cls(VamanaIndex(VamanaConfig(dimension=config.dimension, max_degree=config.vamana_max_degree,
                                            search_list_size=config.vamana_search_list_size,
                                            distance_metric=config.distance_metric, device=config.device)))

    @classmethod
    def new_spann(cls, config: BackendConfig):
        if not config.use_pq:
            raise L.ShodhError(L.ERR_STATE, "SPANN requires PQ (use_pq=true): posting lists store only PQ codes")
        return cls(SpannIndex(config.dimension, config.spann_probes, config.distance_metric, config.device))

    def backend_type(self):
        return BackendType.Vamana if isinstance(self.inner, VamanaIndex) else BackendType.Spann

    def add_vector(self, vector):
        if isinstance(self.inner, VamanaIndex):
            return self.inner.add_vector(vector)
        vid = self.inner.len()
        self.inner.insert(vid, vector)
        return vid

    def search(self, query, k):
        return self.inner.search(query, k)

    def len(self):
        return self.inner.len()

    def is_empty(self):
        return self.len() == 0

    def save_to_file(self, path):                                   # vector_db/mod.rs:188-193
        return self.inner.save_to_file(path)

    @classmethod
    def load_from_file(cls, path, backend_type, device=0):          # :196-201
        return cls(VamanaIndex.load_from_file(path, device=device) if backend_type == BackendType.Vamana
                   else SpannIndex.load_from_file(path, device=device))

    @staticmethod
    def verify_index_file(path, backend_type):                      # :260-265
        return VamanaIndex.verify_index_file(path) if backend_type == BackendType.Vamana else SpannIndex.verify_index_file(path)

    def build(self, vectors):
        return self.inner.build(vectors)

    def needs_rebuild(self):
        return self.inner.needs_rebuild() if isinstance(self.inner, VamanaIndex) else False

    def auto_rebuild_if_needed(self):
        return self.inner.auto_rebuild_if_needed() if isinstance(self.inner, VamanaIndex) else False

    def incremental_insert_count(self):
        return self.inner.incremental_insert_count() if isinstance(self.inner, VamanaIndex) else 0

    def deleted_count(self):
        return self.inner.deleted_count() if isinstance(self.inner, VamanaIndex) else 0

    def deletion_ratio(self):
        return self.inner.deletion_ratio() if isinstance(self.inner, VamanaIndex) else 0.0

    def needs_compaction(self):
        return self.inner.needs_compaction() if isinstance(self.inner, VamanaIndex) else False

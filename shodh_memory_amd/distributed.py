"""Row-sharded corpus across the GPUs of one node (new design -- the reference is
single-process).

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm). Rank r holds the
contiguous global id range [base_r, base_r + n_r): `global_id = id_base + local_row`, so ids
stay dense and sequential as add_vector requires (vamana.rs:854-855). A search is
  local exact top-k on every shard  ->  all_gather of the (ids, dist) blocks [nq,k]  ->
  merge by (dist total_cmp, id)  (shodh_topk_merge_device)
Every local distance was produced in the reference's accumulation order, so the merged list is
bit-identical to a single-device search over the concatenated corpus. The payload is tiny
(nq*k*8 bytes per rank): the collective is latency-bound, not bandwidth-bound.

torch is used for process-group plumbing and device buffers only.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .index import VamanaConfig, VamanaIndex


def shard_range(n_total, world, rank):
    """contiguous row range of `rank`: ceil(n/world) rows per rank, last ranks may be short/empty"""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def merge_gathered_numpy(ids_all, dist_all, k):
    """CPU form of shodh_topk_merge_device (ShardedFlatIndex.merge_cpu): ids_all/dist_all [world, nq, k]."""
    world, nq, kk = ids_all.shape
    out_i = np.full((nq, k), 0xFFFFFFFF, np.uint32)
    out_d = np.full((nq, k), np.inf, np.float32)
    counts = np.zeros(nq, np.uint32)
    bits = dist_all.view(np.uint32)
    key32 = np.where(bits & 0x80000000, ~bits, bits | 0x80000000).astype(np.uint64)
    keys = (key32 << np.uint64(32)) | ids_all.astype(np.uint64)
    for q in range(nq):
        kq = keys[:, q, :].reshape(-1)
        valid = ids_all[:, q, :].reshape(-1) != 0xFFFFFFFF
        kq = np.sort(kq[valid])[:k]
        m = len(kq)
        counts[q] = m
        out_i[q, :m] = (kq & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        k32 = (kq >> np.uint64(32)).astype(np.uint32)
        b = np.where(k32 & 0x80000000, k32 ^ 0x80000000, ~k32).astype(np.uint32)
        out_d[q, :m] = b.view(np.float32)
    return out_i, out_d, counts


class ShardedFlatIndex:
    """One shard of a row-sharded flat index + the collective search.

    `index_factory(dim, order, scan_mode, device, reserve_rows, id_base)` builds the per-shard index (default: the library's VamanaIndex on this rank's
    GPU) and `merge(pack_all, world, nq, k, out_ids, out_dist, out_counts)` folds the gathered blocks (default: shodh_topk_merge_strided_device). The
    world-size 2 / 3 / 8 `gloo` tests (tests/test_distributed_cpu.py) inject a CPU per-shard search and the CPU merge below and run THIS class -- its
    shard ranges, its buffers, its one all-gather per search and its merge call -- so that the first run on eight GPUs needs no code that has not run."""

    def __init__(self, dim=384, n_total=0, order=L.ORDER_SCALAR4, scan_mode=L.SCAN_AUTO, device=None, group=None, index_factory=None, merge=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if device is None and index_factory is None:
            device = torch.cuda.current_device()
        self.device = device
        self.lo, self.hi = shard_range(n_total, self.world, self.rank)
        if index_factory is None:
            index_factory = lambda **kw: VamanaIndex(VamanaConfig(dimension=kw["dim"], order=kw["order"], scan_mode=kw["scan_mode"], device=kw["device"],     # noqa: E731
                                                                  reserve_rows=kw["reserve_rows"], id_base=kw["id_base"]))
        # an empty shard (more ranks than rows, or a short tail) still takes part in every collective: it answers with empty lists
        self.index = index_factory(dim=dim, order=order, scan_mode=scan_mode, device=self.device, reserve_rows=max(self.hi - self.lo, 1), id_base=self.lo)
        self._merge = merge or self._merge_device
        self._bufs = {}

    def build_local(self, rows):
        """rows: this rank's [hi-lo, dim] slice (numpy or torch CUDA tensor)."""
        assert rows.shape[0] == self.hi - self.lo
        if self.hi > self.lo:
            self.index.build(rows)

    @staticmethod
    def _merge_device(pack_all, world, nq, k, out_ids, out_dist, out_counts):
        import torch
        st = torch.cuda.current_stream(pack_all.device).cuda_stream
        base = pack_all.data_ptr()
        L.check(L.lib().shodh_topk_merge_strided_device(base, base + nq * k * 4, 2 * nq * k, world, nq, k,
                                                        out_ids.data_ptr(), out_dist.data_ptr(), out_counts.data_ptr(), C.c_void_p(st)))

    @staticmethod
    def merge_cpu(pack_all, world, nq, k, out_ids, out_dist, out_counts):
        """the same fold on CPU tensors (merge_gathered_numpy): what the gloo tests inject"""
        import torch
        g = pack_all.numpy()
        m_ids, m_dd, counts = merge_gathered_numpy(np.ascontiguousarray(g[:, 0]).view(np.uint32), np.ascontiguousarray(g[:, 1]).view(np.float32), k)
        out_ids.copy_(torch.from_numpy(m_ids.view(np.int32)))
        out_dist.copy_(torch.from_numpy(m_dd))
        out_counts.copy_(torch.from_numpy(counts.view(np.int32)))

    def search_batch_device(self, queries, k):
        """queries: torch tensor [nq, dim] (CUDA in production), identical on every rank. Returns merged (ids, dist, counts)."""
        import torch
        nq = queries.shape[0]
        key = (nq, k)
        if key not in self._bufs:
            dev = queries.device
            # one packed [ids | dist] block per rank -> ONE all-gather per search (the payload is tiny, the collective is
            # latency-bound: two collectives cost twice as much as one)
            pack = torch.empty((2, nq, k), dtype=torch.int32, device=dev)
            self._bufs[key] = dict(
                pack=pack, ids=pack[0], dist=pack[1].view(torch.float32), counts=torch.empty((nq,), dtype=torch.int32, device=dev),
                pack_all=torch.empty((self.world, 2, nq, k), dtype=torch.int32, device=dev),
                out_ids=torch.empty((nq, k), dtype=torch.int32, device=dev), out_dist=torch.empty((nq, k), dtype=torch.float32, device=dev),
                out_counts=torch.empty((nq,), dtype=torch.int32, device=dev))
        b = self._bufs[key]
        if self.hi > self.lo:
            self.index.search_batch_device(queries, k, out=(b["ids"], b["dist"], b["counts"]))
        else:           # an empty shard: the "no entry" markers the merge skips (ids 0xFFFFFFFF, distance +inf)
            b["ids"].fill_(-1)
            b["dist"].fill_(float("inf"))
            b["counts"].zero_()
        if self.world == 1:
            return b["ids"], b["dist"], b["counts"]
        # concatenation form [world*2*nq, k] (accepted by both RCCL and gloo); same memory as [world, 2, nq, k]
        self.dist.all_gather_into_tensor(b["pack_all"].view(self.world * 2 * nq, k), b["pack"].view(2 * nq, k), group=self.group)
        self._merge(b["pack_all"], self.world, nq, k, b["out_ids"], b["out_dist"], b["out_counts"])
        return b["out_ids"], b["out_dist"], b["out_counts"]


class ShardedEmbedder:
    """Data-parallel encoder over several handles ("Encoder: pure data-parallel over texts, no collective"): one MiniLMEmbedder per device
    (or several on one device: the single-GPU stand-in of the tests), texts dealt to the handles in contiguous, near-equal shares, every share encoded on
    its own host thread, vectors returned in input order.

    Only the calls whose result does not depend on batch mates are offered: `encode_each` / `encode_ids(scope=PER_TEXT)` (N x encode(): what `remember`
    and `recall` compute, minilm.rs:883-982) and, for fp32 / bf16 handles, `encode_batch` (texts never interact there). An INT8 `encode_batch` under
    SHODH_QUANT_SCOPE_BATCH is ONE function of the whole batch (the activation ranges span it, minilm.rs:588-593): it is not split, it runs on handle 0."""

    def __init__(self, embedders):
        assert embedders, "at least one handle"
        self.embedders = list(embedders)

    def dimension(self):
        return self.embedders[0].dimension()

    def close(self):
        for e in self.embedders:
            e.close()

    def _shares(self, n):
        g = min(len(self.embedders), max(n, 1))
        per = (n + g - 1) // g
        return [(i * per, min((i + 1) * per, n)) for i in range(g) if i * per < n]

    def _run(self, n, call):
        """call(handle, lo, hi) -> array [hi-lo, dim]; shares run concurrently (the library calls release the GIL), results in input order"""
        import threading
        shares = self._shares(n)
        out, err = [None] * len(shares), [None] * len(shares)

        def work(i, lo, hi):
            try:
                out[i] = call(self.embedders[i], lo, hi)
            except Exception as ex:      # noqa: BLE001
                err[i] = ex
        th = [threading.Thread(target=work, args=(i, lo, hi)) for i, (lo, hi) in enumerate(shares[1:], 1)]
        for t in th:
            t.start()
        if shares:
            work(0, *shares[0])
        for t in th:
            t.join()
        for ex in err:
            if ex is not None:
                raise ex
        return out

    def encode_ids(self, ids, mask, scope=L.QUANT_SCOPE_PER_TEXT):
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, self.embedders[0].max_length)
        mask = np.ascontiguousarray(mask, np.uint8).reshape(ids.shape)
        if scope != L.QUANT_SCOPE_PER_TEXT and any(getattr(e, "_cfg", None) is not None and e._cfg.dtype == L.DTYPE_INT8 for e in self.embedders):
            return self.embedders[0].encode_ids(ids, mask, scope=scope)          # one function of the whole batch: not split
        parts = self._run(ids.shape[0], lambda e, lo, hi: e.encode_ids(ids[lo:hi], mask[lo:hi], scope=scope))
        return np.concatenate(parts, axis=0) if parts else np.zeros((0, self.dimension()), np.float32)

    def encode_each(self, texts):
        texts = list(texts)
        parts = self._run(len(texts), lambda e, lo, hi: np.asarray(e.encode_each(texts[lo:hi]), np.float32).reshape(hi - lo, -1))
        return [row for p in parts for row in p]

    def encode(self, text):
        return self.embedders[0].encode(text)

    def encode_query(self, text):
        return self.embedders[0].encode_query(text)


class MultiGpuIndex:
    """ONE index over several GPUs of a node from ONE process, entirely behind the C ABI (`shodh_sharded_index_*`,
    csrc/sharded.hip): per-device shards, RCCL all-gather of the per-shard top-k, device merge. The VamanaIndex / SpannIndex
    method subset a caller needs (`build`, `add_vector(s)`, `search`, `mark_deleted`, `len`, `extract_all_vectors`,
    `set_trained_state`, `insert`). `devices` may repeat an ordinal (several shards on one GPU: the exchange then uses device
    copies instead of RCCL -- the single-GPU stand-in used by the tests)."""

    def __init__(self, devices, dim=384, kind=L.INDEX_FLAT, order=L.ORDER_SCALAR4, scan_mode=L.SCAN_AUTO, nprobe=20, block_log2=16,
                 exchange=L.EXCHANGE_AUTO, reserve_rows_per_shard=0):
        cfg = L.ShardedCfg()
        L.lib().shodh_sharded_cfg_default(C.byref(cfg))
        cfg.dim, cfg.kind, cfg.order, cfg.scan_mode, cfg.nprobe = dim, kind, order, scan_mode, nprobe
        cfg.block_log2, cfg.exchange, cfg.reserve_rows_per_shard = block_log2, exchange, reserve_rows_per_shard
        dv = np.ascontiguousarray(devices, np.int32)
        self._h = C.c_void_p()
        L.check(L.lib().shodh_sharded_index_create(C.byref(cfg), dv.ctypes.data, dv.size, C.byref(self._h)))
        self.dim, self.kind = dim, kind

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib().shodh_sharded_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shards(self):
        return int(L.lib().shodh_sharded_index_shards(self._h))

    def uses_rccl(self):
        return bool(L.lib().shodh_sharded_index_uses_rccl(self._h))

    def len(self):
        return int(L.lib().shodh_sharded_index_len(self._h))

    def shard_len(self, g):
        return int(L.lib().shodh_sharded_index_shard_len(self._h, g))

    def _rows(self, vectors):
        a = np.ascontiguousarray(vectors, np.float32)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        if a.size and a.shape[1] != self.dim:
            raise L.ShodhError(L.ERR_DIM, "Vector dimension %d doesn't match config %d" % (a.shape[1], self.dim))
        return a.reshape(-1, self.dim)

    def build(self, vectors):
        a = self._rows(vectors)
        L.check(L.lib().shodh_sharded_index_build(self._h, a.ctypes.data, a.shape[0]))

    def add_vectors(self, vectors):
        a = self._rows(vectors)
        first = C.c_uint32()
        L.check(L.lib().shodh_sharded_index_add(self._h, a.ctypes.data, a.shape[0], C.byref(first)))
        return int(first.value)

    def add_vector(self, vector):
        return self.add_vectors(vector)

    def search_batch(self, queries, k):
        q = self._rows(queries)
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        L.check(L.lib().shodh_sharded_index_search(self._h, q.ctypes.data, nq, k, ids.ctypes.data, dist.ctypes.data, counts.ctypes.data))
        return ids[:, :k], dist[:, :k], counts

    def search_batch_device(self, queries, k, out=None, stream=None):
        """queries: float32 CUDA tensor on the index's FIRST device; results stay there (ids int32 view of u32, dist, counts). Asynchronous on the
        current stream of that device: no host round trip (shodh_sharded_index_search_device)."""
        import torch
        nq = queries.shape[0]
        if out is None:
            out = (torch.empty((nq, max(k, 1)), dtype=torch.int32, device=queries.device), torch.empty((nq, max(k, 1)), dtype=torch.float32, device=queries.device),
                   torch.empty((nq,), dtype=torch.int32, device=queries.device))
        st = stream if stream is not None else torch.cuda.current_stream(queries.device).cuda_stream
        L.check(L.lib().shodh_sharded_index_search_device(self._h, queries.data_ptr(), nq, k, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), C.c_void_p(st)))
        return out

    def search(self, query, k):
        ids, dist, counts = self.search_batch(np.asarray(query, np.float32).reshape(1, -1), k)
        return [(int(ids[0, i]), float(dist[0, i])) for i in range(int(counts[0]))]

    def mark_deleted(self, vector_id):
        ok = C.c_int()
        L.check(L.lib().shodh_sharded_index_mark_deleted(self._h, int(vector_id), C.byref(ok)))
        return bool(ok.value)

    def mark_deleted_many(self, vector_ids):
        ids = np.ascontiguousarray(vector_ids, np.uint32)
        m = C.c_uint64()
        L.check(L.lib().shodh_sharded_index_mark_deleted_batch(self._h, ids.ctypes.data, ids.size, C.byref(m)))
        return int(m.value)

    def is_deleted(self, vector_id):
        return bool(L.lib().shodh_sharded_index_is_deleted(self._h, int(vector_id)))

    def deleted_count(self):
        return int(L.lib().shodh_sharded_index_deleted_count(self._h))

    def clear_deleted(self):
        L.check(L.lib().shodh_sharded_index_clear_deleted(self._h))

    def extract_all_vectors(self):
        n = self.len()
        out = np.empty((n, self.dim), np.float32)
        if n:
            L.check(L.lib().shodh_sharded_index_extract_rows(self._h, 0, n, out.ctypes.data))
        return out

    def set_trained_state(self, centroids, codebook, list_off, ids, codes):
        c = np.ascontiguousarray(centroids, np.float32)
        cb = np.ascontiguousarray(codebook, np.float32)
        lo = np.ascontiguousarray(list_off, np.uint64)
        i = np.ascontiguousarray(ids, np.uint32)
        cd = np.ascontiguousarray(codes, np.uint8)
        L.check(L.lib().shodh_sharded_index_set_ivfpq(self._h, c.ctypes.data, c.shape[0], cb.ctypes.data, cb.shape[0], cb.shape[1],
                                                      lo.ctypes.data, i.ctypes.data if i.size else None, cd.ctypes.data if cd.size else None))

    def insert(self, vector_id, vector):
        v = self._rows(vector)
        L.check(L.lib().shodh_sharded_index_ivfpq_insert(self._h, int(vector_id), v.ctypes.data))

    def set_coalesce(self, enabled, linger_us=30):
        L.check(L.lib().shodh_sharded_index_set_coalesce(self._h, int(bool(enabled)), int(linger_us)))

    def coalesce_stats(self, reset=False):
        a = (C.c_uint64 * 6)()
        L.check(L.lib().shodh_sharded_index_coalesce_stats(self._h, C.byref(a), int(bool(reset))))
        return dict(passes=int(a[0]), calls=int(a[1]), largest=int(a[2]), lingered=int(a[3]), pass_us=int(a[4]), linger_us=int(a[5]))

    def host_timings_us(self):
        a = (C.c_float * 4)()
        L.check(L.lib().shodh_sharded_index_host_timings(self._h, C.byref(a)))
        return dict(enqueue=a[0], exchange=a[1], merge_wait=a[2], total=a[3])


def rccl_info():
    buf = C.create_string_buffer(512)
    rc = L.lib().shodh_rccl_info(buf, 512)
    return rc == L.OK, buf.value.decode()

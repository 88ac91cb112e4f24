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
    """CPU mirror of shodh_topk_merge_device for the gloo tests: ids_all/dist_all [world, nq, k]."""
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
    """One shard of a row-sharded flat index + the collective search."""

    def __init__(self, dim=384, n_total=0, order=L.ORDER_SCALAR4, scan_mode=L.SCAN_AUTO, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.cuda.current_device() if device is None else device
        self.lo, self.hi = shard_range(n_total, self.world, self.rank)
        self.index = VamanaIndex(VamanaConfig(dimension=dim, order=order, scan_mode=scan_mode, device=self.device,
                                              reserve_rows=max(self.hi - self.lo, 1), id_base=self.lo))
        self._bufs = {}

    def build_local(self, rows):
        """rows: this rank's [hi-lo, dim] slice (numpy or torch CUDA tensor)."""
        assert rows.shape[0] == self.hi - self.lo
        self.index.build(rows)

    def search_batch_device(self, queries, k):
        """queries: torch CUDA [nq, dim], identical on every rank. Returns merged (ids, dist, counts)."""
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
        self.index.search_batch_device(queries, k, out=(b["ids"], b["dist"], b["counts"]))
        if self.world == 1:
            return b["ids"], b["dist"], b["counts"]
        # concatenation form [world*2*nq, k] (accepted by both RCCL and gloo); same memory as [world, 2, nq, k]
        self.dist.all_gather_into_tensor(b["pack_all"].view(self.world * 2 * nq, k), b["pack"].view(2 * nq, k), group=self.group)
        st = torch.cuda.current_stream(queries.device).cuda_stream
        base = b["pack_all"].data_ptr()
        L.check(L.lib().shodh_topk_merge_strided_device(base, base + nq * k * 4, 2 * nq * k, self.world, nq, k,
                                                        b["out_ids"].data_ptr(), b["out_dist"].data_ptr(), b["out_counts"].data_ptr(),
                                                        C.c_void_p(st)))
        return b["out_ids"], b["out_dist"], b["out_counts"]

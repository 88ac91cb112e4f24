"""`gloo` tests of the row-sharded search on CPU, world sizes 2, 3 and 8 (the target): the REAL `ShardedFlatIndex` object (shodh_memory_amd/distributed.py)
-- its shard ranges, buffers, the one all-gather per search and the merge call -- with the per-shard search injected: a CPU stand-in that answers with the
oracle's exact lists (no GPU in this container; on the GPU box the per-shard search is the library's VamanaIndex and the merge the device kernel, both
covered by the GPU suite). Short and EMPTY last shards, duplicate rows straddling every shard boundary, k larger than a shard: the merged result must
equal a single exact search over the whole corpus, ids and distance bytes."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


class CpuShard:
    """the VamanaIndex surface ShardedFlatIndex uses (build, search_batch_device with out=), answered by the oracle; ids = id_base + local row"""

    def __init__(self, O, id_base):
        self.O, self.id_base, self.rows = O, id_base, None

    def build(self, rows):
        self.rows = np.ascontiguousarray(rows, np.float32)

    def search_batch_device(self, queries, k, out):
        import torch
        q = queries.numpy()
        ids, dd = self.O.brute_force_batch(self.rows, q, k)                 # [nq, k]; a shard shorter than k fills min(k, rows) entries
        m = min(k, self.rows.shape[0])
        valid = np.zeros(ids.shape, bool)
        valid[:, :m] = True
        ids = np.where(valid, ids + np.uint32(self.id_base), np.uint32(0xFFFFFFFF)).astype(np.uint32)     # "no entry" as the library marks it
        dd = np.where(valid, dd, np.float32(np.inf)).astype(np.float32)
        out[0].copy_(torch.from_numpy(ids.view(np.int32).copy()))
        out[1].copy_(torch.from_numpy(dd.copy()))
        out[2].copy_(torch.from_numpy(valid.sum(axis=1).astype(np.int32)))
        return out


def _worker(rank, world, port, tmpdir, n, k):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from shodh_memory_amd.distributed import ShardedFlatIndex, shard_range
    from tests import synth
    nq = 9
    q = synth.queries(nq)
    rows = synth.corpus(n, queries=q)
    for r in range(1, world):                 # duplicate rows straddling EVERY shard boundary: the id tie-break must survive the merge
        b = shard_range(n, world, r)[0]
        if 1 <= b < n - 1:
            rows[b - 1] = rows[0]; rows[b] = rows[0]; rows[b + 1] = rows[0]
    sh = ShardedFlatIndex(dim=384, n_total=n, index_factory=lambda **kw: CpuShard(O, kw["id_base"]), merge=ShardedFlatIndex.merge_cpu)
    assert (sh.lo, sh.hi) == shard_range(n, world, rank) and sh.world == world and sh.rank == rank
    sh.build_local(rows[sh.lo:sh.hi])
    tq = torch.from_numpy(q)
    ok = True
    for it in range(2):                       # (twice: the second search reuses the object's buffers)
        m_ids, m_dd, m_cnt = sh.search_batch_device(tq, k)
        e_ids, e_dd = O.brute_force_batch(rows, q, k)
        want = min(k, n)
        ok = ok and bool(np.array_equal(m_ids.numpy().view(np.uint32), e_ids) and m_dd.numpy().tobytes() == e_dd.tobytes() and (m_cnt.numpy() == want).all())
    open(os.path.join(tmpdir, "ok%d" % rank), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range():
    sys.path.insert(0, ROOT)
    from shodh_memory_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000, 1_000_000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, i) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= (n + w - 1) // w


@pytest.mark.parametrize("world,n,k", [(2, 5003, 10),        # two even shards
                                        (3, 1000, 120),      # recall's own k; 334 + 334 + 332
                                        (8, 83, 10),         # the target world size: shards of 11 rows, the last one short (6)
                                        (8, 17, 10),         # ... 3 rows per shard, the last TWO shards empty, k larger than any shard
                                        (8, 40003, 120)])
def test_gloo_sharded_search_runs_the_real_object(tmp_path, world, n, k):
    import torch.multiprocessing as mp
    from shodh_memory_amd import build
    build.build()
    from oracle import oracle as O
    O.build()
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), n, k), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        assert open(tmp_path / ("ok%d" % r)).read() == "1", "rank %d of %d" % (r, world)

"""world_size-2 `gloo` test of the row-sharded search (shodh_memory_amd/distributed.py) on CPU: shard
ranges, the all-gather of per-shard (ids, dist) blocks and the (dist total_cmp, id) merge. The per-shard
search is played by the oracle here (no GPU in this container); the collective and the merge are the code
under test, and the merged result must equal a single-index search over the whole corpus."""
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


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from shodh_memory_amd.distributed import merge_gathered_numpy, shard_range
    from tests import synth
    n, nq, k = 5003, 9, 10
    q = synth.queries(nq)
    rows = synth.corpus(n, queries=q)
    rows[4000:4003] = rows[10]            # duplicates straddling the shard boundary: id tie-break must survive the merge
    lo, hi = shard_range(n, world, rank)
    ids, dd = O.brute_force_batch(rows[lo:hi], q, k)                     # this rank's shard, local ids
    ids = (ids + np.uint32(lo)).astype(np.uint32)                         # id_base
    # one packed [ids | dist] block per rank and ONE all-gather, the call shape of ShardedFlatIndex.search_batch_device
    pack = torch.empty((2, nq, k), dtype=torch.int32)
    pack[0] = torch.from_numpy(ids.view(np.int32).copy())
    pack[1] = torch.from_numpy(dd.view(np.int32).copy())
    pack_all = torch.empty((world, 2, nq, k), dtype=torch.int32)
    dist.all_gather_into_tensor(pack_all.view(world * 2 * nq, k), pack.view(2 * nq, k))
    g = pack_all.numpy()
    all_ids = np.ascontiguousarray(g[:, 0]).view(np.uint32)
    all_dd = np.ascontiguousarray(g[:, 1]).view(np.float32)
    m_ids, m_dd, counts = merge_gathered_numpy(all_ids, all_dd, k)
    e_ids, e_dd = O.brute_force_batch(rows, q, k)
    ok = bool(np.array_equal(m_ids, e_ids) and m_dd.tobytes() == e_dd.tobytes() and (counts == k).all())
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


def test_two_rank_gloo_sharded_search(tmp_path):
    import torch.multiprocessing as mp
    from shodh_memory_amd import build
    build.build()
    from oracle import oracle as O
    O.build()
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    assert open(tmp_path / "ok0").read() == "1" and open(tmp_path / "ok1").read() == "1"

"""The N > 1 branch of bench.py, exercised on ONE GPU: two ranks launched the way the driver launches them
(torch.distributed.run, one process per rank), gloo instead of RCCL for the two collectives (SHODH_BENCH_BACKEND=gloo; both ranks share
the one device), a small corpus. Not a measurement -- a check that the branch runs and that the one JSON line it prints carries the contract's
fields for N > 1, so that the first real SCALE run produces a curve."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_on_one_gpu_print_one_contract_line():
    env = dict(os.environ, SHODH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000", "--steps", "3", "--warmup", "1", "--prewarm-ms", "5", "--no-extra-configs",
           "--no-cpu-baseline", "--no-latency", "--sustained-s", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line; nothing else reaches stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "queries/s" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    batch = d["config"]["batch"]
    assert abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3     # ANSWERED queries per second over the whole (2 x 200k) corpus, not queries x shards
    assert "configs[4]" in d["config"]["workload"] or "sharded" in d["config"]["workload"], d["config"]["workload"]
    assert d["config"]["rows_total"] == 400000 and d["config"]["parallelism"].startswith("row-shard")
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and 0 < rf["frac"] and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d.get("gpu_matches_cpu_bit_exact") in (None, True)

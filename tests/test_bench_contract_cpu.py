"""The driver's contract for `bench.py`, checked without a GPU: the command line it is launched with parses, the defaults are the N = 1 /
minutes-long run the contract asks for, and the committed line of the latest round's last run (`profiles/rN_bench_line.json`) carries every
field the driver and the judge read, with figures that agree with one another (value = queries / step time, roofline.frac = achieved /
peak = algorithmic bytes / kernel time / peak, kernel time <= step time)."""
import ast
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_arguments():
    """{flag: default} of bench.py's argparse calls, read from the source (importing bench.py would load the GPU library)"""
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args and isinstance(node.args[0], ast.Constant):
            kw = {k.arg: k.value for k in node.keywords}
            default = kw.get("default")
            out[node.args[0].value] = ast.literal_eval(default) if default is not None else ("flag" if "action" in kw else None)
    return out


def test_command_line_of_the_contract():
    args = _bench_arguments()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in args, flag
    assert args["--gpus"] == 1                                   # no flags: one GPU ...
    assert 1 <= args["--steps"] <= 1000 and 0 <= args["--warmup"] <= 100     # ... and a K / W that finish within minutes
    assert args["--cpu-seconds"] <= 30.0                          # the CPU baseline leg is a bounded sample (10 - 30 s)


def test_committed_line_is_a_contract_line():
    import glob
    import re
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")), key=lambda p: int(re.search(r"r(\d+)_bench_line", p).group(1)))
    if not lines:
        pytest.skip("no committed bench line")
    path = lines[-1]                                              # the latest round's
    rnd = int(re.search(r"r(\d+)_bench_line", path).group(1))
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "queries/s" and "workload" in d["config"] and "model" not in d["config"]
    batch = d["config"]["batch"]
    assert abs(d["value"] - batch / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3          # whole-job throughput = answered queries / step time
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    kern_s = r["launch_us_mean"] * 1e-6
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / kern_s / 1e9) / r["achieved"] < 2e-3
    assert r["algorithmic_bytes_per_launch"] == r["rows_live"] * 384 * 4                       # live rows x D x 4, once per batch
    assert kern_s * 1e3 <= d["ms_per_step"]                                                    # the kernel is part of the step
    assert r["traffic"] is None or 0.4 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.1 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "queries/s"
    lat = d.get("latency_single_query")
    assert lat and lat["nq"] == 1 and 0 < lat["p50_ms"] <= lat["p95_ms"]
    names = [cfg["name"] for cfg in d.get("configs", [])]
    for want in ("cfg1_10k_b1", "flat_10M_b256", "cfg4_ivfpq_10M", "cfg3_pipeline", "encoder_bf16_b8192", "encoder_int8_b4096", "sharded_c_abi_1M_b256", "sharded_c_abi_10M_b256") \
            + (("cfg3_pipeline_int8", "encoder_int8_pertext_b4096") if rnd >= 4 else ()):      # round 4: INT8 per text (N x encode(), what remember / recall run)
        assert want in names, want
    pipe = [c for c in d["configs"] if c["name"] == "cfg3_pipeline"][0]
    assert "1000000 synthetic texts" in pipe["workload"]                                       # configs[2] at its stated size

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


def _latest_line():
    import glob
    import re
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")), key=lambda p: int(re.search(r"r(\d+)_bench_line", p).group(1)))
    if not lines:
        pytest.skip("no committed bench line")
    path = lines[-1]                                              # the latest round's
    rnd = int(re.search(r"r(\d+)_bench_line", path).group(1))
    raw = open(path).read().strip().splitlines()[-1]
    return rnd, path, raw, json.loads(raw)


def _check_contract_fields(d):
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


_WANTED = ("cfg1_10k_b1", "flat_10M_b256", "cfg4_ivfpq_10M", "cfg3_pipeline", "encoder_bf16_b8192", "encoder_int8_b4096", "sharded_c_abi_1M_b256",
           "sharded_c_abi_10M_b256")


def test_committed_line_is_a_contract_line():
    rnd, path, raw, d = _latest_line()
    _check_contract_fields(d)
    if rnd >= 6:
        # round 6 on: the committed line is what the driver parses -- the BOUNDED line (BENCH_r05.json: parsed = null on a 27 KB line)
        assert len(raw) <= 4096, "the contract line grew to %d bytes" % len(raw)
        assert max(len(v) for v in _strings(d)) <= 120
        detail = path.replace("_bench_line.json", "_bench_detail.json")
        assert os.path.exists(detail), "the full record of the same run is committed beside the line"
        full = json.loads(open(detail).read().strip().splitlines()[-1])
        assert full["value"] == d["value"] and full["ms_per_step"] == d["ms_per_step"]        # one run, two renderings
    else:
        full = d
    names = [cfg["name"] for cfg in d.get("configs", [])]
    for want in _WANTED + (("cfg3_pipeline_int8", "encoder_int8_pertext_b4096") if rnd >= 4 else ()):      # round 4: INT8 per text (N x encode(), what remember / recall run)
        assert want in names, want
    pipe = [c for c in full["configs"] if c["name"] == "cfg3_pipeline"][0]
    assert "1000000 synthetic texts" in pipe["workload"]                                       # configs[2] at its stated size


def _strings(x):
    if isinstance(x, str):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _strings(v)
    elif isinstance(x, list):
        for v in x:
            yield from _strings(v)


def _bench_module():
    """bench.py imports only the standard library at module level (torch and the GPU library are imported inside main())"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench_for_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contract_line_is_bounded_whatever_the_record_holds():
    """compact_line() of the largest record ever produced (round 5: 27 KB, the one the driver could not parse) is a contract line under 4 KB;
    a record with many more and much longer entries still is (entries are dropped from the end, with a count, never the contract fields)."""
    b = _bench_module()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r5_bench_line.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    line = b.compact_line(full, "gpurun_out/bench_detail.json")
    raw = json.dumps(line)
    assert len(raw) <= b.CONTRACT_LINE_MAX_BYTES <= 4096 and len(raw) < 8192
    _check_contract_fields(line)
    assert [c["name"] for c in line["configs"]] == [c["name"] for c in full["configs"]]        # nothing dropped at today's size
    assert "configs_dropped_for_size" not in line
    k120 = [c for c in line["configs"] if c["name"] == "flat_1M_b256_k120"][0]
    assert k120["ms_per_step"] == 0.3776 and k120["frac"] == 0.483
    assert max(len(v) for v in _strings(line)) <= 120
    fat = dict(full)
    fat["configs"] = [dict(c, name="%s_%d" % (c["name"], i), workload="x" * 5000) for i in range(12) for c in full["configs"]]
    fat["config"] = dict(full["config"], workload="y" * 3000)
    fat["cpu_baseline"] = dict(full["cpu_baseline"], sample="z" * 3000)
    line = b.compact_line(fat, None)
    assert len(json.dumps(line)) <= b.CONTRACT_LINE_MAX_BYTES
    _check_contract_fields(line)
    assert line["configs_dropped_for_size"] > 0 and len(line["configs"]) > 10
    # a line of an N > 1 run (no cpu_baseline, no configs, no latency) goes through as well
    multi = {k: full[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline")}
    multi.update({"n_gpus": 8, "scaling": "weak", "cpu_baseline": None, "latency_single_query": None, "sustained": None})
    multi["config"] = dict(full["config"], shard_queries_per_s=8 * full["value"], weak_scaling_reference="per-GPU work is fixed at 10000000 rows: " + "w" * 400)
    line = b.compact_line(multi)
    assert line["n_gpus"] == 8 and line["cpu_baseline"] is None and "configs" not in line
    # ... with the work all ranks do per second next to the answered rate (round 6), its explanation shortened like every string
    assert line["config"]["shard_queries_per_s"] == 8 * full["value"] and len(line["config"]["weak_scaling_reference"]) <= 120
    assert len(json.dumps(line)) <= b.CONTRACT_LINE_MAX_BYTES
    single = b.compact_line(full, None)
    assert "shard_queries_per_s" not in single["config"] and "weak_scaling_reference" not in single["config"]      # (N = 1: not there)


def test_last_stdout_line_is_the_contract_line_whatever_else_is_printed(tmp_path):
    """What the driver does: run the command, take the LAST line of stdout, json.loads it. Native libraries print on fd 1 behind Python's back -- RCCL its
    version banner at communicator creation and a banner at exit (after the line was written) -- and Python code may print too: stdout must still be
    exactly the one line. Run in a child process with libc printf / write(1) noise before and after the line and at interpreter exit."""
    import subprocess
    import sys
    full = os.path.join(ROOT, "profiles", "r5_bench_line.json")
    code = r"""
import atexit, ctypes, importlib.util, json, os, sys
spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
libc = ctypes.CDLL(None)
fd = b.claim_stdout()
libc.puts(b'NCCL version 2.26.6+hip7.0 (a banner printed by a native library)')
os.write(1, b'a stray write(1)\n'); print('a stray print')
def bye():
    libc.puts(b'RCCL exit banner'); libc.fflush(None); os.write(1, b'{"not": "the line"}\n')
atexit.register(bye)
rec = json.loads(open(%r).read().strip().splitlines()[-1])
b.emit_contract(fd, rec, 'gpurun_out/bench_detail.json')
libc.puts(b'more noise after the line'); print('{"also": "not the line"}')
""" % (os.path.join(ROOT, "bench.py"), full)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = p.stdout.decode()
    assert out.endswith("\n") and len(out.splitlines()) == 1, out[-500:]
    last = out.strip().splitlines()[-1]
    assert len(last) <= 4096
    d = json.loads(last)
    _check_contract_fields(d)
    assert "RCCL exit banner" in p.stderr.decode() and "NCCL version" in p.stderr.decode()     # the noise went to stderr

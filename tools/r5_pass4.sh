#!/bin/bash
# round 5, pass 4: hipGraph replay of the one-text INT8 forward, clock sampling, from-source build on a fresh box, whole GPU suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5p4; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
ls shodh_memory_amd/build 2>&1 | head -3
( time python -c "import __graft_entry__ as g; g.build()" ) 2>&1 | grep "build()\|real" | tee $OUT/build.txt
timeout 900 python -m pytest tests/test_concurrent_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 | tee $OUT/gpu_tests.txt
for G in 0 1; do echo "SHODH_ENC_GRAPH=$G"; SHODH_ENC_GRAPH=$G timeout 300 python tools/enc_latency_probe.py int8 2>&1 | tail -1; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-configs concurrent_encode_callers > $OUT/line.json 2> $OUT/err.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p4/line.json").read().strip().splitlines()[-1])
print("main:", d["value"], d["ms_per_step"], d["roofline"]["gpu_sclk_mhz"], d["roofline"]["traffic_profiled"])
for c in d.get("configs", []):
    if "runs" in c:
        print(c["name"], c.get("summary"))
        for r in c["runs"]:
            if r["coalesce"]: print("  ", {k: r[k] for k in ("dtype", "threads", "texts_per_s", "p50_us", "p99_us", "mean_texts_per_forward", "mean_forward_us", "mean_linger_us", "mismatches")})
PY

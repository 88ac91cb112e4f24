#!/bin/bash
# round 2, GPU pass 1: the parity suite, then the driver's bench command with the new `configs` array
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/r2_gputests_a.log 2>&1; echo "pytest rc=$?" >> $OUT/r2_gputests_a.log
tail -5 $OUT/r2_gputests_a.log
timeout 900 python bench.py --steps 50 --warmup 5 > $OUT/r2_bench_line_a.json 2> $OUT/r2_bench_a.err; echo "bench rc=$?"
tail -c 1500 $OUT/r2_bench_a.err
python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r2_bench_line_a.json")
try:
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("frac_actual_bytes"), d.get("sustained"))
    print(d["cpu_baseline"])
    for c in d.get("configs",[]): print(json.dumps(c)[:700])
except Exception as e: print("parse failed",e)
PY

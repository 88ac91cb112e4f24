#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4flat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py --steps 50 --warmup 5 --only-configs flat_1M_b256_k120 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_quick.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", {k: d["roofline"][k] for k in ("frac", "launch_us_mean", "step_frac_algorithmic")})
for c in d.get("configs", []): print(c["name"], c.get("ms_per_step"), {k: v for k, v in c.items() if "frac" in k})
print(d.get("latency_single_query"))
PY
tail -3 $OUT/bench_quick.err

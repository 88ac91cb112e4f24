#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SECONDS=0; timeout 900 python $ROOT/bench.py > $OUT/r4_bench_default.json 2> $OUT/r4_bench_default.err
echo "bench wall seconds: $SECONDS"; tail -3 $OUT/r4_bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/r4_bench_default.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "step_frac", d["roofline"]["step_frac_algorithmic"], "cpu", d["cpu_baseline"]["value"])
for c in d.get("configs", []): print(c["name"], c.get("ms_per_step"), c.get("wall_s"), {k: v for k, v in c.items() if k in ("texts_per_s", "ingest_texts_per_s", "encode_s", "queries_per_s", "mfma_frac")})
PY

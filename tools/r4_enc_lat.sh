#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/bench.py --steps 20 --warmup 3 --only-configs encoder --no-cpu-baseline --no-latency --sustained-s 0 > $OUT/r4_enc_lat.json 2> $OUT/r4_enc_lat.err
python - <<PY
import json
d = json.loads(open("$OUT/r4_enc_lat.json").read().strip().splitlines()[-1])
for c in d.get("configs", []): print(c["name"], {k: v for k, v in c.items() if k in ("ms_per_step", "p50_ms", "p95_ms", "texts_per_s")})
PY
tail -2 $OUT/r4_enc_lat.err

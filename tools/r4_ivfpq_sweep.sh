#!/bin/bash
# configs[3] at other batch sizes, both scans (SHODH_BENCH_IVFPQ_AB=1): where the list-major scan starts to pay
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4ivf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHODH_BENCH_IVFPQ_AB=1 timeout 500 python $ROOT/bench.py --steps 10 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for c in d['configs']: print(c['name'], c.get('ms_per_step'), json.dumps(c.get('batch_sweep_ms_per_step')))" > $OUT/sweep.txt
cat $OUT/sweep.txt

#!/bin/bash
# Round 6: configs[3] step with the probe-selection path (score pre-scan + one wave per query) against the general pipeline (SHODH_PROBE_SELECT=0), same box; timeline of the new step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd $ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
IV="python bench.py --steps 30 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
for rep in 1 2; do
  for ps in 0 1; do
    echo -n "SHODH_PROBE_SELECT=$ps  " | tee -a $OUT/psel_ab.txt
    SHODH_PROBE_SELECT=$ps SHODH_BENCH_EXTRAS_INPROC=1 timeout 600 $IV 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        for c in d.get('configs', []):
            print(c)
" | tee -a $OUT/psel_ab.txt
  done
done
bash tools/r6_ivfpq_timeline.sh

#!/bin/bash
# diagnostics: rocprofv3 kernel stats of one bench.py config (bash tools/prof_cfg.sh flat_1M_b256_k120)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python $ROOT/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs "$1" > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/pc/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]: print("%-70s %6s %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY

#!/bin/bash
# diagnostics: per-kernel time of the fused FFN under ablation masks (build with SHODH_EXTRA_FLAGS=-DSHODH_FFN_ABLATE first)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pa_$v; SHODH_FFN_ABLATE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$v -- python $ROOT/tools/enc_bench.py bf16 4096 > /dev/null 2>&1
  echo "== ABL $v"; python - <<PY
import csv,glob
f=glob.glob("/tmp/pa_$v/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:7]: print("%-60s %5s %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done

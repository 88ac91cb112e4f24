#!/bin/bash
# the driver's bench command N times in a row on one box (default 5): one line per run (value, ms per step, emit kernel us, clock) and the count of faults
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6final; mkdir -p $OUT; N=${1:-5}
cd /tmp; export TMPDIR=/tmp
: > $OUT/bench_soak.txt
for i in $(seq 1 $N); do
  python $ROOT/bench.py --steps 50 --warmup 5 > /tmp/soak_line.json 2> /tmp/soak_err.txt; rc=$?
  f=$(grep -c "Memory access fault" /tmp/soak_err.txt)
  python - >> $OUT/bench_soak.txt <<PY
import json
try:
    d=json.loads(open("/tmp/soak_line.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("run $i rc=$rc faults=$f value", d["value"], "ms", d["ms_per_step"], "emit_us", r["launch_us_mean"], "sclk", r.get("sclk_mhz"), "traffic", r["traffic"], "k120_ms", [c for c in d["configs"] if c["name"]=="flat_1M_b256_k120"][0]["ms_per_step"], "configs", len(d["configs"]), "bytes", len(open("/tmp/soak_line.json").read()))
except Exception as e:
    print("run $i rc=$rc faults=$f UNPARSED", e)
PY
done
cat $OUT/bench_soak.txt

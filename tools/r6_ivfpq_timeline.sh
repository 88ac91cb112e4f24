#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SHODH_BENCH_EXTRAS_INPROC=1
IV="python $ROOT/bench.py --steps 10 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
rm -rf /tmp/pt; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- $IV > /dev/null 2>&1
python $ROOT/tools/step_timeline.py /tmp/pt lm_merge_kernel 12 > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt

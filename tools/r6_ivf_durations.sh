#!/bin/bash
# Round 6: per-launch durations of the IVF-PQ step's kernels over the last launches of a bench run (the two query batches alternate)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp SHODH_BENCH_EXTRAS_INPROC=1
rm -rf /tmp/pt; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $ROOT/bench.py --steps 10 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq > /dev/null 2>&1
python $ROOT/tools/kernel_durations.py /tmp/pt lm_merge_kernel 262144 14 | tee $OUT/durations.txt

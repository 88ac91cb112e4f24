#!/bin/bash
# sample stride at k = 40 (search_ids with limit 10: k_index = limit * 4) on the contract corpus: bench.py --k 40 under SHODH_SAMPLE_STRIDE
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6thr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
: > $OUT/stride_k40.txt
for rep in 1 2; do for S in 0 28 24 20 16; do
  if [ "$S" = "0" ]; then unset SHODH_SAMPLE_STRIDE; else export SHODH_SAMPLE_STRIDE=$S; fi
  python $ROOT/bench.py --k 40 --steps 100 --warmup 10 --no-cpu-baseline --no-latency --sustained-s 0 --no-extra-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('stride $S ms_per_step', d['ms_per_step'], 'emit_us', d['roofline']['launch_us_mean'])" >> $OUT/stride_k40.txt
done; done
cat $OUT/stride_k40.txt

#!/bin/bash
# threshold_kernel section timers (needs tools/build_variant.sh prof -DSHODH_PROF): cycles per section of a few query slots at k = 10 / 40 / 120
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6thr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof
: > $OUT/thr.txt
for K in 10 40 120; do
  echo "== k $K" >> $OUT/thr.txt
  ITERS=2 K=$K timeout 300 python $ROOT/tools/step_time.py 2>&1 | grep "thr slot" | tail -7 >> $OUT/thr.txt
done
cat $OUT/thr.txt

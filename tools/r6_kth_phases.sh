#!/bin/bash
# block_kth_u32 section timers (tools/build_variant.sh prof -DSHODH_PROF): one workgroup's call in threshold_kernel (NT 256) and in the final stage (NT 512) at k = 10 / 40 / 120
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6thr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof
: > $OUT/kth.txt
for K in 10 40 120; do
  echo "== k $K" >> $OUT/kth.txt
  ITERS=2 K=$K timeout 300 python $ROOT/tools/step_time.py 2>&1 | grep "^kth " | tail -6 >> $OUT/kth.txt
done
cat $OUT/kth.txt

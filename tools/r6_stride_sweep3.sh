#!/bin/bash
# sample stride of the threshold pre-pass on the final library (measured error bound, threshold tail on sixteen lanes): step us at k = 10 / 40 / 120 / 300
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6thr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
: > $OUT/stride.txt
for rep in 1 2; do
for cfg in "10 0" "10 24" "10 48" "10 64" "40 0" "40 24" "40 48" "120 0" "120 9" "120 16" "120 20" "120 24" "120 32" "300 0" "300 8" "300 12"; do
  set -- $cfg; K=$1; S=$2
  if [ "$S" = "0" ]; then unset SHODH_SAMPLE_STRIDE; else export SHODH_SAMPLE_STRIDE=$S; fi
  echo "k=$K stride=${S} $(ITERS=300 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-140)" >> $OUT/stride.txt
done; done
cat $OUT/stride.txt

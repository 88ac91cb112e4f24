#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4flat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/sweep.txt
for K in 10 120; do for S in 6 9 12 16 24 32; do echo "k=$K S=$S $(SHODH_SAMPLE_STRIDE=$S ITERS=300 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-120)" >> $OUT/sweep.txt; done; done
cat $OUT/sweep.txt

#!/bin/bash
# Round 6 (after the hand-over change): sample stride against step time, batch 256 on 1M x 384: tools/r6_stride_sweep2.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2; do
for cfg in "120 9" "120 12" "120 16" "120 20" "120 24" "120 32" "10 24" "10 32" "10 48" "10 64" "300 6" "300 10" "300 16" "40 24" "40 32" "40 48"; do set -- $cfg
  echo -n "k $1 stride $2: " | tee -a $OUT/stride_sweep2.txt
  SHODH_SAMPLE_STRIDE=$2 NQ=256 K=$1 ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-170 | tee -a $OUT/stride_sweep2.txt
done; done

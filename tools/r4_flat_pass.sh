#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4flat; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py -x -q -m gpu 2>&1 | tail -6 > $OUT/tests.txt
cd /tmp
: > $OUT/steps.txt
for K in 10 40 120; do echo "k=$K" >> $OUT/steps.txt; ITERS=200 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 >> $OUT/steps.txt; done
timeout 200 python $ROOT/tools/stress_parity.py 2>&1 | tail -3 >> $OUT/steps.txt
cat $OUT/tests.txt $OUT/steps.txt

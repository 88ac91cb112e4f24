#!/bin/bash
# round 6: the exact fallback merges its own partial lists (one launch instead of two) -- parity tests, then the step with and without (SHODH_EXACT_FUSED_MERGE=0) on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fm; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
timeout 1500 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py tests/test_sharded_gpu.py tests/test_concurrent_gpu.py tests/test_round2_gpu.py -q -m gpu 2>&1 | tail -12 > $OUT/tests.txt
cd /tmp
: > $OUT/steps.txt
for rep in 1 2 3; do
for fm in 1 0; do
  for K in 10 120; do
    echo "fused_merge=$fm k=$K" >> $OUT/steps.txt
    SHODH_EXACT_FUSED_MERGE=$fm ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-120 >> $OUT/steps.txt
  done
done
done
timeout 300 python $ROOT/tools/stress_parity.py 2>&1 | tail -2 >> $OUT/steps.txt
cat $OUT/tests.txt $OUT/steps.txt

#!/bin/bash
# round 6: measured rounding residuals in the pre-scan's error bound -- parity tests on the new library, then new vs HEAD (libshodh_hip.so.head) on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6eps; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
timeout 1500 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py tests/test_ivfpq_gpu.py tests/test_ivfpq_listmajor_gpu.py tests/test_sharded_gpu.py tests/test_concurrent_gpu.py -q -m gpu 2>&1 | tail -12 > $OUT/tests.txt
cd /tmp
: > $OUT/steps.txt
for rep in 1 2; do
for lib in "" head; do
  for K in 10 120; do
    echo "lib=${lib:-new} k=$K" >> $OUT/steps.txt
    if [ -n "$lib" ]; then export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$lib; else unset SHODH_HIP_LIB; fi
    ITERS=300 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 >> $OUT/steps.txt
  done
done
done
for lib in "" head; do
  if [ -n "$lib" ]; then export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$lib; else unset SHODH_HIP_LIB; fi
  echo "lib=${lib:-new} latency" >> $OUT/steps.txt
  timeout 200 python $ROOT/tools/latency_probe.py 2>&1 | tail -6 >> $OUT/steps.txt
done
unset SHODH_HIP_LIB
timeout 300 python $ROOT/tools/stress_parity.py 2>&1 | tail -3 >> $OUT/steps.txt
cat $OUT/tests.txt $OUT/steps.txt

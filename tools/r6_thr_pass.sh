#!/bin/bash
# round 6: threshold_kernel's tail spread over sixteen lanes, batched LDS reads in the k-long loops of block_kth_u32 / the level placement -- parity tests, section timers
# (libshodh_hip.so.prof), then new vs the library before (libshodh_hip.so.prev) on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6thr; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
timeout 1500 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py tests/test_dynamic_threshold_gpu.py tests/test_probe_select_gpu.py tests/test_ivfpq_gpu.py tests/test_ivfpq_listmajor_gpu.py tests/test_ivfpq_fuzz_gpu.py tests/test_sharded_gpu.py tests/test_round2_gpu.py tests/test_concurrent_gpu.py -q -m gpu 2>&1 | tail -8 > $OUT/tests.txt
cd /tmp
bash $ROOT/tools/r6_thr_phases.sh > $OUT/thr_phases.txt 2>&1
export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof
for K in 10 120; do echo "== final stage k $K" >> $OUT/thr_phases.txt; ITERS=2 K=$K timeout 300 python $ROOT/tools/step_time.py 2>&1 | grep "^final q" | tail -5 >> $OUT/thr_phases.txt; done
unset SHODH_HIP_LIB
: > $OUT/steps.txt
for rep in 1 2 3; do for lib in "" prev; do
  if [ -n "$lib" ]; then export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$lib; else unset SHODH_HIP_LIB; fi
  for K in 10 40 120; do echo "lib=${lib:-new} k=$K $(ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-100)" >> $OUT/steps.txt; done
done; done
for lib in "" prev; do
  if [ -n "$lib" ]; then export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$lib; else unset SHODH_HIP_LIB; fi
  echo "lib=${lib:-new} single query" >> $OUT/steps.txt
  timeout 200 python $ROOT/tools/latency_probe.py 2>&1 | grep p50 | cut -c1-260 >> $OUT/steps.txt
done
unset SHODH_HIP_LIB
timeout 300 python $ROOT/tools/stress_parity.py 2>&1 | tail -1 >> $OUT/steps.txt
cat $OUT/tests.txt $OUT/thr_phases.txt $OUT/steps.txt

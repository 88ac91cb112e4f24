#!/bin/bash
# (historical: the [query][tile] layout and its -DMF_BM_T switch were removed again after this measurement -- profiles/r6_blockmax_layout_ab.txt, NOTEBOOK 12.18;
#  to repeat it, re-apply the bm_index() helper of the working tree this script ran on)
# round 6: sampled tile maxima stored [query][tile] (threshold_kernel reads a query's maxima contiguously) -- parity tests on the new library, threshold_kernel's section
# timers (libshodh_hip.so.prof), then per-kernel tables of a step with the new layout and the old one (libshodh_hip.so.bmt0, -DMF_BM_T=0) on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6bmt; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
timeout 1500 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py tests/test_dynamic_threshold_gpu.py tests/test_probe_select_gpu.py tests/test_ivfpq_gpu.py tests/test_sharded_gpu.py -q -m gpu 2>&1 | tail -8 > $OUT/tests.txt
cd /tmp
bash $ROOT/tools/r6_thr_phases.sh > $OUT/thr_phases.txt 2>&1
: > $OUT/steps.txt
for lib in "" bmt0; do
  if [ -n "$lib" ]; then export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$lib; else unset SHODH_HIP_LIB; fi
  for K in 10 120; do
    rm -rf /tmp/pk
    K=$K ITERS=300 GRAFT_REPO_ROOT=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python $ROOT/tools/step_time.py > /tmp/st.txt 2>&1
    echo "== lib=${lib:-new} k=$K: $(grep '^step' /tmp/st.txt | cut -c1-90)" >> $OUT/steps.txt
    python $ROOT/tools/stats_to_md.py /tmp/pk "x" | sed -n 6,12p | cut -c1-150 >> $OUT/steps.txt
  done
done
unset SHODH_HIP_LIB
for rep in 1 2; do for lib in "" bmt0; do
  if [ -n "$lib" ]; then export SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$lib; else unset SHODH_HIP_LIB; fi
  for K in 10 120; do echo "lib=${lib:-new} k=$K $(ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-100)" >> $OUT/steps.txt; done
done; done
unset SHODH_HIP_LIB
timeout 300 python $ROOT/tools/stress_parity.py 2>&1 | tail -1 >> $OUT/steps.txt
cat $OUT/tests.txt $OUT/thr_phases.txt $OUT/steps.txt

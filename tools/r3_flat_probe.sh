#!/bin/bash
# Round 3, flat step: kernel tables at k = 10 and k = 120 (batch 256, 1M rows) + the per-section cycles of the final stage (profiling variant)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 10 120; do
  rm -rf /tmp/pf$K; K=$K ITERS=40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf$K -- python $ROOT/tools/step_time.py > $OUT/r3_step_k$K.txt 2>/dev/null
  python $ROOT/tools/stats_to_md.py /tmp/pf$K "step_time K=$K" | head -16 > $OUT/r3_step_k${K}_stats.md
  cat $OUT/r3_step_k$K.txt; cat $OUT/r3_step_k${K}_stats.md
done
if [ -f $ROOT/shodh_memory_amd/libshodh_hip.so.prof ]; then
  for K in 10 120; do SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof K=$K ITERS=1 python $ROOT/tools/step_time.py 2>&1 | grep "^final" | tail -8 > $OUT/r3_final_prof_k$K.txt; cat $OUT/r3_final_prof_k$K.txt; done
fi
python $ROOT/tools/latency_probe.py > $OUT/r3_latency_probe.txt 2>/dev/null; cat $OUT/r3_latency_probe.txt

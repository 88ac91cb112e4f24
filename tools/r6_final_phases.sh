#!/bin/bash
# Round 6: phase timers (s_memtime ticks) of final_stage_kernel at the headline and recall shapes (tools/build_variant.sh prof -DSHODH_PROF)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for CFG in "256 120" "256 10" "1 120"; do set -- $CFG
echo "== nq $1 k $2"
SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof NQ=$1 K=$2 ITERS=1 timeout 200 python tools/step_time.py 2>&1 | grep "^final" | tail -7
done | tee $OUT/final_phases.txt

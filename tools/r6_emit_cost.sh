#!/bin/bash
# Round 6: what a survivor block and a queue drain cost the emit scan's waves (tools/build_variant.sh prof -DSHODH_PROF): tools/r6_emit_cost.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for CFG in "64 120" "256 120" "256 10"; do set -- $CFG
echo "== nq $1 k $2"
SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof NQ=$1 K=$2 ITERS=1 timeout 200 python tools/step_time.py 2>&1 | grep "^wave" | tail -16 | sort | cut -c1-230
done | tee -a $OUT/emit_cost.txt

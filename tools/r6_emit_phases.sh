#!/bin/bash
# Round 6: per-tile section cycles of the emit scan's waves (tools/build_variant.sh prof -DSHODH_PROF), headline shape and k = 120: tools/r6_emit_phases.sh [suffix ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for suf in ${@:-prof}; do
for CFG in "256 10" "256 120"; do set -- $CFG
echo "== $suf nq $1 k $2"
SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$suf NQ=$1 K=$2 ITERS=1 timeout 200 python tools/step_time.py 2>&1 | grep "^wave\|^block" | tail -10 | sort
done
done | tee -a $OUT/emit_phases.txt

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
for CFG in "64 120" "256 120" "64 10"; do set -- $CFG
echo "== nq $1 k $2"
SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof NQ=$1 K=$2 ITERS=1 timeout 200 python tools/step_time.py 2>&1 | grep "^final" | tail -6
done

#!/bin/bash
# Round 6: what the emit scan of a SMALL batch spends its time on (diagnostic build, results invalid under ablation): nothing emitted (8), the stream alone (40)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for NQ in 64 8; do for K in 120 10; do for A in 0 8 40; do
  echo -n "nq $NQ k $K ablate $A: " | tee -a $OUT/small_nq_probe.txt
  SHODH_ABLATE=$A SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.diag NQ=$NQ K=$K ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-110 | tee -a $OUT/small_nq_probe.txt
done; done; done

#!/bin/bash
# Round 6: what bounds the emit scan -- diagnostic builds (results invalid): SHODH_ABLATE 0 / 8 (nothing emitted) / 40 (stream alone: DMA, waits, barriers), with and without the tile barrier
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
for suf in diag diagnobar; do
  for A in 0 8 40; do
    echo -n "$suf ablate $A: " | tee -a $OUT/flat_ablate.txt
    SHODH_ABLATE=$A SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$suf NQ=256 K=10 ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-110 | tee -a $OUT/flat_ablate.txt
  done
done

#!/bin/bash
# Round 6: is the LDS port or the matrix pipe what the emit scan's chains wait for? Diagnostic builds (results invalid): every second A-fragment read left out
# (halflds), every second MFMA left out (halfmfma), against the diagnostic build of the product; nothing emitted (SHODH_ABLATE=8) and everything on (0).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2; do
for suf in "$@"; do
  for A in 8 0; do
    echo -n "$suf ablate $A: " | tee -a $OUT/bound_probe.txt
    SHODH_ABLATE=$A SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$suf NQ=256 K=10 ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-110 | tee -a $OUT/bound_probe.txt
  done
done
done

#!/bin/bash
# Round 6: library variants at several batch sizes: NQS="256 64" KS="10 120" tools/r6_variants_nq.sh <suffix|product> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2; do for suf in "$@"; do
  lib=libshodh_hip.so; [ "$suf" != "product" ] && lib=libshodh_hip.so.$suf
  for NQ in ${NQS:-256 64}; do for K in ${KS:-10 120}; do
    echo -n "$suf nq $NQ k $K: " | tee -a $OUT/variants_nq.txt
    SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib NQ=$NQ K=$K ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-180 | tee -a $OUT/variants_nq.txt
  done; done
done; done
# (the static bound for comparison: SHODH_DYN_THR=0 on the product)
if [ -n "$STATIC" ]; then for rep in 1 2; do for NQ in ${NQS:-256 64}; do for K in ${KS:-10 120}; do
  echo -n "static nq $NQ k $K: " | tee -a $OUT/variants_nq.txt
  SHODH_DYN_THR=0 NQ=$NQ K=$K ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-180 | tee -a $OUT/variants_nq.txt
done; done; done; fi

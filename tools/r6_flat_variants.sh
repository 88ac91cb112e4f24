#!/bin/bash
# Round 6: same-box step / emit-kernel times of library variants at the headline shape and k = 120: tools/r6_flat_variants.sh <suffix|product> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2; do
for suf in "$@"; do
  lib=libshodh_hip.so; [ "$suf" != "product" ] && lib=libshodh_hip.so.$suf
  for K in 10 120; do
    echo -n "$lib k $K: " | tee -a $OUT/flat_variants.txt
    SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib NQ=256 K=$K ITERS=200 timeout 200 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-150 | tee -a $OUT/flat_variants.txt
  done
done
done

#!/bin/bash
# Round 6: single-query scan, library variants against the product, alternating processes: tools/r6_solo_ab2.sh <suffix> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2 3; do for suf in product "$@"; do for K in 10 120; do
  lib=libshodh_hip.so; [ "$suf" != "product" ] && lib=libshodh_hip.so.$suf
  echo -n "$suf " | tee -a $OUT/solo_ab.txt
  K=$K SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib timeout 200 python tools/solo_time.py 2>&1 | grep "^k" | tee -a $OUT/solo_ab.txt
done; done; done

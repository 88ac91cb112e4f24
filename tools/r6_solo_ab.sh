#!/bin/bash
# Round 6: single-query scan A/B, alternating processes on one box: direct loads over contiguous slices (0 0) / over dealt blocks (0 1) / LDS-DMA ring (1 1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for rep in 1 2 3; do for MODE in "0 0" "0 1" "1 1"; do set -- $MODE; for K in 10 120; do
  echo -n "STREAM=$1 ILV=$2 " | tee -a $OUT/solo_ab.txt
  K=$K SHODH_SOLO_STREAM=$1 SHODH_SOLO_ILV=$2 timeout 200 python tools/solo_time.py 2>&1 | grep "^k" | tee -a $OUT/solo_ab.txt
done; done; done

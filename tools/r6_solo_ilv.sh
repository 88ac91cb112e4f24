#!/bin/bash
# Round 6: single-query scan -- direct loads over contiguous slices (SHODH_SOLO_STREAM=0 SHODH_SOLO_ILV=0), direct loads over blocks dealt round-robin
# (STREAM=0), LDS-DMA ring of dealt 32-row tiles (the product at 384-d): parity tests, then one host-pointer call at k = 10 / 120
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_single_query_gpu.py tests/test_concurrent_gpu.py tests/test_flat_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed\|error\|Error" | tail -5 | tee -a $OUT/solo_ilv.txt
for rep in 1 2; do for MODE in "0 0" "0 1" "1 1"; do set -- $MODE; for K in 10 120; do
  echo -n "STREAM=$1 ILV=$2 " | tee -a $OUT/solo_ilv.txt
  SHODH_SOLO_STREAM=$1 SHODH_SOLO_ILV=$2 timeout 200 python tools/small_batch_probe.py 1 $K 200 2>&1 | grep "^nq" | cut -c1-130 | tee -a $OUT/solo_ilv.txt
done; done; done

#!/bin/bash
# Round 6: 768 / 1024 dimensions, half-tile hand-over by progress words (the product) against the s_barrier form (libshodh_hip.so.b3bar): parity, then step times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed\|error" | tail -3 | tee -a $OUT/big3_ab.txt
for rep in 1 2; do for suf in b3bar product; do for D in 768 1024; do for K in 10 120; do
  lib=libshodh_hip.so; [ "$suf" != "product" ] && lib=libshodh_hip.so.$suf
  echo -n "$suf dim $D k $K: " | tee -a $OUT/big3_ab.txt
  SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib DIM=$D NQ=256 K=$K ITERS=100 timeout 300 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-150 | tee -a $OUT/big3_ab.txt
done; done; done; done

#!/bin/bash
# 768 / 1024 dimensions: parity tests under SHODH_BIG_SCAN_V1=$1 (default: the product selection), then the scan kernels against each other on ONE box
# (SHODH_BIG_SCAN_V1: 0 product selection, 1 the round-2 kernel everywhere, 3 the sixteen-queries-per-wave kernel everywhere)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4big; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
SHODH_BIG_SCAN_V1=${1:-0} timeout 600 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py -x -q -m gpu -k "768 or 1024 or dims or big or fuzz" 2>&1 | tail -4 > $OUT/tests.txt
cd /tmp
for V in 0 1; do
  SHODH_BIG_SCAN_V1=$V timeout 300 python $ROOT/bench.py --steps 20 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs bigdim 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for c in d['configs']: print('v=$V', c['name'], c.get('ms_per_step'), {k: v for k, v in c.items() if 'kernel_us' in k or 'kernel_hbm_frac_alg' in k or 'kernel_mfma_frac' in k})"
done > $OUT/ab.txt
cat $OUT/tests.txt $OUT/ab.txt

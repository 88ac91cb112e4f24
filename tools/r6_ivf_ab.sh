#!/bin/bash
# Round 6: IVF-PQ parity tests on the new library, then a same-box A/B of the configs[3] step: libshodh_hip.so.r6a (before) against the product library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd $ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_ivfpq_gpu.py tests/test_ivfpq_listmajor_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/ab_tests.txt
IV="python bench.py --steps 30 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
for rep in 1 2; do
  for lib in libshodh_hip.so.r6a libshodh_hip.so; do
    echo "== $lib" | tee -a $OUT/ab_lines.txt
    SHODH_BENCH_EXTRAS_INPROC=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib timeout 600 $IV 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        for c in d.get('configs', []):
            print(c)
" | tee -a $OUT/ab_lines.txt
  done
done

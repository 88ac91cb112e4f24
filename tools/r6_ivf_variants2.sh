#!/bin/bash
# Round 6: configs[3] step of library variants against the product, alternating on one box: tools/r6_ivf_variants2.sh <suffix> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
IV="python bench.py --steps 30 --warmup 5 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
for rep in 1 2; do for suf in product "$@"; do
  lib=libshodh_hip.so; [ "$suf" != "product" ] && lib=libshodh_hip.so.$suf
  echo -n "$suf: " | tee -a $OUT/ivf_variants2.txt
  SHODH_BENCH_EXTRAS_INPROC=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib timeout 600 $IV 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        for c in d.get('configs', []):
            if 'ivfpq' in c.get('name', ''): print(c)
" | tee -a $OUT/ivf_variants2.txt
done; done

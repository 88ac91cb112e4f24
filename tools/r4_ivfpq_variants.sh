#!/bin/bash
# configs[3] with every libshodh_hip.so.<variant> next to the product library, on ONE box (tools/build_variant.sh <suffix> -D...)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4ivf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IV="python $ROOT/bench.py --steps 10 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
for L in $ROOT/shodh_memory_amd/libshodh_hip.so $(ls $ROOT/shodh_memory_amd/libshodh_hip.so.* | grep -v "srchash\|prof"); do
  SHODH_HIP_LIB=$L timeout 400 $IV 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for c in d['configs']: print('$(basename $L)', c['name'], c.get('ms_per_step'), c.get('queries_per_s'))"
done > $OUT/variants.txt
cat $OUT/variants.txt

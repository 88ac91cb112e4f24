#!/bin/bash
# configs[3] (IVF-PQ 10M, nlist 4096, nprobe 32, batch 1024): query-major against list-major scan on ONE box, then kernel stats of the default
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4ivf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IV="python $ROOT/bench.py --steps 10 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
for LM in 0 1; do
  SHODH_ADC_LIST_MAJOR=$LM timeout 400 $IV 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for c in d['configs']: print('list_major=$LM', c['name'], c.get('ms_per_step'), c.get('queries_per_s'))"
done > $OUT/ab.txt
rm -rf /tmp/pi; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- $IV > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/pi "ivfpq" | grep -E "adc_|lm_|Memset|fill" | cut -c1-170 >> $OUT/ab.txt
cat $OUT/ab.txt

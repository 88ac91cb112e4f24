#!/bin/bash
# phase timers of adc_list_kernel (needs tools/build_variant.sh prof -DSHODH_LMPROF)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4ivf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof SHODH_ADC_LIST_MAJOR=1 timeout 120 python $ROOT/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq 2>&1 | grep nearprof | tail -10 > $OUT/phases.txt
cat $OUT/phases.txt

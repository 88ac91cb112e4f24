#!/bin/bash
# Round 6: step timeline of configs[3] and the phase timers of the persistent adc_list_kernel (tools/build_variant.sh prof -DSHODH_LMPROF)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd $ROOT; python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
bash tools/r6_ivfpq_timeline.sh
cd /tmp
SHODH_BENCH_EXTRAS_INPROC=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof timeout 300 python $ROOT/bench.py --steps 1 --warmup 0 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq 2>&1 | grep "lmprof" | tail -24 > $OUT/phases.txt
cat $OUT/phases.txt

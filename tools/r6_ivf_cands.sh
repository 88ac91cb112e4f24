#!/bin/bash
# Round 6: candidate-list statistics of the list-major scan at configs[3] (tools/build_variant.sh prof -DSHODH_LMPROF)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd /tmp
SHODH_BENCH_EXTRAS_INPROC=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof timeout 300 python $ROOT/bench.py --steps 3 --warmup 0 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq 2>&1 | grep "lmprof candidates" | sort | uniq -c > $OUT/cands.txt
cat $OUT/cands.txt

#!/bin/bash
# per-phase cycles of mfma_scan_big2_kernel (needs tools/build_variant.sh bigprof -DSHODH_BIGPROF)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4big; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.bigprof timeout 200 python $ROOT/bench.py --steps 2 --warmup 0 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs bigdim 2>&1 | grep bigprof | tail -16 > $OUT/phases.txt
cat $OUT/phases.txt

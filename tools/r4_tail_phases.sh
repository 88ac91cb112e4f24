#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHODH_ENC_PER_TEXT=1 SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof timeout 200 python $ROOT/tools/enc_bench.py int8 2>&1 | grep "outprof" | head -16 > $OUT/outprof.txt
cat $OUT/outprof.txt

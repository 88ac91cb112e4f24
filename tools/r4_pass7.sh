#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4p7; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_encoder_int8_pertext_gpu.py tests/test_encoder_int8_gpu.py tests/test_encoder_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -6 > $OUT/tests.txt
cd /tmp
for T in 8 40 128; do echo "$(python $ROOT/tools/enc_latency_probe.py int8 $T 2>/dev/null | tail -1)"; done > $OUT/lat.txt
for m in 0 1; do SHODH_ENC_PER_TEXT=$m timeout 300 python $ROOT/tools/enc_bench.py int8 4096 2>&1 | tail -1; done >> $OUT/lat.txt
SHODH_ENC_PER_TEXT=1 timeout 300 python $ROOT/tools/enc_bench.py int8 16 2>&1 | tail -1 >> $OUT/lat.txt
cat $OUT/tests.txt $OUT/lat.txt

#!/bin/bash
# round 4, pass 5: encoder test suite (both scopes) + timings
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4p5; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_encoder_int8_pertext_gpu.py tests/test_encoder_int8_gpu.py tests/test_encoder_fuzz_gpu.py tests/test_encoder_gpu.py tests/test_round2_gpu.py -x -q -m gpu 2>&1 | tail -12 > $OUT/encoder_tests.txt
cd /tmp
for m in 0 1; do SHODH_ENC_PER_TEXT=$m timeout 300 python $ROOT/tools/enc_bench.py int8 4096 2>&1 | tail -1; done > $OUT/enc_bench.txt
SHODH_ENC_EXPORT=u8 SHODH_ENC_PER_TEXT=1 timeout 300 python $ROOT/tools/enc_bench.py int8 4096 2>&1 | tail -1 >> $OUT/enc_bench.txt
SHODH_ENC_EXPORT=u8 timeout 300 python $ROOT/tools/enc_bench.py int8 4096 2>&1 | tail -1 >> $OUT/enc_bench.txt
timeout 300 python $ROOT/tools/enc_bench.py bf16 8192 2>&1 | tail -1 >> $OUT/enc_bench.txt
cat $OUT/encoder_tests.txt $OUT/enc_bench.txt

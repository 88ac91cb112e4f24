#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4zw; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_int8_pertext_gpu.py tests/test_encoder_int8_gpu.py -x -q -m gpu 2>&1 | tail -4 > $OUT/tests.txt
cd /tmp
: > $OUT/bench.txt
for PT in 1 0; do SHODH_ENC_EXPORT=u8 SHODH_ENC_PER_TEXT=$PT timeout 300 python $ROOT/tools/enc_bench.py int8 4096 2>&1 | tail -1 >> $OUT/bench.txt; done
rm -rf /tmp/pe9; SHODH_ENC_PER_TEXT=1 SHODH_ENC_EXPORT=u8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe9 -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/pe9 "export per-text" | sed -n 5,12p | cut -c1-150 >> $OUT/bench.txt
cat $OUT/tests.txt $OUT/bench.txt

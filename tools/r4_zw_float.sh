#!/bin/bash
# weight zero points in float arithmetic (stage bit 8) against the integer epilogue, export-style weights, both scopes, ONE box; then the INT8 tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4zw; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/bench.txt
for ST in 0x1EF 0xEF; do for PT in 1 0; do
  echo "stages $ST per_text $PT $(SHODH_INT8_STAGES=$ST SHODH_ENC_EXPORT=u8 SHODH_ENC_PER_TEXT=$PT timeout 300 python $ROOT/tools/enc_bench.py int8 4096 2>&1 | tail -1 | cut -c1-120)" >> $OUT/bench.txt
done; done
rm -rf /tmp/pz; SHODH_ENC_PER_TEXT=1 SHODH_ENC_EXPORT=u8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pz -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/pz "export per-text" | sed -n 5,11p | cut -c1-150 >> $OUT/bench.txt
cd $ROOT
timeout 900 python -m pytest tests/test_encoder_int8_pertext_gpu.py tests/test_encoder_int8_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $OUT/bench.txt
cat $OUT/bench.txt

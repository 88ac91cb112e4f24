#!/bin/bash
# Round 3, INT8 encoder: parity tests of the fused layer, then timing + rocprofv3 kernel table (run through gpurun).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_encoder_int8_gpu.py -x -q -s 2>&1 | tail -60 > $OUT/r3_int8_tests.txt
tail -25 $OUT/r3_int8_tests.txt
cd /tmp && export TMPDIR=/tmp
for ST in 47 15 0; do
  SHODH_INT8_STAGES=$ST timeout 300 python $ROOT/tools/enc_bench.py int8 > $OUT/r3_encoder_int8_line_st$ST.json 2>$OUT/r3_enc_st$ST.err; cat $OUT/r3_encoder_int8_line_st$ST.json
done
rm -rf /tmp/pe8; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe8 -- python $ROOT/tools/enc_bench.py int8 > $OUT/r3_encoder_int8_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe8 "round 3 -- rocprofv3 --kernel-trace --stats of \`python tools/enc_bench.py int8\` (MiniLM-L6 INT8, 4096 texts padded to 256 positions, lengths U[8,128]; 13 calls)" | head -24 > $OUT/r3_encoder_int8_kernel_stats.md
cat $OUT/r3_encoder_int8_kernel_stats.md

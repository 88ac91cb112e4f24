#!/bin/bash
# round 4, pass 2: kernel table of the per-text INT8 forward (v1) + sharded tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4p2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe8; SHODH_ENC_PER_TEXT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe8 -- python $ROOT/tools/enc_bench.py int8 > $OUT/pertext_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe8 "per-text v1" | head -24 > $OUT/pertext_kernel_stats.md
cd $ROOT
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_sharded_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -15 > $OUT/sharded_tests.txt
cat $OUT/pertext_line.json; cut -c1-200 $OUT/pertext_kernel_stats.md; cat $OUT/sharded_tests.txt

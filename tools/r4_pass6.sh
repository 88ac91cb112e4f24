#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4p6; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_encoder_fuzz_gpu.py tests/test_round2_gpu.py -x -q -m gpu 2>&1 | tail -6 > $OUT/tests.txt
cd /tmp
for T in 8 40 128; do echo "$(python $ROOT/tools/enc_latency_probe.py bf16 $T 2>/dev/null | tail -1)"; done > $OUT/lat.txt
rm -rf /tmp/pl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python $ROOT/tools/enc_latency_probe.py bf16 40 > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/pl "bf16 one text" | sed -n 5,16p | cut -c1-150 >> $OUT/lat.txt
cat $OUT/tests.txt $OUT/lat.txt

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4lat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ST in 0xEF 0x4F 0x47 0x0F 0; do echo "stages $ST $(SHODH_INT8_STAGES=$ST python $ROOT/tools/enc_latency_probe.py int8 40 2>/dev/null | tail -1)"; done > $OUT/int8_stages.txt
cat $OUT/int8_stages.txt

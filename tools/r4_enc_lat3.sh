#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4lat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in 8 40 128; do
  echo "fused   $(python $ROOT/tools/enc_latency_probe.py bf16 $T 2>/dev/null | tail -1)"
  echo "unfused $(SHODH_ENC_UNFUSED=1 python $ROOT/tools/enc_latency_probe.py bf16 $T 2>/dev/null | tail -1)"
done > $OUT/unfused.txt
cat $OUT/unfused.txt

#!/bin/bash
# one text per call on the INT8 model: which kernels the 0.59 ms of device time are
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5b1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
for D in int8 bf16; do
rm -rf /tmp/pb1; SHODH_ENC_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -- python $ROOT/tools/enc_latency_probe.py $D 40 > $OUT/line_$D.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pb1 "one text (40 tokens) per call, $D, 320 calls" | head -30 > $OUT/kernels_$D.md
cat $OUT/line_$D.json; cut -c1-150 $OUT/kernels_$D.md | sed -n 5,26p
done

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4lat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for D in bf16 int8; do
  python $ROOT/tools/enc_latency_probe.py $D 40 2>/dev/null | tail -1 > $OUT/lat_$D.json
  rm -rf /tmp/pl_$D; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$D -- python $ROOT/tools/enc_latency_probe.py $D 40 > /dev/null 2>&1
  python $ROOT/tools/stats_to_md.py /tmp/pl_$D "$D one text" | sed -n 5,24p | cut -c1-130 > $OUT/stats_$D.md
  cat $OUT/lat_$D.json; cat $OUT/stats_$D.md
done

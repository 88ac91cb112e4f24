#!/bin/bash
# the kernels of one 768-d / 1024-d search step (rocprofv3 kernel trace -> tools/step_timeline.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4big; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/timeline.txt
for DIM in 768 1024; do
  rm -rf /tmp/pb; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -- python $ROOT/tools/bigdim_probe.py $DIM 256 > /dev/null 2>&1
  echo "== $DIM" >> $OUT/timeline.txt
  python $ROOT/tools/step_timeline.py /tmp/pb merge_topk_kernel -2 >> $OUT/timeline.txt 2>&1
done
cat $OUT/timeline.txt

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4flat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 10 120; do
  echo "== k=$K prof" >> $OUT/prof.txt
  ITERS=3 K=$K SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.prof timeout 200 python $ROOT/tools/step_time.py 2>&1 | grep -E "per tile|block 200|step" | head -14 >> $OUT/prof.txt
  echo "== k=$K base" >> $OUT/prof.txt
  ITERS=50 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -2 >> $OUT/prof.txt
done
rm -rf /tmp/pk; K=120 ITERS=50 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python $ROOT/tools/step_time.py > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/pk "k=120" | sed -n 5,14p | cut -c1-150 >> $OUT/prof.txt
cat $OUT/prof.txt

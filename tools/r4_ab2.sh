#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4flat; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/ab2.txt
for rep in 1 2; do for K in 10 120; do for V in base nowait nodel; do
  LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$V; [ $V = base ] && LIB=$ROOT/shodh_memory_amd/libshodh_hip.so
  echo "k=$K $V $(SHODH_HIP_LIB=$LIB ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-100)" >> $OUT/ab2.txt
done; done; done
cat $OUT/ab2.txt

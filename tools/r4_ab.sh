#!/bin/bash
# A/B on ONE box: product library vs a variant (tools/build_variant.sh <suffix> ...): usage r4_ab.sh <suffix> [K ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4flat; mkdir -p $OUT
SUF=$1; shift; KS=${@:-10 120}
cd /tmp && export TMPDIR=/tmp
: > $OUT/ab_$SUF.txt
for rep in 1 2; do for K in $KS; do
  echo "k=$K new  $(ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-100)" >> $OUT/ab_$SUF.txt
  echo "k=$K $SUF $(SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$SUF ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-100)" >> $OUT/ab_$SUF.txt
done; done
cat $OUT/ab_$SUF.txt

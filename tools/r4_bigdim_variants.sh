#!/bin/bash
# 768 / 1024-d scan with every libshodh_hip.so.<variant> next to the product library, on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4big; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in $ROOT/shodh_memory_amd/libshodh_hip.so $(ls $ROOT/shodh_memory_amd/libshodh_hip.so.* | grep -v "srchash\|prof"); do
  for DIM in 768 1024; do echo "$(basename $L) $(SHODH_HIP_LIB=$L timeout 100 python $ROOT/tools/bigdim_probe.py $DIM 256 2>/dev/null | tail -1)"; done
done > $OUT/variants.txt
cat $OUT/variants.txt

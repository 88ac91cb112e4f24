#!/bin/bash
# mfma_scan_big2_kernel with parts switched off at compile time (results invalid): for m in 16 32 64 128 112 240; do tools/build_variant.sh ba$m -DSHODH_BIG_ABL=$m; done
# bits: 16 no global loads after the prologue, 32 no staging writes, 64 no exchange, 128 no MFMAs; SHODH_ABLATE=8: nothing emitted
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4big; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/ablate.txt
for L in $ROOT/shodh_memory_amd/libshodh_hip.so $ROOT/shodh_memory_amd/libshodh_hip.so.ba*; do
  echo "$(basename $L): $(SHODH_HIP_LIB=$L timeout 100 python $ROOT/tools/bigdim_probe.py 768 256 2>/dev/null | tail -1)" >> $OUT/ablate.txt
done
cat $OUT/ablate.txt

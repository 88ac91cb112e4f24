#!/bin/bash
# FFN-down kernel (i8_ktile_ln_kernel) with parts switched off at compile time: build the variants first,
#   for m in 1 2 3 4 8 16 32 63; do tools/build_variant.sh kt$m -DSHODH_KT_ABL=$m; done      (results invalid for masks != 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4kt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SHODH_ENC_PER_TEXT=1
: > $OUT/abl.txt
for L in $ROOT/shodh_memory_amd/libshodh_hip.so $ROOT/shodh_memory_amd/libshodh_hip.so.kt*; do
  rm -rf /tmp/pk; SHODH_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
  echo "$(basename $L): $(python $ROOT/tools/stats_to_md.py /tmp/pk x | grep i8_ktile | cut -d'|' -f3-5)" >> $OUT/abl.txt
done
cat $OUT/abl.txt

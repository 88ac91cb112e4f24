#!/bin/bash
# INT8 on export-style weights, per-text scope: every libshodh_hip.so.z* variant against the product library on ONE box (forward ms + the two FFN-up kernels)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4zw; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/variants.txt
for L in $ROOT/shodh_memory_amd/libshodh_hip.so $ROOT/shodh_memory_amd/libshodh_hip.so.z*; do
  rm -rf /tmp/pz; SHODH_HIP_LIB=$L SHODH_ENC_PER_TEXT=1 SHODH_ENC_EXPORT=u8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pz -- python $ROOT/tools/enc_bench.py int8 > /tmp/zline.txt 2>/dev/null
  echo "$(basename $L): $(tail -1 /tmp/zline.txt | cut -c1-60) | $(python $ROOT/tools/stats_to_md.py /tmp/pz x | grep 'i8_stream_gelu' | cut -d'|' -f2,5 | tr '\n' ' ' | cut -c1-220)" >> $OUT/variants.txt
done
cat $OUT/variants.txt

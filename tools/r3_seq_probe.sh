#!/bin/bash
# INT8 layer probes: timing + rocprofv3 kernel table of build variants. usage: tools/r3_seq_probe.sh <variant suffixes...>  ("base" = the product library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$V
  [ "$V" = base ] && LIB=$ROOT/shodh_memory_amd/libshodh_hip.so
  echo "== $V"
  SHODH_HIP_LIB=$LIB timeout 120 python $ROOT/tools/enc_bench.py int8 2>&1 | grep '"encoder"'
  rm -rf /tmp/pe_$V; SHODH_HIP_LIB=$LIB timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$V -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
  python $ROOT/tools/stats_to_md.py /tmp/pe_$V "$V" | head -15 | tail -9 | cut -c1-150
done

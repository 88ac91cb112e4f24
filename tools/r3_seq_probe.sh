#!/bin/bash
# INT8 layer probes: timing + rocprofv3 kernel table. usage: tools/r3_seq_probe.sh <stage masks...>  (SHODH_INT8_STAGES values; SHODH_HIP_LIB is passed through)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  echo "== stages $V"
  SHODH_INT8_STAGES=$V timeout 300 python $ROOT/tools/enc_bench.py int8 2>&1 | grep '"encoder"'
  rm -rf /tmp/pe_$V; SHODH_INT8_STAGES=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$V -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
  python $ROOT/tools/stats_to_md.py /tmp/pe_$V "stages $V" | head -15 | tail -9 | cut -c1-150
done

#!/bin/bash
# Round 6: duration of probe_select_kernel (and the score pre-scan) in tools/psel_probe.py for library variants: tools/r6_psel_kernel_time.sh <suffix|product> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for suf in "$@"; do
  lib=libshodh_hip.so; [ "$suf" != "product" ] && lib=libshodh_hip.so.$suf
  rm -rf /tmp/pk; SHODH_HIP_LIB=$ROOT/shodh_memory_amd/$lib ITERS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python $ROOT/tools/psel_probe.py > /dev/null 2>&1
  echo "== $lib"; python $ROOT/tools/stats_to_md.py /tmp/pk x | grep "probe_select\|mfma_scan_kernel<2\|convert_queries\|adc_\|lm_" | cut -c1-150
done

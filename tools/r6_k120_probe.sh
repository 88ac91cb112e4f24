#!/bin/bash
# Round 6: k = 120 and the threshold that tightens during the emit scan. tools/step_time.py per setting, one box.
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
run() { # label, env...
  label=$1; shift
  for k in ${KS:-10 120}; do
    echo -n "$label k=$k: "; env "$@" K=$k ITERS=${ITERS:-600} timeout 300 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-220
  done
}
run static SHODH_DYN_THR=0
run dynamic SHODH_DYN_THR=1
run static_again SHODH_DYN_THR=0
run dynamic_again SHODH_DYN_THR=1
for v in $VARIANTS; do [ -f shodh_memory_amd/libshodh_hip.so.$v ] && run dynamic_$v SHODH_HIP_LIB=$PWD/shodh_memory_amd/libshodh_hip.so.$v; done
[ -n "$DIAG" ] && [ -f shodh_memory_amd/libshodh_hip.so.diag ] && run mechanism_without_publishing SHODH_HIP_LIB=$PWD/shodh_memory_amd/libshodh_hip.so.diag SHODH_ABLATE=16
[ -n "$DIAG" ] && [ -f shodh_memory_amd/libshodh_hip.so.diag ] && run floor_nothing_emitted SHODH_HIP_LIB=$PWD/shodh_memory_amd/libshodh_hip.so.diag SHODH_ABLATE=8
for s in ${STRIDES}; do run dynamic_stride$s SHODH_DYN_THR=1 SHODH_SAMPLE_STRIDE=$s; done

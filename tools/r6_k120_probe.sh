#!/bin/bash
# Round 6: where does k = 120 lose its time? Variants of the library side by side on ONE box (tools/build_variant.sh), tools/step_time.py per variant.
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
run() { # label, lib suffix, env...
  label=$1; suf=$2; shift 2
  for k in 10 120; do
    lib=shodh_memory_amd/libshodh_hip.so; [ -n "$suf" ] && lib=$lib.$suf
    echo -n "$label k=$k: "; env "$@" SHODH_HIP_LIB=$PWD/$lib K=$k ITERS=${ITERS:-600} timeout 300 python tools/step_time.py 2>&1 | grep "^step" | cut -c1-200
  done
}
run product "" X=1
run product_again "" X=1
[ -f shodh_memory_amd/libshodh_hip.so.diag ] && run floor_nothing_emitted diag SHODH_ABLATE=8
[ -f shodh_memory_amd/libshodh_hip.so.slots16 ] && run slots16 slots16 X=1
for v in $EXTRA_VARIANTS; do [ -f shodh_memory_amd/libshodh_hip.so.$v ] && run $v $v X=1; done
run stride16 "" SHODH_SAMPLE_STRIDE=16

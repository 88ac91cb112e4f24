#!/bin/bash
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for k in 10 120; do for d in 0 1; do echo "== k=$k dyn=$d"; SHODH_DYN_THR=$d SHODH_HIP_LIB=$PWD/shodh_memory_amd/libshodh_hip.so.prof K=$k ITERS=3 timeout 120 python tools/step_time.py 2>&1 | grep -E "^wave|^block" | sort | uniq -c | sort -rn | head -12; done; done
echo "== after the change: threshold kernel"; STRIDES="" KS="120" tools/r6_k120_probe.sh 2>&1 | head -4

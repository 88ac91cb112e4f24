#!/bin/bash
# round 6, final library: the bench's entries under the allocation guard, then long randomised parity runs (device-pointer batches against the exact scan; host-pointer calls of 1 .. 200 queries)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6final; mkdir -p $OUT
cd $ROOT; export TMPDIR=/tmp
bash tools/r6_guard.sh 1 bench 2>&1 | tail -6 > $OUT/guard_bench.txt
cp gpurun_out/guard_m1/summary.txt $OUT/guard_bench_summary.txt 2>/dev/null
unset SHODH_GUARD AMD_SERIALIZE_KERNEL
cd /tmp
timeout 1200 python $ROOT/tools/stress_parity.py 2000 300000 384,128,768,1024 2>&1 | tail -2 > $OUT/stress_parity.txt
timeout 900 python $ROOT/tools/stress_host_parity.py 1500 2>&1 | tail -3 > $OUT/stress_host_parity.txt
cat $OUT/guard_bench.txt $OUT/stress_parity.txt $OUT/stress_host_parity.txt

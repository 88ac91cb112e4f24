#!/bin/bash
# last pass of round 2: the whole GPU suite, the driver's bench line, the single-query profile (outputs under gpurun_out/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT && timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/r2_final_gpu_tests.txt
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 50 --warmup 5 > $OUT/r2_bench_line.json 2> $OUT/r2_bench.err; tail -c 200 $OUT/r2_bench_line.json; echo
bash $ROOT/tools/r2_single_query_profile.sh 2>&1 | tail -12

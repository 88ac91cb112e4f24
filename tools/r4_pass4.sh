#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/tools/r4_pass3.sh > /dev/null 2>&1
bash $ROOT/tools/r4_prof.sh > /dev/null 2>&1
cd $ROOT; O=gpurun_out/r4p3
cat $O/tail_vs_v1.txt $O/pertext_tests.txt $O/pertext_line.json $O/batch_line.json; sed -n 5,12p $O/pertext_kernel_stats.md | cut -c1-150; head -8 gpurun_out/r4prof/outprof.txt

#!/bin/bash
# final stage with 256 instead of 512 threads at small k (SHODH_FINAL_NT): step us, three repetitions -- no gain at k = 10 (250.2 vs 250.7), worse at k = 40 (268.5 vs 267.0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
for rep in 1 2 3; do for NT in 0 256; do for K in 10 40; do
  if [ "$NT" = "0" ]; then unset SHODH_FINAL_NT; else export SHODH_FINAL_NT=$NT; fi
  echo "final_nt=$NT k=$K $(ITERS=400 K=$K timeout 200 python $ROOT/tools/step_time.py 2>&1 | tail -1 | cut -c1-40)"
done; done; done

#!/bin/bash
# diagnostics: time the emit scan with the survivor path switched off (SHODH_ABLATE=8: results are NOT valid in that mode).
# Needs a diagnostic build of the library: SHODH_EXTRA_FLAGS=-DSHODH_DIAG python -m shodh_memory_amd.build --force
# (the production build compiles the switch out); rebuild without the flag afterwards.
for a in ${ABL:-0 8}; do
  echo "== SHODH_ABLATE=$a"
  SHODH_ABLATE=$a ITERS=${ITERS:-2000} NQ=${NQ:-256} timeout 200 python tools/step_time.py 2>&1 | grep "^step"
done

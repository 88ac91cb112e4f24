#!/bin/bash
# diagnostics: time the emit scan with the survivor path switched off (SHODH_ABLATE=8: results are NOT valid in that mode)
for a in ${ABL:-0 8}; do
  echo "== SHODH_ABLATE=$a"
  SHODH_ABLATE=$a ITERS=${ITERS:-2000} NQ=${NQ:-256} timeout 200 python tools/step_time.py 2>&1 | grep "^step"
done

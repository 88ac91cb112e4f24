#!/bin/bash
# diagnostics: time the emit scan with parts removed (results are NOT valid in these modes)
for a in ${ABL:-0 1 2 3}; do
  echo "== SHODH_ABLATE=$a"
  SHODH_ABLATE=$a timeout 200 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import shodh_memory_amd as S
dev = torch.device("cuda", 0)
rows = bench.synth_rows(torch, 1_000_000, 384, 1, dev)
q = bench.synth_rows(torch, 256, 384, 2, dev)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, scan_mode=2, reserve_rows=1_000_000))
idx.build(rows)
for _ in range(3): idx.search_batch_device(q, 10)
torch.cuda.synchronize(); idx.kernel_timing(True)
for _ in range(20): idx.search_batch_device(q, 10)
torch.cuda.synchronize()
print("emit kernel mean/min us:", idx.kernel_timing(True))
PY
done

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5p3; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
cat > /tmp/trace_run.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import bench as B, shodh_memory_amd as S
from shodh_memory_amd import _lib as L
from tools import callers as CL
dev = torch.device("cuda:0")
q = B.synth_rows(torch, 256, 384, B.SEED + 1, dev); rows = B.synth_rows(torch, 1_000_000, 384, B.SEED, dev, adversarial_queries=q)
idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=1_000_000)); idx.build(rows)
qh = q.cpu().numpy()
T = int(sys.argv[1])
r = CL.search(L.lib(), idx.handle, qh, 120, threads=T, calls_per_thread=40, warmup=3)
print(r.as_dict("queries"), idx.coalesce_stats())
PY
for P in 0 1; do
echo "== predictive $P"
SHODH_COALESCE_PREDICTIVE=$P SHODH_COALESCE_TRACE=1 timeout 300 python /tmp/trace_run.py 64 2>&1 | grep "combiner\|queries_per_s" | tail -14
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-configs concurrent_callers > $OUT/line.json 2> $OUT/err.txt
python - <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5p3/line.json").read().strip().splitlines()[-1])
for c in d.get("configs", []):
    if "runs" in c:
        print(c.get("summary"))
        for r in c["runs"]:
            if r["coalesce"]: print("  k %3d T %3d: %8.0f q/s p50 %6.1f p99 %6.1f | callers/pass %6.2f pass %6.1f us linger %5.1f us" % (r["k"], r["threads"], r["queries_per_s"], r["p50_us"], r["p99_us"], r["mean_callers_per_pass"], r["mean_pass_us"], r["mean_linger_us"]))
            if r["mismatches"] or r["errors"]: print("  WRONG:", r)
PY

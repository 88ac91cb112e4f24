#!/bin/bash
# round 5: the concurrency tests and the concurrent bench entries, repeated (rare races show up as one wrong list in thousands of calls)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5stress; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_concurrent_gpu.py -x -q -m gpu 2>&1 | tail -1; done | tee $OUT/tests.txt
for i in 1 2 3 4; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs concurrent_callers,concurrent_encode_callers > $OUT/line_$i.json 2>/dev/null
python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5stress/line_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
for c in d["configs"]:
    bad = [r for r in c["runs"] if r["mismatches"] or r["errors"]]
    print(sys.argv[1], c["name"], c["summary"], "calls", sum(r["calls"] for r in c["runs"]), "BAD" if bad else "all equal", bad[:2])
PY
done | tee $OUT/bench.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs flat_10M_b256,cfg4_ivfpq > $OUT/line_big.json 2>$OUT/err_big.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5stress/line_big.json").read().strip().splitlines()[-1])
for c in d["configs"]:
    print(c["name"], c["ms_per_step"], {k: v for k, v in c.items() if k.startswith("callers_")})
PY

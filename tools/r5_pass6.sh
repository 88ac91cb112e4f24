#!/bin/bash
# round 5, pass 6: zero-copy small multi-query passes -- parity (flat suites + concurrency), then the pass time per batch size and the concurrent entry
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5p6; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
timeout 1500 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_single_query_gpu.py tests/test_concurrent_gpu.py tests/test_sharded_gpu.py -x -q -m gpu 2>&1 | tail -3
for Z in 0 1; do echo "SHODH_ZERO_COPY_MAX_NQ=$((Z*128))"; SHODH_ZERO_COPY_MAX_NQ=$((Z*128)) timeout 300 python tools/small_batch_probe.py 2>&1 | grep "^nq" | cut -c1-110; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs concurrent_callers > $OUT/line.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p6/line.json").read().strip().splitlines()[-1])
for c in d["configs"]:
    print(c["summary"])
    for r in c["runs"]:
        if r["coalesce"]: print("  k %3d T %3d: %8.0f q/s p50 %6.1f p99 %6.1f | callers/pass %6.2f pass %6.1f us linger %5.1f us  mism %d" % (r["k"], r["threads"], r["queries_per_s"], r["p50_us"], r["p99_us"], r["mean_callers_per_pass"], r["mean_pass_us"], r["mean_linger_us"], r["mismatches"]))
PY

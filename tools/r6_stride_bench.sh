#!/bin/bash
# sample stride at k = 120 / 40 on the bench's own corpora (contract corpus with tombstones and duplicates; 1000 vMF-like clusters): bench.py entries under SHODH_SAMPLE_STRIDE
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6thr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp SHODH_TRUST_PREBUILT=1
: > $OUT/stride_bench.txt
for S in 0 16 20 24 28; do
  if [ "$S" = "0" ]; then unset SHODH_SAMPLE_STRIDE; else export SHODH_SAMPLE_STRIDE=$S; fi
  python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs flat_1M_b256_k120,flat_1M_clustered_b256 > /dev/null 2>&1
  python - >> $OUT/stride_bench.txt <<PY
import json
d=json.load(open("$ROOT/gpurun_out/bench_detail.json"))
o={"stride":"$S","k10_ms":d["ms_per_step"]}
for c in d["configs"]:
    if c["name"]=="flat_1M_b256_k120": o["k120_ms"]=c["ms_per_step"]; o["k120_emit_us"]=c.get("scan_kernel_us"); o["k120_emitted"]=c.get("survivors_emitted_per_query")
    if c["name"]=="flat_1M_clustered_b256": o["clu_k10_ms"]=c["ms_per_step"]; o["clu_k120_ms"]=(c.get("k120") or {}).get("ms_per_step")
print(json.dumps(o))
PY
done
cat $OUT/stride_bench.txt

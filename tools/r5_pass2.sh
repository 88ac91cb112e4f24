#!/bin/bash
# round 5, pass 2: where the time of a coalesced pass goes (host call time and kernel breakdown per batch size), and whether a fresh box builds from source
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5p2; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
ls -la /dev/kfd shodh_memory_amd/build 2>&1 | head -8 > $OUT/box.txt
( time python -c "import __graft_entry__ as g; g.build()" ) 2>&1 | tail -5 >> $OUT/box.txt
cat $OUT/box.txt
timeout 300 python tools/small_batch_probe.py 2>&1 | grep "^nq" > $OUT/small_batch.txt
cat $OUT/small_batch.txt
cd /tmp
for SH in "64 120" "16 120" "64 10"; do
  set -- $SH
  rm -rf /tmp/prof_sb; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sb -o sb -- python $ROOT/tools/small_batch_probe.py $1 $2 60 > /dev/null 2>&1
  python $ROOT/tools/stats_to_md.py /tmp/prof_sb "one host-pointer search of $1 queries, k = $2, 1M x 384 (65 calls)" > $OUT/kernels_nq$1_k$2.md
  grep -v "^#\|^Names\|^$" $OUT/kernels_nq$1_k$2.md | head -14
done
cd $ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-configs concurrent_callers > $OUT/line.json 2> $OUT/err.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p2/line.json").read().strip().splitlines()[-1])
for c in d.get("configs", []):
    if "runs" in c:
        print(c["name"], c.get("summary"))
        for r in c["runs"]:
            if r["coalesce"]: print("  ", r)
PY

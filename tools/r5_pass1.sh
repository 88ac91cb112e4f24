#!/bin/bash
# round 5, first GPU pass: from-source build on the box, the new concurrency tests, the whole GPU suite, the concurrent-caller bench entries
cd ${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; mkdir -p gpurun_out; OUT=gpurun_out
nproc > $OUT/r5_box.txt; lscpu | grep -i "model name\|hypervisor\|^CPU(s)" >> $OUT/r5_box.txt
( time python -c "import __graft_entry__ as g; g.build()" ) 2>&1 | tail -6 > $OUT/r5_build_from_source.txt
cat $OUT/r5_build_from_source.txt
timeout 900 python -m pytest tests/test_concurrent_gpu.py -x -q -s -m gpu 2>&1 | tail -40 > $OUT/r5_concurrent_tests.txt
cat $OUT/r5_concurrent_tests.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -12 > $OUT/r5_gpu_tests.txt
cat $OUT/r5_gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-configs concurrent_callers,concurrent_encode_callers,sharded_c_abi_1M > $OUT/r5_concurrent_line.json 2> $OUT/r5_concurrent_err.txt
tail -3 $OUT/r5_concurrent_err.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_concurrent_line.json").read().strip().splitlines()[-1])
print("main:", d["value"], d["ms_per_step"])
for c in d.get("configs", []):
    if "runs" in c:
        print(c["name"], c.get("summary"))
        for r in c["runs"]:
            print("  ", r)
    elif "layouts" in c:
        print(c["name"], c["single_index_host_api_ms_per_step"], [(l["layout"], l["ms_per_step"], l["device_pointer_ms_per_step"], l["identical_to_single_index"]) for l in c["layouts"]])
PY

#!/bin/bash
# Round 6: kernel times of one IVF-PQ step (10M rows, nlist 4096, nprobe 32, batch 1024)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
IV="python $ROOT/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
rm -rf /tmp/pi; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- $IV > $OUT/r6_ivfpq_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pi "round 6 -- rocprofv3 --kernel-trace --stats of \`bench.py --only-configs cfg4_ivfpq\` (10M rows, nlist 4096, nprobe 32, batch 1024; the index build's kernels are in the table too)" | head -30 > $OUT/r6_ivfpq_kernel_stats.md
cat $OUT/r6_ivfpq_kernel_stats.md | cut -c1-170

#!/bin/bash
# Round 6: rocprofv3 kernel stats of the configs[3] step, the IVF-PQ kernels only
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6ivf; mkdir -p $OUT
cd $ROOT; python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
export SHODH_BENCH_EXTRAS_INPROC=1
IV="python $ROOT/bench.py --steps 20 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
rm -rf /tmp/pi; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- $IV > $OUT/stats_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pi "round 6 -- rocprofv3 --kernel-trace --stats of bench.py --only-configs cfg4_ivfpq" > $OUT/stats_full.md
grep -i "adc_\|lm_\|final_stage\|mfma_scan\|threshold\|convert\|merge_topk\|flat_exact" $OUT/stats_full.md | cut -c1-200

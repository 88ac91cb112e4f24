R=r1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 50 --warmup 5 > $OUT/${R}_bench_line.json 2> $OUT/${R}_bench.err
rm -rf /tmp/prof_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-latency > $OUT/${R}_bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
python $ROOT/tools/stats_to_md.py /tmp/prof_stats "round 1 -- rocprofv3 --kernel-trace --stats of \`python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-latency\` (1M x 384 f32, 256-query batches, top-10, 1 x MI355X)" > $OUT/${R}_bench_kernel_stats.md
cd $ROOT
python bench_extra.py ivfpq 2>/dev/null | tail -1 > $OUT/ivfpq_after_fix.json
python bench_extra.py latency 2>/dev/null | grep "exact-order f32 scan only" > $OUT/exact_after_fix.json
head -c 300 $OUT/${R}_bench_line.json; echo; cat $OUT/ivfpq_after_fix.json | cut -c1-200; cat $OUT/exact_after_fix.json | cut -c1-260

#!/bin/bash
# One GPU-box pass that regenerates everything under profiles/ for round $1 (default r1): bench line, rocprofv3 kernel stats,
# PMC traffic, secondary benchmarks. Outputs go to gpurun_out/ (merged back by gpurun); copy what is to be judged into profiles/.
R=${1:-r1}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 50 --warmup 5 > $OUT/${R}_bench_line.json 2> $OUT/${R}_bench.err
rm -rf /tmp/prof_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-latency > $OUT/${R}_bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
python $ROOT/tools/stats_to_md.py /tmp/prof_stats "round ${R#r} -- rocprofv3 --kernel-trace --stats of \`python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-latency\` (1M x 384 f32, 256-query batches, top-10, 1 x MI355X)" > $OUT/${R}_bench_kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python $ROOT/bench.py --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency > /dev/null 2>&1; done
python $ROOT/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 1000000 384 256 10 > $OUT/pmc_traffic.json
for W in encoder ivfpq; do
  rm -rf /tmp/prof_$W
  ARGS=""; [ $W = ivfpq ] && ARGS="--rows 4000000"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -- python $ROOT/bench_extra.py $W $ARGS > /dev/null 2>&1
  python $ROOT/tools/stats_to_md.py /tmp/prof_$W "round ${R#r} -- rocprofv3 --kernel-trace --stats of \`python bench_extra.py $W $ARGS\`" > $OUT/${R}_${W}_kernel_stats.md
done
cd $ROOT
for W in flat10m k120 latency refshape pcie encoder pipeline kmeans; do timeout 400 python bench_extra.py $W; done > $OUT/${R}_extra_benchmarks.jsonl 2> $OUT/${R}_extra.err
timeout 400 python bench_extra.py ivfpq --rows 10000000 2>> $OUT/${R}_extra.err | tail -1 >> $OUT/${R}_extra_benchmarks.jsonl
tail -c 600 $OUT/${R}_bench_line.json; echo; cat $OUT/${R}_extra_benchmarks.jsonl | cut -c 1-400

R=r2; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/latency_probe.py > $OUT/${R}_single_query_line.json 2>$OUT/sq.err
rm -rf /tmp/psq; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psq -- python $ROOT/tools/latency_probe.py > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/psq "round 2 -- rocprofv3 --kernel-trace --stats of \`python tools/latency_probe.py\` (1M x 384, 5 % tombstones, one query per call through host pointers: 320 calls at k = 10 on the single-pass scan, 320 at k = 120 on the batch pipeline)" | head -20 > $OUT/${R}_single_query_kernel_stats.md
cat $OUT/${R}_single_query_line.json; head -16 $OUT/${R}_single_query_kernel_stats.md | cut -c1-160; tail -3 $OUT/sq.err

#!/bin/bash
# Round 3, after the INT8 layer work: the parts of profiles/ that depend on the INT8 encoder (bench line, INT8 kernel table, INT8 SQ counters).
# Run through gpurun; copy gpurun_out/r3_* into profiles/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/bench.py --steps 50 --warmup 5 > $OUT/r3_bench_line.json 2> $OUT/r3_bench.err
rm -rf /tmp/pe8; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe8 -- python $ROOT/tools/enc_bench.py int8 > $OUT/r3_encoder_int8_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe8 "round 3 -- rocprofv3 --kernel-trace --stats of \`python tools/enc_bench.py int8\` (MiniLM-L6 INT8, 4096 texts padded to 256 positions, lengths U[8,128]; 13 calls)" | head -24 > $OUT/r3_encoder_int8_kernel_stats.md
$ROOT/tools/r3_int8_pmc.sh > /dev/null 2>&1
cp $OUT/r3_int8_pmc.txt $OUT/r3_encoder_int8_pmc_sq.txt
tail -c 400 $OUT/r3_bench_line.json; echo; cat $OUT/r3_encoder_int8_line.json; head -16 $OUT/r3_encoder_int8_kernel_stats.md | cut -c1-160

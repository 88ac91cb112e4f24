#!/bin/bash
# profiles/r4_bigdim_*: kernel stats and counters of the 768 / 1024-dimension configs (bench.py --only-configs bigdim)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BD="python $ROOT/bench.py --steps 20 --warmup 3 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs bigdim"
rm -rf /tmp/pbd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbd -- $BD > $OUT/r4_bigdim_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pbd "round 4 -- rocprofv3 --kernel-trace --stats of \`bench.py --only-configs bigdim\` (1M rows at 768-d and 1024-d, 5 % tombstoned, batch 256, top-10): mfma_scan_big3_kernel<1, 48 | 64> = the emit scans, <0, .> the sample scans" | head -22 > $OUT/r4_bigdim_kernel_stats.md
rm -rf /tmp/pbd2; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pbd2 -- $BD > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pbd2 > $OUT/r4_bigdim_pmc.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pbd_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pbd_$C -- $BD > /dev/null 2>&1; python $ROOT/tools/pmc_summary.py /tmp/pbd_$C | grep big3 >> $OUT/r4_bigdim_pmc.txt 2>&1; done
# IVF-PQ kernel table on the final library
IV="python $ROOT/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
rm -rf /tmp/pi; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- $IV > $OUT/r4_ivfpq_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pi "round 4 -- rocprofv3 --kernel-trace --stats of \`bench.py --only-configs cfg4_ivfpq\` (10M rows, nlist 4096, nprobe 32, batch 1024 + the batch sweep 1 ... 4096)" | head -24 > $OUT/r4_ivfpq_kernel_stats.md
grep big3 $OUT/r4_bigdim_pmc.txt | cut -c1-220; sed -n 5,9p $OUT/r4_bigdim_kernel_stats.md | cut -c1-150

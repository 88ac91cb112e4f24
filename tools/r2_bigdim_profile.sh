#!/bin/bash
# round 2: evidence for mfma_scan_big_kernel (768 / 1024 dimensions): kernel stats + HBM traffic (FETCH_SIZE: does the second sub-pass hit L2?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs bigdim"
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- $CMD > $OUT/r2_bigdim_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pb "round 2 -- rocprofv3 --kernel-trace --stats of \`bench.py --only-configs bigdim\` (1M rows at 768-d and at 1024-d, batch 256; the exact-order comparison runs are in the table too)" | head -20 > $OUT/r2_bigdim_kernel_stats.md
: > $OUT/r2_bigdim_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pbm_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pbm_$C -- $CMD > /dev/null 2>&1; python $ROOT/tools/pmc_summary.py /tmp/pbm_$C | grep -E "mfma_scan_big|flat_exact" >> $OUT/r2_bigdim_pmc.txt; done
rm -rf /tmp/pbm_sq; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pbm_sq -- $CMD > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pbm_sq | grep -E "mfma_scan_big" >> $OUT/r2_bigdim_pmc.txt
cat $OUT/r2_bigdim_pmc.txt; head -12 $OUT/r2_bigdim_kernel_stats.md | cut -c1-150

#!/bin/bash
# Round 6: one GPU-box pass that regenerates what profiles/ holds (run through gpurun; outputs land in gpurun_out/, copy the
# summaries into profiles/). rocprofv3 PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
R=r6; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 50 --warmup 5"
LIGHT="--no-extra-configs --no-cpu-baseline --no-latency --sustained-s 0"
# 3. HBM traffic (separate PMC passes)
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python $ROOT/bench.py --steps 10 --warmup 2 --prewarm-ms 0 $LIGHT > /dev/null 2>&1; done
python $ROOT/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 1000000 384 256 10 > $OUT/pmc_traffic.json
cp $OUT/pmc_traffic.json $ROOT/profiles/pmc_traffic.json      # (the bench line below reads it: stamped with the hash of the scan source this run was built from)
# 1. the driver's command: the bounded contract line (stdout) and the full record (gpurun_out/bench_detail.json)
$BENCH > $OUT/${R}_bench_line.json 2> $OUT/${R}_bench.err; cp $OUT/bench_detail.json $OUT/${R}_bench_detail.json
# 2. kernel stats of the contract workload
rm -rf /tmp/ps; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- $BENCH $LIGHT > /dev/null 2>&1
cp $(find /tmp/ps -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
python $ROOT/tools/stats_to_md.py /tmp/ps "round 6 -- rocprofv3 --kernel-trace --stats of \`python bench.py --steps 50 --warmup 5 $LIGHT\` (1M x 384 f32 incl. 5 % tombstones, 256-query batches, top-10, 1 x MI355X)" > $OUT/${R}_bench_kernel_stats.md
# 4. SQ counters of the same workload: MFMA busy, LDS bank conflicts
rm -rf /tmp/pmc_sq; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -- python $ROOT/bench.py --steps 10 --warmup 2 --prewarm-ms 0 $LIGHT > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc_sq > $OUT/${R}_bench_pmc_sq.txt 2>&1
# 5. encoder (bf16, int8): kernel stats + SQ counters
for D in bf16 int8; do
  rm -rf /tmp/pe_$D; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$D -- python $ROOT/tools/enc_bench.py $D > $OUT/${R}_encoder_${D}_line.json 2>/dev/null
  python $ROOT/tools/stats_to_md.py /tmp/pe_$D "round 6 -- rocprofv3 --kernel-trace --stats of \`python tools/enc_bench.py $D\` (MiniLM-L6, 4096 texts, lengths U[8,128]; 13 calls)" | head -24 > $OUT/${R}_encoder_${D}_kernel_stats.md
done
rm -rf /tmp/pmc_enc; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_enc -- python $ROOT/tools/enc_bench.py bf16 > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc_enc > $OUT/${R}_encoder_bf16_pmc_sq.txt 2>&1
# 5b. k = 120 (recall's own k): kernels of one step, static bound vs the bound that tightens during the scan
$ROOT/tools/r6_k120_profile.sh > $OUT/${R}_k120_profile.txt 2>&1; cd /tmp
# 6. IVF-PQ at configs[3] size (10M): kernel stats + LDS conflict counters
IV="python $ROOT/bench.py --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs cfg4_ivfpq"
rm -rf /tmp/pi; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- $IV > /dev/null 2>&1; cp $OUT/bench_detail.json $OUT/${R}_ivfpq_line.json
python $ROOT/tools/stats_to_md.py /tmp/pi "round 6 -- rocprofv3 --kernel-trace --stats of \`bench.py --only-configs cfg4_ivfpq\` (10M rows, nlist 4096, nprobe 32, batch 1024)" | head -24 > $OUT/${R}_ivfpq_kernel_stats.md
rm -rf /tmp/pmc_iv; rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_iv -- $IV > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pmc_iv > $OUT/${R}_ivfpq_pmc_sq.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmci_$C; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmci_$C -- $IV > /dev/null 2>&1; python $ROOT/tools/pmc_summary.py /tmp/pmci_$C >> $OUT/${R}_ivfpq_pmc_sq.txt 2>&1; done
# 7. single-query path (solo_scan_kernel + final stage): host-pointer latency and kernel stats
python $ROOT/tools/latency_probe.py > $OUT/${R}_single_query_line.json 2>/dev/null
rm -rf /tmp/psq; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psq -- python $ROOT/tools/latency_probe.py > /dev/null 2>&1
python $ROOT/tools/stats_to_md.py /tmp/psq "round 6 -- rocprofv3 --kernel-trace --stats of \`python tools/latency_probe.py\` (1M x 384, 5 % tombstones, one query per call through host pointers: 320 calls at k = 10 on the single-pass scan, 320 at k = 120 on the batch pipeline)" | head -20 > $OUT/${R}_single_query_kernel_stats.md
ls -la $OUT | tail -20; tail -c 300 $OUT/${R}_bench_line.json; echo; cat $OUT/${R}_bench_pmc_sq.txt | head; cat $OUT/${R}_ivfpq_pmc_sq.txt | head -20
# 8. (round 5) the reference's call pattern: T callers x one query (bench.py concurrent_callers) -- the line, and the kernels a coalesced pass runs
python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --sustained-s 0 --only-configs concurrent_callers,concurrent_encode_callers > /dev/null 2>&1; cp $OUT/bench_detail.json $OUT/${R}_concurrent_line.json
rm -rf /tmp/pcc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pcc -- python $ROOT/tools/small_batch_probe.py 64 120 200 > $OUT/${R}_pass_nq64_k120.txt 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pcc "round 6 -- rocprofv3 --kernel-trace --stats of \`python tools/small_batch_probe.py 64 120 200\`: ONE host-pointer search of 64 queries, k = 120, on 1M x 384 (205 calls) = the pass 64 coalesced callers share" | head -22 > $OUT/${R}_pass_nq64_k120_kernel_stats.md
python $ROOT/tools/small_batch_probe.py > $OUT/${R}_small_batch.txt 2>/dev/null
# 9. INT8 encoder, per-text scope (4096 x encode()): kernel table and traffic per kernel
export SHODH_ENC_PER_TEXT=1
rm -rf /tmp/pe8; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe8 -- python $ROOT/tools/enc_bench.py int8 > $OUT/${R}_encoder_int8_pertext_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe8 "round 6 -- rocprofv3 --kernel-trace --stats of \`SHODH_ENC_PER_TEXT=1 python tools/enc_bench.py int8\` (MiniLM-L6 INT8, quant_scope PER_TEXT = 4096 x encode(): 4096 texts padded to 256 positions, lengths U[8,128]; 13 calls)" | head -24 > $OUT/${R}_encoder_int8_pertext_kernel_stats.md
: > $OUT/${R}_encoder_int8_pertext_pmc_sq.txt
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pi_$C; timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pi_$C -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1; python $ROOT/tools/pmc_summary.py /tmp/pi_$C | grep -E "i8_|qkv_attn|attn_out|act_quant|embed_ln" >> $OUT/${R}_encoder_int8_pertext_pmc_sq.txt; done
unset SHODH_ENC_PER_TEXT
# 10. the whole GPU suite and the smoke test, as the driver runs them
cd $ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/${R}_final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $OUT/${R}_final_gpu_tests.txt
cat $OUT/${R}_final_gpu_tests.txt; head -c 600 $OUT/${R}_concurrent_line.json; echo; cat $OUT/${R}_small_batch.txt | head -20

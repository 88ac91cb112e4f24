#!/bin/bash
# INT8 encoder: SQ counters per kernel (two --pmc passes; no tracing besides --kernel-trace). Output: gpurun_out/r3_int8_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/r3_int8_pmc.txt
rm -rf /tmp/pi_a; timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/pi_a -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pi_a | grep -E "i8_|qkv_attn|act_quant" >> $OUT/r3_int8_pmc.txt
rm -rf /tmp/pi_b; timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d /tmp/pi_b -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pi_b | grep -E "i8_|qkv_attn|act_quant" >> $OUT/r3_int8_pmc.txt
cat $OUT/r3_int8_pmc.txt
# HBM traffic per kernel (separate passes; gfx950: FETCH_SIZE counts half of a wide streaming read -> doubled below; units KB)
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pi_$C; timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pi_$C -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1; python $ROOT/tools/pmc_summary.py /tmp/pi_$C | grep -E "i8_|qkv_attn|act_quant|embed_ln" >> $OUT/r3_int8_pmc.txt; done
cat $OUT/r3_int8_pmc.txt | tail -16

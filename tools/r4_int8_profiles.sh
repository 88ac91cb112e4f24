#!/bin/bash
# Round 4: INT8 encoder in the per-text scope -- kernel tables (synthetic and export-style weights) and HBM traffic per kernel (separate --pmc passes).
# Run through gpurun; copy gpurun_out/r4_* into profiles/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SHODH_ENC_PER_TEXT=1
rm -rf /tmp/pe8; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe8 -- python $ROOT/tools/enc_bench.py int8 > $OUT/r4_encoder_int8_pertext_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe8 "round 4 -- rocprofv3 --kernel-trace --stats of \`SHODH_ENC_PER_TEXT=1 python tools/enc_bench.py int8\` (MiniLM-L6 INT8, quant_scope PER_TEXT = 4096 x encode(): 4096 texts padded to 256 positions, lengths U[8,128]; 13 calls)" | head -24 > $OUT/r4_encoder_int8_pertext_kernel_stats.md
rm -rf /tmp/pe9; SHODH_ENC_EXPORT=u8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe9 -- python $ROOT/tools/enc_bench.py int8 > $OUT/r4_encoder_int8_pertext_export_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe9 "round 4 -- the same on export-style weights (uint8 with a zero point per tensor: the kernels' zero-point variants)" | head -20 > $OUT/r4_encoder_int8_pertext_export_kernel_stats.md
: > $OUT/r4_encoder_int8_pertext_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pi_$C; timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pi_$C -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1; python $ROOT/tools/pmc_summary.py /tmp/pi_$C | grep -E "i8_|qkv_attn|attn_out|act_quant|embed_ln" >> $OUT/r4_encoder_int8_pertext_pmc.txt; done
rm -rf /tmp/pi_a; timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/pi_a -- python $ROOT/tools/enc_bench.py int8 > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py /tmp/pi_a | grep -E "i8_|qkv_attn|attn_out" >> $OUT/r4_encoder_int8_pertext_pmc.txt
cat $OUT/r4_encoder_int8_pertext_line.json $OUT/r4_encoder_int8_pertext_export_line.json; cut -c1-170 $OUT/r4_encoder_int8_pertext_kernel_stats.md | sed -n 5,13p; cut -c1-170 $OUT/r4_encoder_int8_pertext_export_kernel_stats.md | sed -n 5,12p; cat $OUT/r4_encoder_int8_pertext_pmc.txt

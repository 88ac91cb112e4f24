#!/bin/bash
# round 4, first GPU pass: per-text INT8 scope (v1: per-sequence ranges through the round-3 kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p1
timeout 900 python -m pytest tests/test_encoder_int8_pertext_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4p1/pertext_tests.txt
timeout 600 python -m pytest tests/test_encoder_int8_gpu.py tests/test_encoder_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4p1/int8_tests.txt
for m in 0 1; do SHODH_ENC_PER_TEXT=$m timeout 300 python tools/enc_bench.py int8 4096 2>&1 | tail -1; done > gpurun_out/r4p1/enc_bench.txt
cat gpurun_out/r4p1/*.txt

#!/bin/bash
# the whole GPU suite, as the driver runs it
cd ${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/r4_gpu_tests.txt
cat gpurun_out/r4_gpu_tests.txt

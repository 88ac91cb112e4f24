#!/bin/bash
# Round 6: a library variant (tools/build_variant.sh) through the flat-scan parity tests, then same-box step / emit-kernel times against the product: tools/r6_variant_check.sh <suffix> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r6fp; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for suf in "$@"; do
  echo "== parity, variant $suf" | tee -a $OUT/variant_check.txt
  SHODH_HIP_LIB=$ROOT/shodh_memory_amd/libshodh_hip.so.$suf SHODH_TRUST_PREBUILT=1 timeout 900 python -m pytest tests/test_flat_gpu.py tests/test_flat_fuzz_gpu.py tests/test_dynamic_threshold_gpu.py tests/test_single_query_gpu.py tests/test_probe_select_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed\|error" | tail -3 | tee -a $OUT/variant_check.txt
done
tools/r6_flat_variants.sh product "$@" 2>&1 | cut -c1-140

#!/bin/bash
# round 4, pass 3: fused tail (stage bit 7) vs v1 (SHODH_INT8_STAGES=0x6F) in per-text scope: bytes + time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r4p3; mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python - > $OUT/tail_vs_v1.txt 2>&1 <<'PY'
import os, numpy as np
import shodh_memory_amd as S
from shodh_memory_amd import _lib as L
from tests.test_encoder_int8_gpu import _batch
ids, mask = _batch(64, 31, 30522)
os.environ["SHODH_INT8_STAGES"] = "0x6F"
e1 = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8, quant_scope=L.QUANT_SCOPE_PER_TEXT)
a = e1.encode_ids(ids, mask)
os.environ["SHODH_INT8_STAGES"] = "0xEF"
e2 = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8, quant_scope=L.QUANT_SCOPE_PER_TEXT)
b = e2.encode_ids(ids, mask)
keep = mask.sum(1) > 0
c = (a[keep] * b[keep]).sum(1)
print("v1 vs fused tail: equal bytes", a.tobytes() == b.tobytes(), "min cos", c.min(), "max|diff|", np.abs(a - b).max())
print("nan", np.isnan(b).any(), "norms", np.linalg.norm(b[keep], axis=1)[:4])
PY
timeout 900 python -m pytest tests/test_encoder_int8_pertext_gpu.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pertext_tests.txt
cd /tmp
rm -rf /tmp/pe8; SHODH_ENC_PER_TEXT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe8 -- python $ROOT/tools/enc_bench.py int8 > $OUT/pertext_line.json 2>/dev/null
python $ROOT/tools/stats_to_md.py /tmp/pe8 "per-text, fused tail" | head -20 > $OUT/pertext_kernel_stats.md
timeout 300 python $ROOT/tools/enc_bench.py int8 > $OUT/batch_line.json 2>/dev/null
cat $OUT/tail_vs_v1.txt $OUT/pertext_tests.txt $OUT/pertext_line.json $OUT/batch_line.json; cut -c1-150 $OUT/pertext_kernel_stats.md

#!/bin/bash
# Round 6: the kernels of one k = 120 step (1M x 384, 256 queries), static bound vs the bound that tightens during the scan, sample strides 9 (k = 120 default) and 16
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "static_s9 SHODH_DYN_THR=0" "dyn_s9 SHODH_DYN_THR=1" "dyn_s16 SHODH_DYN_THR=1 SHODH_SAMPLE_STRIDE=16" "dyn_s12 SHODH_DYN_THR=1 SHODH_SAMPLE_STRIDE=12"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/pk_$name
  env "$@" K=120 ITERS=300 GRAFT_REPO_ROOT=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$name -- python $ROOT/tools/step_time.py > $OUT/k120_$name.txt 2>&1
  python $ROOT/tools/stats_to_md.py /tmp/pk_$name "round 6 -- k = 120, 1M x 384, 256 queries, $name: rocprofv3 --kernel-trace --stats of tools/step_time.py (K=120 ITERS=300)" | head -18 > $OUT/r6_k120_${name}_kernel_stats.md
  grep "^step" $OUT/k120_$name.txt | cut -c1-200
  sed -n 6,16p $OUT/r6_k120_${name}_kernel_stats.md | cut -c1-160
done

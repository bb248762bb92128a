#!/bin/bash
# MedFormer: per-launch-shape durations of the depthwise / pointwise kernels
T=${1:-r03_ac}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 3 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
for pat in "k_dwconv3_wgrad_lds" "k_dwconv3<" "k_conv_igemm<cbim::bf16_tag, 1, 2, 0, false" "k_conv_igemm<cbim::bf16_tag, 1, 2, 1, false" "k_conv_wgrad<" "k_mappool_bwd4" "k_norm_bwd_apply"; do
  echo "== $pat"; python $R/tools/rocpd_by_grid.py /tmp/pf_m/p_results.db "$pat" | head -14
done > $O/${T}_medformer_by_grid.txt 2>&1
cat $O/${T}_medformer_by_grid.txt

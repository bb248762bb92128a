#!/bin/bash
# k_conv3_r32 on 4x8x8 tiles with two workgroups per CU (CBIM_CONV_R32_TD=4) on the round-3 tree: per-kernel averages
T=${1:-r03_ah}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for td in 8 4; do
  rm -rf /tmp/pf_m
  CBIM_CONV_R32_TD=$td rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  echo "== TD=$td"; python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 | grep -E "k_conv3_r32|k_conv_igemm|dispatches"
done > $O/${T}_r32_td.txt 2>&1
cat $O/${T}_r32_td.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_v}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_medformer_kernels.txt 2>&1
for k in k_wgrad_r32 k_norm_bwd_apply k_conv3_rw k_conv_pw k_dwconv3_lds k_norm_act_fwd k_partial_sums k_pw_wgrad k_stats_finalize; do python $R/tools/rocpd_by_grid.py /tmp/pf_m/p_results.db $k | head -14; done > $O/${T}_medformer_by_grid.txt 2>&1
cd $R; python tools/aten_sources.py medformer 2>/dev/null | grep -v "Warn\|warn" > $O/${T}_aten_medformer.txt

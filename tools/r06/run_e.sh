#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_e}
cd /tmp; export TMPDIR=/tmp
for m in swin_unetr; do
rm -rf /tmp/pf_$m
rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
for k in k_conv_igemm k_conv_wgrad k_conv3_rw k_wgrad_r32 k_wgrad_reduce k_norm_act k_norm_bwd k_resnorm k_conv_pw k_pw_wgrad elementwise; do python $R/tools/rocpd_by_grid.py /tmp/pf_$m/p_results.db $k; done > $O/${T}_${m}_by_grid.txt 2>&1
done
cd $R; python tools/aten_sources.py swin_unetr > $O/${T}_aten_swin_unetr.txt 2>&1

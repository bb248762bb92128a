#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_c}
timeout 1200 python -m pytest tests -m gpu -x -q -k "swin or rw48 or conv_rw" > $O/${T}_gputest_swin.log 2>&1; tail -5 $O/${T}_gputest_swin.log
for v in 0 1; do
  CBIM_CONV_RW48=$v python bench.py --model swin_unetr --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('CBIM_CONV_RW48=$v swin_unetr ms/step', round(d['ms_per_step'], 3), d['config'].get('graph'))"
done | tee $O/${T}_swin_ab.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_swin
rocprofv3 --kernel-trace --stats -d /tmp/pf_swin -o p -- python $R/bench.py --model swin_unetr --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_swin/p_results.db 7 > $O/${T}_swin_unetr_kernels.txt 2>&1
head -30 $O/${T}_swin_unetr_kernels.txt
for k in k_conv_igemm k_conv_wgrad k_conv3_rw k_wgrad_r32 k_norm_act k_norm_bwd; do python $R/tools/rocpd_by_grid.py /tmp/pf_swin/p_results.db $k; done > $O/${T}_swin_unetr_by_grid.txt 2>&1

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_u}
timeout 1500 python -m pytest tests -m gpu -x -q -k "upcat_skip or wgrad_r32 or swin" > $O/${T}_gputest.log 2>&1; tail -3 $O/${T}_gputest.log
for rep in 1 2; do
python bench.py --model swin_unetr --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('swin_unetr ms/step', round(d['ms_per_step'], 3))"
done | tee $O/${T}_steps.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_s
rocprofv3 --kernel-trace --stats -d /tmp/pf_s -o p -- python $R/bench.py --model swin_unetr --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_s/p_results.db 7 > $O/${T}_swin_unetr_kernels.txt 2>&1
head -24 $O/${T}_swin_unetr_kernels.txt
cd $R; python tools/aten_sources.py swin_unetr 2>/dev/null | grep -v "Warn\|warn" | head -12

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_d}
python tools/r06/conv48_ab.py 10 > $O/${T}_conv48_ab.txt 2>&1; cat $O/${T}_conv48_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "wgrad_r32 or rw48" > $O/${T}_gputest_wgrad.log 2>&1; tail -3 $O/${T}_gputest_wgrad.log
for v in 0 1; do
  CBIM_WGRAD_R32_C16=$v python bench.py --model swin_unetr --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('CBIM_WGRAD_R32_C16=$v swin_unetr ms/step', round(d['ms_per_step'], 3), d['config'].get('graph'))"
done | tee $O/${T}_swin_ab.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_y}
timeout 1500 python -m pytest tests -m gpu -q -k "norm_branches or aug or dwconv or depthwise or medformer or optim" > $O/${T}_gputest.log 2>&1; tail -15 $O/${T}_gputest.log
timeout 1200 python tools/r06/norm_envelope.py 2>&1 | grep -v "Warn\|warn\|bf16 envelope" | tee $O/${T}_norm_envelope.txt
for rep in 1 2; do
python bench.py --model medformer --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('medformer ms/step', round(d['ms_per_step'], 3))"
done | tee $O/${T}_steps.txt
python tools/aten_sources.py medformer 2>/dev/null | grep -v "Warn\|warn" | head -30 | tee $O/${T}_aten_medformer.txt

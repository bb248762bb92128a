#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -k "medformer" > $O/r06_vit_gputest.log 2>&1; tail -6 $O/r06_vit_gputest.log | cut -c1-300
for rep in 1 2; do
python bench.py --model medformer --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('medformer ms/step', round(d['ms_per_step'], 3))"
done | tee $O/r06_vit_steps.txt
timeout 900 python tools/bench_shipped_config.py acdc/medformer_3d.yaml lits/medformer_3d.yaml --graph 1 --steps 10 --warmup 3 2>&1 | grep -v "Warn\|warn\|amdgpu" | tee -a $O/r06_vit_steps.txt
python tools/aten_sources.py medformer 2>/dev/null | grep -v "Warn\|warn" | head -24 | tee $O/r06_vit_aten_medformer.txt

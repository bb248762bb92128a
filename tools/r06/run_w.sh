#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_w}
timeout 1500 python -m pytest tests -m gpu -x -q -k "se_gate or trilinear or medformer or map_branch" > $O/${T}_gputest.log 2>&1; tail -3 $O/${T}_gputest.log
for rep in 1 2; do
python bench.py --model medformer --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('medformer ms/step', round(d['ms_per_step'], 3))"
done | tee $O/${T}_steps.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_medformer_kernels.txt 2>&1
head -3 $O/${T}_medformer_kernels.txt; grep "k_se_\|trilinear\|k_map_gemm" $O/${T}_medformer_kernels.txt
cd $R; python tools/aten_sources.py medformer 2>/dev/null | grep -v "Warn\|warn" | head -14

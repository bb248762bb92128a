#!/bin/bash
# weight gradients on a side stream (CBIM_WGRAD_STREAM=1), joined at the end of backward: step A/B, graph and eager
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_k}
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in 0 1 0 1; do
  CBIM_WGRAD_STREAM=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet wgrad_stream=$v ms/step" | tee -a $O/${T}_bench_ab.txt
done
for v in 0 1; do
  CBIM_WGRAD_STREAM=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline --graph 0 2>>$O/${T}_bench.err | ms "resunet eager wgrad_stream=$v ms/step" | tee -a $O/${T}_bench_ab.txt
done
tail -5 $O/${T}_bench.err

#!/bin/bash
T=${1:-r03_ad}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "dwconv or depthwise" 2>&1 | tail -2
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
timeout 300 python bench.py --model medformer --no-cpu-baseline --no-roofline | ms "medformer ms/step"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 3 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
for pat in "k_dwconv3_wgrad_lds" "k_dwconv3<" "k_dwconv3_wgrad<"; do
  echo "== $pat"; python $R/tools/rocpd_by_grid.py /tmp/pf_m/p_results.db "$pat" | head -14
done > $O/${T}_medformer_by_grid.txt 2>&1
cat $O/${T}_medformer_by_grid.txt

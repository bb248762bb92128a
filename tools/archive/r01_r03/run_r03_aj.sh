#!/bin/bash
T=${1:-r03_aj}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -x -q -k "upcat or trilinear or resunet or golden or envelope" 2>&1 | tail -3
for f in 0 1; do
  echo "== CBIM_UP_GRAM=$f"
  CBIM_UP_GRAM=$f timeout 600 python tools/stream_bench.py --only up_stats 2>&1 | grep -E "up_"
done | tee $O/${T}_stream_up.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for f in 0 1 0 1; do
  CBIM_UP_GRAM=$f timeout 300 python bench.py --no-cpu-baseline --no-roofline | ms "resunet up-gram=$f ms/step"
done

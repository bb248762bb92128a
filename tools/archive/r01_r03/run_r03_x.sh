#!/bin/bash
# matrix-core head backward: tests, streaming-kernel table (with both head kernels), A/B on the ResUNet / MedFormer step
T=${1:-r03_x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "head or stem" 2>&1 | tail -3
timeout 600 python tools/stream_bench.py --json $O/${T}_stream.json 2>&1 | tee $O/${T}_stream.txt | tail -30
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for m in resunet medformer; do
  timeout 300 python bench.py --model $m --no-cpu-baseline --no-roofline | ms "$m ms/step"
done

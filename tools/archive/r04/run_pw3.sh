#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pointwise" 2>&1 | tail -2
python tools/r04/igemm_floor.py 2>&1 | grep "us/launch" | grep "k(1, 1, 1)" | cut -c1-75
for m in medformer swin_unetr; do
python bench.py --model $m --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['ms_per_step'],3), 'ms')"
done

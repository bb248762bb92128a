#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pointwise" 2>&1 | tail -2
for v in 0 1; do
  echo "== CBIM_PW_KS=$v"
  CBIM_PW_KS=$v python tools/r04/igemm_floor.py 2>&1 | grep "us/launch" | grep "k(1, 1, 1)" | cut -c1-75
  CBIM_PW_KS=$v python bench.py --model medformer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('medformer', round(d['ms_per_step'],3), 'ms')"
done | tee gpurun_out/r04_y_pw_ks.txt

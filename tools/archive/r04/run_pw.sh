#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pointwise" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
for pw in 0 1; do
  rm -rf /tmp/pfl
  CBIM_CONV_PW=$pw rocprofv3 --kernel-trace --stats -d /tmp/pfl -o p -- python $R/tools/r04/igemm_floor.py > /dev/null 2>&1
  echo "== CBIM_CONV_PW=$pw"; python $R/tools/rocpd_by_grid.py /tmp/pfl/p_results.db k_conv 2>&1 | grep -v "^columns" | head -12
done | tee $R/gpurun_out/r04_y_pw_floor.txt
cd $R
for pw in 0 1; do
  CBIM_CONV_PW=$pw python bench.py --model medformer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('medformer CBIM_CONV_PW=$pw', round(d['ms_per_step'],3), 'ms', d['config']['final_loss'])"
done | tee -a gpurun_out/r04_y_pw_floor.txt

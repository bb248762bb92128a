#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "dwconv" 2>&1 | tail -2
for v in 0 1; do
  CBIM_DWCONV_LDS=$v python bench.py --model medformer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('medformer CBIM_DWCONV_LDS=$v', round(d['ms_per_step'],3), 'ms', d['config']['final_loss'])"
done | tee gpurun_out/r04_zz_dwconv_ab.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmf
rocprofv3 --kernel-trace --stats -d /tmp/pmf -o p -- python $R/bench.py --model medformer --steps 4 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pmf/p_results.db k_dwconv3 2>&1 | cut -c1-140 | head -30 | tee -a $R/gpurun_out/r04_zz_dwconv_ab.txt

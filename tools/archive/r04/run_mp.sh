#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_u_late_gpu_cases.py -q -x -m gpu -k "mappool" 2>&1 | tail -2
python bench.py --model medformer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('medformer', round(d['ms_per_step'],3), 'ms', d['config']['final_loss'])"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmf
rocprofv3 --kernel-trace --stats -d /tmp/pmf -o p -- python $R/bench.py --model medformer --steps 4 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pmf/p_results.db k_mappool 2>&1 | cut -c1-140 | head -12

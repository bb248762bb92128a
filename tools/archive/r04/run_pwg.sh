#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pointwise" 2>&1 | tail -2
for m in medformer swin_unetr; do
python bench.py --model $m --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['ms_per_step'],3), 'ms', d['config']['final_loss'])"
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmf
rocprofv3 --kernel-trace --stats -d /tmp/pmf -o p -- python $R/bench.py --model medformer --steps 4 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pmf/p_results.db 6 2>&1 | grep "wgrad\|dispatches" | cut -c1-130 | head -12

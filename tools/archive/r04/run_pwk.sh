#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pointwise" 2>&1 | tail -2
for v in 0 1; do
for m in medformer; do
CBIM_PW_KSPLIT=$v python bench.py --model $m --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m CBIM_PW_KSPLIT=$v', round(d['ms_per_step'],3), 'ms', d['config']['final_loss'])"
done; done | tee gpurun_out/r04_zzz_pw_ksplit.txt
python bench.py --model swin_unetr --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin', round(d['ms_per_step'],3), 'ms')"

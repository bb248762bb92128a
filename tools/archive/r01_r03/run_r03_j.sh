#!/bin/bash
T=${1:-r03_r}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -x -q -k "layernorm or swin" 2>&1 | tail -3
for v in 1 0; do
  CBIM_SWIN_FUSED_LN=$v timeout 300 python bench.py --model swin_unetr --no-cpu-baseline --no-roofline | python -c "import sys,json; print('swin CBIM_SWIN_FUSED_LN=$v ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_s
rocprofv3 --kernel-trace --stats -d /tmp/pf_s -o p -- python $R/bench.py --model swin_unetr --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_s/p_results.db 7 > $O/${T}_swin_unetr_kernels.txt 2>&1
head -45 $O/${T}_swin_unetr_kernels.txt

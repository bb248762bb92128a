#!/bin/bash
T=${1:-r03_ag}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -x -q -k "dwconv or medformer" 2>&1 | tail -2
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for f in 0 1 0 1; do
  CBIM_DW_WGRAD_MFMA=$f timeout 300 python bench.py --model medformer --no-cpu-baseline --no-roofline | ms "medformer dw-wgrad-mfma=$f ms/step"
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_medformer_kernels.txt 2>&1
head -24 $O/${T}_medformer_kernels.txt | cut -c1-150

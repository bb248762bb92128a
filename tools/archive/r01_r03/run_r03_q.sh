#!/bin/bash
T=${1:-r03_q}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -x -q -k "dwconv or medformer or depthwise" 2>&1 | tail -3
timeout 300 python bench.py --model medformer --no-cpu-baseline --no-roofline | python -c "import sys,json; print('medformer ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_medformer_kernels.txt 2>&1
head -28 $O/${T}_medformer_kernels.txt

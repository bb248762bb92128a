#!/bin/bash
# separable transposed interpolation (cbim_lin_adjoint_axis) against the one-pass gather: tests, A/B per model, kernel table
T=${1:-r03_w}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "upcat or trilinear" 2>&1 | tail -3
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for m in resunet medformer; do
  for sep in 0 1 0 1; do
    CBIM_UP_SEPARABLE=$sep timeout 300 python bench.py --model $m --no-cpu-baseline --no-roofline | ms "$m separable=$sep ms/step"
  done
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model resunet --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_resunet_kernels.txt 2>&1
grep -n "lin_adjoint\|upcat\|up_tile" $O/${T}_resunet_kernels.txt | head

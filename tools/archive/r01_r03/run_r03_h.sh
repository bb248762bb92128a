#!/bin/bash
T=${1:-r03_h}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "upcat" 2>&1 | tail -2
for v in "CBIM_UP_TILES=1" "CBIM_UP_TILES=0" "CBIM_FUSED_UP=0"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline | python -c "import sys,json; print('resunet $v ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_e
rocprofv3 --kernel-trace --stats -d /tmp/pf_e -o p -- python $R/bench.py --steps 3 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pf_e/p_results.db "k_up" 2>&1 | tee $O/${T}_up_by_level.txt

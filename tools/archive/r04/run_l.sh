#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_l}
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_rw or activated or conv_r32" > $O/${T}_pytest.txt 2>&1; tail -2 $O/${T}_pytest.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in "0 0" "1 1" "1 0" "1 1"; do set -- $v
  CBIM_CONV_RW=$1 CBIM_CONV_RW_WIDE=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet rw=$1 wide=$2 ms/step" | tee -a $O/${T}_bench_ab.txt
done

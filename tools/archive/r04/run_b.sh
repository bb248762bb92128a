#!/bin/bash
# round 4, second GPU call: k_conv3_rw — parity on the GPU, per-layer A/B, step A/B
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_rw or activated or conv_r32" > $O/r04_b_pytest.txt 2>&1; tail -5 $O/r04_b_pytest.txt
timeout 600 python tools/r04/conv_rw_ab.py 10 > $O/r04_b_conv_rw_ab.txt 2>&1; cat $O/r04_b_conv_rw_ab.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in "0 0" "1 0" "1 1" "0 0" "1 1"; do set -- $v
  CBIM_CONV_RW=$1 CBIM_CONV_RW_WIDE=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/r04_b_bench.err | ms "resunet rw=$1 wide=$2 ms/step" | tee -a $O/r04_b_bench_ab.txt
done

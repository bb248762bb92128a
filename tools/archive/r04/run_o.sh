#!/bin/bash
# low-resolution layers: k_conv3_rw split-K against k_conv_igemm's, k_wgrad_r32 strips: parity + step A/B
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_o}
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_rw or activated or conv_r32 or wgrad" > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do set -- $v
  CBIM_CONV_RW_SPLIT=$1 CBIM_WGRAD_R32_SMALL_STRIPS=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet rw_split=$1 small_strips=$2 ms/step" | tee -a $O/${T}_bench_ab.txt
done
CB_SHAPES=256x256x16,128x512x16,576x512x16,320x320x8,256x640x8 timeout 300 python tools/conv_ab.py 10 2>&1 | grep -v amdgpu > $O/${T}_conv_ab_lowres.txt; cat $O/${T}_conv_ab_lowres.txt

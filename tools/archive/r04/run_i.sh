#!/bin/bash
# ping-pong schedule of k_conv3_rw: parity, step A/B, phase cycle profile
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_i}
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_rw or activated or conv_r32" > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in 0 1 0 1 0 1; do
  CBIM_CONV_RW_PP=$v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet pp=$v ms/step" | tee -a $O/${T}_bench_ab.txt
done
cd $R/cbim-medical-image-segmentation_amd/csrc && touch conv_rw.hip && make EXTRA=-DCBIM_RW_PROF 2>&1 | tail -1
cd $R
for v in 0 1; do CBIM_CONV_RW_PP=$v CB_SHAPES=32x32x128 python tools/r04/prof_rw.py 2>&1 | grep -v amdgpu.ids | grep -A1 "wide=0" | grep -v "^--$" ; done > $O/${T}_rw_prof.txt 2>&1
cat $O/${T}_rw_prof.txt

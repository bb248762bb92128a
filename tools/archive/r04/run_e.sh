#!/bin/bash
# parity of k_conv3_rw on the GPU, step A/B, then the phase cycle profile (prof build on the box's copy only)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_e}
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_rw or activated or conv_r32" > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do set -- $v
  CBIM_CONV_RW=$1 CBIM_CONV_RW_WIDE=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet rw=$1 wide=$2 ms/step" | tee -a $O/${T}_bench_ab.txt
done
cd $R/cbim-medical-image-segmentation_amd/csrc && touch conv_rw.hip && make EXTRA=-DCBIM_RW_PROF 2>&1 | tail -1
cd $R
CB_SHAPES=${CB_SHAPES:-32x32x128,96x64x128} python tools/r04/prof_rw.py 2> $O/${T}_rw_prof.txt
grep -v amdgpu.ids $O/${T}_rw_prof.txt

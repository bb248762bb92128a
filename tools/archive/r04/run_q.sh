#!/bin/bash
# k_wgrad_r32 with buffer-addressed plane-major pieces: parity, step A/B against the previous source (the box's copy only), phase profile
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_q}
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or dwconv" > $O/${T}_pytest.txt 2>&1; tail -3 $O/${T}_pytest.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet new wgrad ms/step" | tee -a $O/${T}_bench_ab.txt; done
CB_SHAPES=32x32x128,96x64x128,64x64x64,192x128x64,128x128x32,256x256x16 timeout 300 python tools/conv_bench.py bf16 10 wgrad 2>&1 | grep -v amdgpu | tee $O/${T}_wgrad_new.txt
C=$R/cbim-medical-image-segmentation_amd/csrc
cp $C/conv_wgrad_r32.hip /tmp/new.hip; cp $R/tools/r04/old/conv_wgrad_r32.hip.txt $C/conv_wgrad_r32.hip; (cd $C && make 2>&1 | tail -1)
for v in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet old wgrad ms/step" | tee -a $O/${T}_bench_ab.txt; done
CB_SHAPES=32x32x128,96x64x128,64x64x64,192x128x64,128x128x32,256x256x16 timeout 300 python tools/conv_bench.py bf16 10 wgrad 2>&1 | grep -v amdgpu | tee $O/${T}_wgrad_old.txt
cp /tmp/new.hip $C/conv_wgrad_r32.hip; (cd $C && make EXTRA=-DCBIM_WR32_PROF 2>&1 | tail -1)
CB_SHAPES=32x32x128,96x64x128 python tools/r04/prof_wr32.py 2> $O/${T}_wr32_prof.txt; grep "rep 1" -A1 $O/${T}_wr32_prof.txt | grep prof

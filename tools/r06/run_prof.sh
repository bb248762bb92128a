#!/bin/bash
# phase cycle profile of k_conv3_rw48: builds the library with -DCBIM_RW_PROF on the GPU box (the box's copy only)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r06_g}
cd $R/cbim-medical-image-segmentation_amd/csrc && touch conv_rw.hip && make EXTRA=-DCBIM_RW_PROF 2>&1 | tail -1
cd $R
python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 --secondary 0 > /dev/null 2>&1
python tools/r06/prof_rw48.py 2> $O/${T}_rw48_prof.txt
grep -v amdgpu.ids $O/${T}_rw48_prof.txt

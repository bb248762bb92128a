#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_p}
cd $R/cbim-medical-image-segmentation_amd/csrc && touch conv_wgrad_r32.hip && make EXTRA=-DCBIM_WR32_PROF 2>&1 | tail -1
cd $R
python bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > /dev/null 2>&1
python tools/r04/prof_wr32.py 2> $O/${T}_wr32_prof.txt
grep -v amdgpu.ids $O/${T}_wr32_prof.txt

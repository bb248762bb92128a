#!/bin/bash
# compile-time ablations of k_wgrad_r32:   gpurun -- bash tools/run_wr32_ablate.sh [tag]
T=${1:-r03_n}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R/cbim-medical-image-segmentation_amd/csrc && touch conv_wgrad_r32.hip && make EXTRA=-DCBIM_WR32_ABLATE 2>&1 | tail -2
cd $R
python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 3 > /dev/null 2>&1   # clocks up
python tools/wr32_ablate.py 32x32x128 > $O/${T}_wr32_ablate.txt 2>&1
python tools/wr32_ablate.py 96x64x128 >> $O/${T}_wr32_ablate.txt 2>&1
python tools/wr32_ablate.py 256x256x16 >> $O/${T}_wr32_ablate.txt 2>&1
cat $O/${T}_wr32_ablate.txt

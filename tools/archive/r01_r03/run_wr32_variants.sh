#!/bin/bash
# builds k_wgrad_r32 with alternative tuning knobs and times each on two layer shapes:  gpurun -- bash tools/run_wr32_variants.sh [tag]
T=${1:-r03_o}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 3 > /dev/null 2>&1   # clocks up
for V in "" "-DWR32_RING=7" "-DWR32_RING=9" "-DWR32_SETPRIO=1" "-DWR32_DMA_STEPS=1" "-DWR32_RING=7 -DWR32_SETPRIO=1" "-DWR32_RING=7 -DWR32_DMA_STEPS=1 -DWR32_SETPRIO=1"; do
  (cd cbim-medical-image-segmentation_amd/csrc && touch conv_wgrad_r32.hip && make EXTRA="$V" 2>&1 | grep -E "error|spill" )
  echo "## variant [$V]"
  for sh in 32x32x128 96x64x128 192x128x64; do echo -n "$sh: "; WR_ONLY0=1 python tools/wr32_ablate.py $sh 2>&1 | grep waves | awk '{printf "%s %s  ", $1, $4}'; echo; done
done 2>&1 | tee $O/${T}_wr32_variants.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 3 > /dev/null 2>&1
for mv in "" 0; do
  echo "## CB_R32_MINVOX=[$mv]"
  CB_R32_MINVOX=$mv CB_SHAPES=128x512x16,256x256x16,576x512x16,256x640x8,320x320x8,128x128x32,64x256x32 python tools/conv_ab.py 20 2>&1 | grep "\->" | awk '{print $1,$2,$3,"fwdR",$8,"dgA",$12}'
done 2>&1 | tee $O/r03_t_small_layers_r32.txt

#!/bin/bash
# HBM-side traffic of k_conv3_rw / k_wgrad_r32 on the 128^3 layers (separate --pmc passes; FETCH_SIZE x2 per the gfx950 note)
R=$GRAFT_REPO_ROOT; T=${1:-r04_f}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  CB_SHAPES=${CB_SHAPES:-32x32x128,96x64x128} rocprofv3 --kernel-trace --pmc $P -d /tmp/pm$i -o p -- python $R/tools/r04/conv_rw_ab.py 3 > /dev/null 2>&1
  for K in "k_conv3_rw<false, 1, false" "k_conv3_rw<true, 1, false" "k_conv3_r32<0, false, false, 8, false" "k_conv3_r32<1, false, true, 8, false" "k_conv3_rw<false, 1, true" "k_conv3_rw<false, 2, true" "k_conv3_rw<true, 1, true" "k_conv3_r32<0, false, false, 8, true" "k_conv3_r32<1, false, true, 8, true"; do echo "## $K"; python $R/tools/pmc_query.py /tmp/pm$i/p_results.db "$K" 30; done
done 2>&1 | tee $O/${T}_pmc_rw.txt

#!/bin/bash
# round 4: kernel-level durations of k_conv3_r32 / k_conv3_rw on the big layers (rocprofv3), then the step A/B
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_c}
python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 3 > /dev/null 2>&1   # clocks up, page-in
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_c
CB_SHAPES=${CB_SHAPES:-32x32x128,96x64x128,64x64x64,192x128x64,128x128x32,384x256x32} rocprofv3 --kernel-trace --stats -d /tmp/pf_c -o p -- python $R/tools/r04/conv_rw_ab.py 5 > $O/${T}_ab_under_prof.txt 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pf_c/p_results.db k_conv3_r > $O/${T}_conv_by_grid.txt 2>&1
cat $O/${T}_conv_by_grid.txt
cd $R
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in "0 0" "1 0" "1 1" "0 0" "1 0"; do set -- $v
  CBIM_CONV_RW=$1 CBIM_CONV_RW_WIDE=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet rw=$1 wide=$2 ms/step" | tee -a $O/${T}_bench_ab.txt
done

#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_g}
python bench.py --no-cpu-baseline --no-roofline --steps 5 --warmup 3 > /dev/null 2>&1
CB_SHAPES=96x64x128,192x128x64,384x256x32,64x64x64 timeout 600 python tools/r04/conv_rw_ab.py 10 > $O/${T}_conv_rw_ab.txt 2>&1; cat $O/${T}_conv_rw_ab.txt
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in "0 0" "1 0" "1 1" "0 0" "1 1"; do set -- $v
  CBIM_CONV_RW=$1 CBIM_CONV_RW_WIDE=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>>$O/${T}_bench.err | ms "resunet rw=$1 wide=$2 ms/step" | tee -a $O/${T}_bench_ab.txt
done

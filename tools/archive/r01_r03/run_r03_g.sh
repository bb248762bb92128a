#!/bin/bash
T=${1:-r03_g}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
rm -rf /tmp/pf_e
CBIM_UP_TILES=$v rocprofv3 --kernel-trace --stats -d /tmp/pf_e -o p -- python $R/bench.py --steps 3 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
echo "== CBIM_UP_TILES=$v"
python $R/tools/rocpd_by_grid.py /tmp/pf_e/p_results.db "k_up" 
done 2>&1 | tee $O/${T}_up_by_level.txt

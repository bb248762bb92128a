#!/bin/bash
# replayed-graph kernel table of the ResUNet step + per-launch-shape durations of every kernel (eager trace) + PMC fetch of the dgrad
T=${1:-r04_h}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_graph /tmp/pf_eager
rocprofv3 --kernel-trace --stats -d /tmp/pf_graph -o p -- python $R/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_graph/p_results.db 13 > $O/${T}_resunet_graph_kernels.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pf_eager -o p -- python $R/bench.py --steps 4 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pf_eager/p_results.db k_ > $O/${T}_resunet_by_grid.txt 2>&1
rm -rf /tmp/pm1
CB_SHAPES=96x64x128,192x128x64 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm1 -o p -- python $R/tools/r04/conv_rw_ab.py 3 > /dev/null 2>&1
for K in "k_conv3_rw<false, 1, true" "k_conv3_rw<false, 2, true" "k_conv3_rw<true, 1, true" "k_conv3_rw<true, 2, true" "k_conv3_r32<1, false, true, 8, true"; do echo "## $K"; python $R/tools/pmc_query.py /tmp/pm1/p_results.db "$K" 30; done > $O/${T}_pmc_fetch.txt 2>&1
head -c 400 $O/${T}_resunet_bench.json; echo
head -50 $O/${T}_resunet_graph_kernels.txt; cat $O/${T}_pmc_fetch.txt

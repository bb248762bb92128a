#!/bin/bash
# ResUNet bench JSON + rocprofv3 kernel table (gpurun -- bash tools/run_profile_resunet.sh <tag>)
T=${1:-r01_i}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python $R/bench.py > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_r
rocprofv3 --kernel-trace --stats -d /tmp/pf_r -o p -- python $R/bench.py --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_r/p_results.db 7 > $O/${T}_resunet_kernels.txt 2>&1
head -c 600 $O/${T}_resunet_bench.json; echo; head -32 $O/${T}_resunet_kernels.txt

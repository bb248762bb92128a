#!/bin/bash
# MedFormer bench JSON + rocprofv3 kernel table (gpurun -- bash tools/run_profile_medformer.sh <tag>)
T=${1:-r01_h}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python $R/bench.py --model medformer --cpu-size 64 > $O/${T}_medformer_bench.json 2> $O/${T}_medformer_bench.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_medformer_kernels.txt 2>&1
head -c 420 $O/${T}_medformer_bench.json; echo; head -40 $O/${T}_medformer_kernels.txt

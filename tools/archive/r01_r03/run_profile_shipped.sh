#!/bin/bash
# rocprofv3 kernel table of one shipped configuration's training step
# (gpurun -- bash tools/run_profile_shipped.sh <tag> <config> [steps]), e.g. lits/medformer_3d.yaml
T=${1:-r01_l}; C=${2:-lits/medformer_3d.yaml}; S=${3:-2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
N=$(echo $C | tr '/.' '__')
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_s
rocprofv3 --kernel-trace --stats -d /tmp/pf_s -o p -- python $R/tools/bench_shipped_config.py $C --steps $S --warmup 1 > $O/${T}_${N}_bench.log 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_s/p_results.db $((S + 1)) > $O/${T}_${N}_kernels.txt 2>&1
head -30 $O/${T}_${N}_kernels.txt

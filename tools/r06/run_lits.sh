#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in lits acdc; do
rm -rf /tmp/pf_$c
rocprofv3 --kernel-trace --stats -d /tmp/pf_$c -o p -- python $R/tools/bench_shipped_config.py $c/medformer_3d.yaml --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_$c/p_results.db 7 > $O/r06_${c}_medformer_kernels.txt 2>&1
head -28 $O/r06_${c}_medformer_kernels.txt | cut -c1-150
done

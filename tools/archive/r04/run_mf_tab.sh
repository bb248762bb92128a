#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmf
rocprofv3 --kernel-trace --stats -d /tmp/pmf -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pmf/p_results.db 7 > $R/gpurun_out/r04_zzz_medformer_kernels.txt 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pmf/p_results.db k_ > $R/gpurun_out/r04_zzz_medformer_by_grid.txt 2>&1
head -45 $R/gpurun_out/r04_zzz_medformer_kernels.txt | cut -c1-130

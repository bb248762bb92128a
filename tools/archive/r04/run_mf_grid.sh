#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmf
rocprofv3 --kernel-trace --stats -d /tmp/pmf -o p -- python $R/bench.py --model medformer --steps 4 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pmf/p_results.db k_ > $R/gpurun_out/r04_z_medformer_by_grid.txt 2>&1
grep "k_dwconv3\|k_conv_wgrad\|k_conv_pw" $R/gpurun_out/r04_z_medformer_by_grid.txt | cut -c1-150 | head -60

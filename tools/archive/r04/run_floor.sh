#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
python tools/r04/igemm_floor.py 2>&1 | grep "us/launch" | tee gpurun_out/r04_x_igemm_floor.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pfl
rocprofv3 --kernel-trace --stats -d /tmp/pfl -o p -- python $R/tools/r04/igemm_floor.py > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pfl/p_results.db k_ 2>&1 | head -30 | tee -a $R/gpurun_out/r04_x_igemm_floor.txt

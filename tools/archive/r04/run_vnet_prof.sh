#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/vprof && rocprofv3 --kernel-trace --stats -d /tmp/vprof -o p -- python $R/tools/r04/vnet_time.py 2 > $R/gpurun_out/r04_vnet_time.txt 2>&1
python $R/tools/rocpd_summary.py /tmp/vprof/p_results.db 13 > $R/gpurun_out/r04_vnet_kernels.txt 2>&1
python $R/tools/rocpd_by_grid.py /tmp/vprof/p_results.db k_ > $R/gpurun_out/r04_vnet_by_grid.txt 2>&1
grep "ms/step" $R/gpurun_out/r04_vnet_time.txt
cd $R && python tools/r04/vnet_time.py 2 graph 2>&1 | tail -4 | tee -a gpurun_out/r04_vnet_time.txt
python -m pytest tests/test_u_late_gpu_cases.py -q -x -m gpu -k "vnet" 2>&1 | tail -3
cut -c1-160 $R/gpurun_out/r04_vnet_kernels.txt | head -32

#!/bin/bash
# VNet on the GPU: parity tests, the ops touched by the VNet change, step time + kernel table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_u_late_gpu_cases.py -q -x -m gpu -k "vnet" -s 2>&1 | tail -25 > gpurun_out/r04_vnet_tests.txt
python -m pytest tests/test_gpu_ops.py tests/test_shipped_configs.py -q -x -m gpu 2>&1 | tail -8 >> gpurun_out/r04_vnet_tests.txt
python tools/r04/vnet_time.py 2 > gpurun_out/r04_vnet_time.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/vprof -o vnet -- python "$OLDPWD/tools/r04/vnet_time.py" 2 > /dev/null 2>&1)
f=$(find /tmp/vprof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > gpurun_out/r04_vnet_kernel_stats.csv
cat gpurun_out/r04_vnet_tests.txt gpurun_out/r04_vnet_time.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -k "bn or vnet or norm_branches or batchnorm or affine" > $O/r06_bn_gputest.log 2>&1; tail -5 $O/r06_bn_gputest.log | cut -c1-300
timeout 900 python tools/bench_shipped_config.py acdc/vnet_3d.yaml --graph 1 --steps 10 --warmup 3 2>&1 | grep -v "Warn\|warn\|amdgpu" | tee $O/r06_bn_steps.txt
timeout 900 python tools/bench_shipped_config.py acdc/vnet_3d.yaml --graph 0 --steps 10 --warmup 3 2>&1 | grep -v "Warn\|warn\|amdgpu" | tee -a $O/r06_bn_steps.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_v
rocprofv3 --kernel-trace --stats -d /tmp/pf_v -o p -- python $R/tools/bench_shipped_config.py acdc/vnet_3d.yaml --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_v/p_results.db 7 > $O/r06_bn_vnet_kernels.txt 2>&1
head -22 $O/r06_bn_vnet_kernels.txt | cut -c1-150

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -k "one_wide_head or gemm_attention or attn" > $O/r06_awg2_gputest.log 2>&1; tail -8 $O/r06_awg2_gputest.log | cut -c1-300
( timeout 900 python tools/bench_shipped_config.py lits/medformer_3d.yaml acdc/medformer_3d.yaml --graph 1 --steps 10 --warmup 3 2>&1
  timeout 900 python tools/bench_shipped_config.py acdc/medformer_3d.yaml --graph 0 --steps 10 --warmup 3 2>&1 ) | grep -v "Warn\|warn\|amdgpu" | tee $O/r06_awg2_steps.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_a
rocprofv3 --kernel-trace --stats -d /tmp/pf_a -o p -- python $R/tools/bench_shipped_config.py acdc/medformer_3d.yaml --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_a/p_results.db 7 > $O/r06_awg2_acdc_medformer_kernels.txt 2>&1
head -30 $O/r06_awg2_acdc_medformer_kernels.txt | cut -c1-150

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "one_wide_head or lits" > $O/r06_awg_gputest.log 2>&1; tail -12 $O/r06_awg_gputest.log | cut -c1-300
grep "gemm attention vs\|bf16 envelope" $O/r06_awg_gputest.log | cut -c1-400
timeout 600 python tools/bench_shipped_config.py lits/medformer_3d.yaml acdc/medformer_3d.yaml 2>&1 | grep -v "Warn\|warn\|amdgpu" | tee $O/r06_awg_steps.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_l
rocprofv3 --kernel-trace --stats -d /tmp/pf_l -o p -- python $R/tools/bench_shipped_config.py lits/medformer_3d.yaml --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_l/p_results.db 7 > $O/r06_awg_lits_medformer_kernels.txt 2>&1
head -24 $O/r06_awg_lits_medformer_kernels.txt | cut -c1-150

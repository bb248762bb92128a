#!/bin/bash
# Swin window attention on the matrix cores: GPU correctness, bench.py A/B lines and the per-kernel table.
#   gpurun --timeout 900 -- bash tools/run_swin_check.sh [tag]
T=${1:-r02_i}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -x -q -k "window or swin or head" > $O/${T}_swin_tests.log 2>&1; tail -3 $O/${T}_swin_tests.log
python bench.py --model swin_unetr --no-cpu-baseline > $O/${T}_swin_unetr_bench.json 2> $O/${T}_swin_unetr_bench.err; head -c 400 $O/${T}_swin_unetr_bench.json; echo
CBIM_WINATTN_MFMA_BWD=0 python bench.py --model swin_unetr --no-cpu-baseline --no-roofline 2>/dev/null | head -c 200; echo " <- MFMA_BWD=0"
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_swin
rocprofv3 --kernel-trace --stats -d /tmp/pf_swin -o p -- python $R/bench.py --model swin_unetr --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_swin/p_results.db 7 > $O/${T}_swin_unetr_kernels.txt 2>&1
head -30 $O/${T}_swin_unetr_kernels.txt

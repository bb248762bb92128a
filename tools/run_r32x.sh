#!/bin/bash
# generalised conv_r32 (channel multiples of 32): correctness, per-layer table with the default selection, ResUNet step
T=${1:-r02_n}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "r32 or conv" > $O/${T}_r32_tests.log 2>&1; tail -3 $O/${T}_r32_tests.log
python tools/conv_bench.py bf16 10 fwd,dgrad 2>&1 | grep -v amdgpu.ids | tee $O/${T}_conv_bench.txt
python bench.py --no-cpu-baseline > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err; head -c 330 $O/${T}_resunet_bench.json; echo
CBIM_CONV_R32=1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | head -c 330; echo " <- CBIM_CONV_R32=1"

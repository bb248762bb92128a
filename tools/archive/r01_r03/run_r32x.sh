#!/bin/bash
# generalised conv_r32 (channel multiples of 32): correctness + per-layer A/B against k_conv_igemm (forced both ways)
T=${1:-r02_p}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "r32" > $O/${T}_r32_tests.log 2>&1; tail -3 $O/${T}_r32_tests.log
export CB_SHAPES=32x32x128,96x64x128,64x96x128,32x128x64,128x32x64,64x64x64,192x128x64,128x192x64,128x128x32,384x256x32,256x384x32
(CBIM_CONV_R32=1 python tools/conv_bench.py bf16 10 fwd,dgrad; CB_R32_MINVOX=0 python tools/conv_bench.py bf16 10 fwd,dgrad) 2>&1 | grep -v amdgpu.ids | tee $O/${T}_conv_bench_ab.txt

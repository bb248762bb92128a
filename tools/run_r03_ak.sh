#!/bin/bash
# k_conv3_r32: half of the waves run their epilogue before the tile barrier (R32_SKEW), epilogue operands requested inside the
# MFMA loop (R32_RQ_STEP); libcbim_hip.so = HEAD
T=${1:-r03_ak}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
P=$R/cbim-medical-image-segmentation_amd
echo "== tests with s1q6"
CBIM_HIP_LIBRARY=$P/libcbim_hip_s1q6.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_headline_parity.py -m gpu -x -q -k "conv or fused or headline" 2>&1 | tail -2
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for lib in libcbim_hip.so libcbim_hip_s0q6.so libcbim_hip_s1q6.so libcbim_hip_s1q7.so libcbim_hip.so libcbim_hip_s0q6.so libcbim_hip_s1q6.so libcbim_hip_s1q7.so; do
  CBIM_HIP_LIBRARY=$P/$lib timeout 300 python bench.py --no-cpu-baseline --no-roofline | ms "resunet $lib ms/step"
done 2>&1 | tee $O/${T}_skew.txt

#!/bin/bash
# same-box A/B of the matrix-core stem / head kernels on the ResUNet and MedFormer steps
R=$GRAFT_REPO_ROOT; cd $R
ms() { python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for m in resunet medformer; do
  for f in 0 1 0 1; do
    CBIM_STEM_MFMA=$f CBIM_HEAD_MFMA=$f timeout 300 python bench.py --model $m --no-cpu-baseline --no-roofline | ms "$m stem/head mfma=$f ms/step"
  done
done

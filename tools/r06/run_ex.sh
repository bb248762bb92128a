#!/bin/bash
# the README's command list on the box: examples + shipped-config step times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 600 python examples/train_synthetic.py --config amos_ct/resunet_3d.yaml --iters 10 --epochs 1 2>&1 | tail -6
  timeout 600 python examples/train_synthetic.py --config amos_ct/medformer_3d.yaml --iters 6 --epochs 1 2>&1 | tail -4
  timeout 900 python tools/bench_shipped_config.py acdc/medformer_3d.yaml lits/medformer_3d.yaml bcv/medformer_3d.yaml acdc/vnet_3d.yaml acdc/unet++_3d.yaml amos_ct/attention_unet_3d.yaml kits/swin_unetr_3d.yaml 2>&1 | tail -12 ) | grep -v "Warn\|warn\|amdgpu" | tee $O/r06_ex.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout 900 python tools/bench_shipped_config.py lits/medformer_3d.yaml acdc/medformer_3d.yaml bcv/medformer_3d.yaml kits/medformer_3d.yaml amos_ct/medformer_3d.yaml --graph 1 --steps 10 --warmup 3 2>&1
  timeout 900 python tools/bench_shipped_config.py lits/medformer_3d.yaml acdc/medformer_3d.yaml --graph 0 --steps 10 --warmup 3 2>&1 ) | grep -v "Warn\|warn\|amdgpu" | tee $O/r06_g_shipped_medformer.txt

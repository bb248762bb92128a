#!/bin/bash
# per-call extras of run_profile_r06.sh (this call: VNet / ACDC / LiTS step times on the final tree)
cd $GRAFT_REPO_ROOT
timeout 900 python tools/bench_shipped_config.py acdc/vnet_3d.yaml acdc/medformer_3d.yaml lits/medformer_3d.yaml acdc/unet++_3d.yaml --graph 1 --steps 10 --warmup 3 2>&1 | grep -v "Warn\|warn\|amdgpu"

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -k "1x3x3 or acdc or aniso or shipped_training_size" > $O/r06_embed_gputest.log 2>&1; tail -5 $O/r06_embed_gputest.log | cut -c1-300
timeout 1500 python tools/r06/embed_ab.py 2>&1 | grep -v "Warn\|warn\|amdgpu" | tee $O/r06_embed_ab.txt

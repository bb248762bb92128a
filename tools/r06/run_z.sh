#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python tools/r06/norm_envelope.py 2>&1 | grep -v "Warn\|warn\|bf16 envelope" | tee $O/r06_z_norm_envelope.txt

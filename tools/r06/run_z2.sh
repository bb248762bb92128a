#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -k "norm_branches" > $O/r06_z2_gputest.log 2>&1; tail -12 $O/r06_z2_gputest.log
grep "bf16 envelope over" $O/r06_z2_gputest.log

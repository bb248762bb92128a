#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_f}
timeout 2400 python -m pytest tests -m gpu -q -k "envelope or unetpp or vnet_bf16 or bcv" > $O/${T}_gputest_envelope.log 2>&1; tail -40 $O/${T}_gputest_envelope.log

#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_n}
timeout 2400 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_parity.py tests/test_gpu_benchmarked_sizes.py -m gpu -q -s -k "not trained" > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
grep -E "grad |engine vs|inside the bf16|passed|failed|^E  " $O/${T}_gputest.log | head -120

#!/bin/bash
# the full GPU suite + smoke (parity record -> gpurun_out/r04_parity.json)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04_m}
rm -f $O/r04_parity.json
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
tail -30 $O/${T}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -3 $O/${T}_smoke.log

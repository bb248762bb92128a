#!/bin/bash
# full -m gpu suite + smoke + the driver's bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_p}
rm -f $O/r06_parity.json
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
tail -25 $O/${T}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -3 $O/${T}_smoke.log
( time python bench.py ) > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -c 1500 $O/${T}_bench.json; tail -4 $O/${T}_bench.err

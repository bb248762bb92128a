#!/bin/bash
# every shipped 3-D yaml of the in-scope models at its own training size, step replayed from one hipGraph
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
CFGS=$(python -c "import json; print(' '.join(sorted(json.load(open('tests/golden/shipped_configs.json')))))")
for c in $CFGS; do
  timeout 300 python tools/bench_shipped_config.py $c --graph 1 --steps 10 --warmup 3 2>&1 | grep -v "Warn\|warn\|amdgpu" | tail -1
done | tee $O/r06_all24_shipped_configs.txt

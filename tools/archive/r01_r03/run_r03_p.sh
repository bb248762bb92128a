#!/bin/bash
T=${1:-r03_p}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for m in resunet medformer swin_unetr; do
timeout 300 python bench.py --model $m --no-cpu-baseline > $O/${T}_${m}_bench.json 2> $O/${T}_${m}_bench.err
python - <<PY
import json
d = json.loads(open("$O/${T}_${m}_bench.json").read().strip().splitlines()[-1])
print("$m ms/step", round(d["ms_per_step"], 3), {k: (v["launches_per_step"], round(v["avg_launch_ms"] * 1e3, 1), round(v["frac_of_peak"], 3)) for k, v in d["roofline"]["kernels"].items()})
PY
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_graph
rocprofv3 --kernel-trace --stats -d /tmp/pf_graph -o p -- python $R/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_graph/p_results.db 13 > $O/${T}_resunet_graph_kernels.txt 2>&1
head -14 $O/${T}_resunet_graph_kernels.txt

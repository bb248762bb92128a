#!/bin/bash
# Round-3 first GPU call: full GPU suite, the A/B table of tools/conv_ab.py, bench lines for the four corners of
# (CBIM_MATERIALIZE, CBIM_WGRAD_R32).   gpurun -- bash tools/run_r03_first.sh [tag]
T=${1:-r03_a}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
tail -5 $O/${T}_gputest.log
timeout 600 python tools/conv_ab.py 10 > $O/${T}_conv_ab.txt 2>&1
cat $O/${T}_conv_ab.txt
for m in 1 0; do for w in 1 0; do
  CBIM_MATERIALIZE=$m CBIM_WGRAD_R32=$w timeout 300 python bench.py --no-cpu-baseline > $O/${T}_bench_m${m}_w${w}.json 2> $O/${T}_bench_m${m}_w${w}.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/${T}_bench_m${m}_w${w}.json").read().strip().splitlines()[-1])
    print("materialize=$m wgrad_r32=$w  ms/step", round(d["ms_per_step"], 3), {k: (v["launches_per_step"], round(v["avg_launch_ms"] * 1e3, 1), round(v["frac_of_peak"], 3)) for k, v in d["roofline"]["kernels"].items()})
except Exception as e:
    print("bench m=$m w=$w failed", e)
PY
done; done
CBIM_WGRAD_R32_WAVES=4 timeout 300 python bench.py --no-cpu-baseline > $O/${T}_bench_m1_w1_waves4.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/${T}_bench_m1_w1_waves4.json').read().strip().splitlines()[-1]); print('waves4 ms/step', d['ms_per_step'])"

#!/bin/bash
# Round-3: clean-build A/B table (ablation switches compiled out), bench line, kernel table of one eager + one replayed step
T=${1:-r03_c}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2>&1   # clocks up
CB_SHAPES=${CB_SHAPES:-32x32x128,96x64x128,64x64x64,192x128x64,128x128x32,384x256x32,256x256x16,576x512x16,320x320x8} timeout 600 python tools/conv_ab.py 10 > $O/${T}_conv_ab.txt 2>&1
cat $O/${T}_conv_ab.txt
timeout 300 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<PY
import json
d = json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), {k: (v["launches_per_step"], round(v["avg_launch_ms"] * 1e3, 1), round(v["frac_of_peak"], 3)) for k, v in d["roofline"]["kernels"].items()})
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_e /tmp/pf_g
rocprofv3 --kernel-trace --stats -d /tmp/pf_e -o p -- python $R/bench.py --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_e/p_results.db 7 > $O/${T}_resunet_kernels.txt 2>&1
head -45 $O/${T}_resunet_kernels.txt

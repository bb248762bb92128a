#!/bin/bash
# Round-3: GPU suite + bench lines of the three models + kernel tables of MedFormer / SwinUNETR
T=${1:-r03_d}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
tail -4 $O/${T}_gputest.log
for m in resunet medformer swin_unetr; do
  timeout 400 python bench.py --model $m --no-cpu-baseline > $O/${T}_${m}_bench.json 2> $O/${T}_${m}_bench.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/${T}_${m}_bench.json").read().strip().splitlines()[-1])
    print("$m ms/step", round(d["ms_per_step"], 3), {k: (v["launches_per_step"], round(v["avg_launch_ms"] * 1e3, 1), round(v["frac_of_peak"], 3)) for k, v in d["roofline"]["kernels"].items()})
except Exception as e:
    print("bench $m failed", e); print(open("$O/${T}_${m}_bench.err").read()[-1500:])
PY
done
CBIM_FUSED_UP=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline | python -c "import sys,json; print('resunet CBIM_FUSED_UP=0 ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-roofline --aug 1 > $O/${T}_resunet_aug_bench.json 2> $O/${T}_resunet_aug_bench.err; python -c "import json; d=json.loads(open('$O/${T}_resunet_aug_bench.json').read().strip().splitlines()[-1]); print('aug ms/step', d['ms_per_step'], d['config']['workload'][-60:])"
cd /tmp; export TMPDIR=/tmp
for m in resunet medformer swin_unetr; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
  head -32 $O/${T}_${m}_kernels.txt
done

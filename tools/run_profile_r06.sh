#!/bin/bash
# Round-6 measurements.  gpurun --timeout 2400 -- bash tools/run_profile_r06.sh <tag> [suite] [pmc] [models]
#   suite : the full -m gpu test-suite + smoke() first
#   pmc   : HBM-traffic counter passes (FETCH_SIZE / WRITE_SIZE, separate runs) -> tools/pmc_to_traffic_r06.py
#   models: rocprofv3 kernel tables of MedFormer / SwinUNETR (eager) next to the ResUNet's replayed-graph table
T=${1:-r06_a}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
if [[ " $* " == *" suite "* ]]; then
  rm -f $O/r06_parity.json
  timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
  tail -6 $O/${T}_gputest.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -3 $O/${T}_smoke.log
fi
# the driver's command: headline + roofline + cpu_baseline + secondary (MedFormer, SwinUNETR, --aug 1 in the same process)
( time python $R/bench.py ) > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err
head -c 600 $O/${T}_resunet_bench.json; echo; tail -4 $O/${T}_resunet_bench.err
python - <<PY
import json
try:
    d = json.loads(open("$O/${T}_resunet_bench.json").read().strip().splitlines()[-1])
    print("headline ms", round(d["ms_per_step"], 3), "| secondary:", {k: (round(v.get("ms_per_step", -1), 2), v.get("error", ""))  for k, v in d.get("secondary", {}).items()})
    print("roofline:", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "step_frac_mfma", "traffic_ratio", "step_traffic_ratio")})
    print("cpu_baseline:", d.get("cpu_baseline"))
except Exception as e:
    print("bench json unreadable:", e)
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_graph
rocprofv3 --kernel-trace --stats -d /tmp/pf_graph -o p -- python $R/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_graph/p_results.db 13 > $O/${T}_resunet_graph_kernels.txt 2>&1
head -32 $O/${T}_resunet_graph_kernels.txt
if [[ " $* " == *" models "* ]]; then
  for m in medformer swin_unetr; do
    rm -rf /tmp/pf_$m
    rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
    python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
    head -24 $O/${T}_${m}_kernels.txt
  done
fi
if [[ " $* " == *" pmc "* ]]; then
  for m in resunet medformer swin_unetr; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/p_$c
      rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o f -- python $R/bench.py --model $m --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
      python $R/tools/pmc_dump.py /tmp/p_$c/f_results.db $c $O/${T}_pmc_${m}_$c.json
    done
    [[ " $* " == *" pmc1 "* ]] && break
  done
  cd $R && python tools/pmc_to_traffic_r06.py $T 3 gpurun_out && cp profiles/r06_traffic*.json $O/
fi
# per-call A/B lines (written for the call at hand, removed afterwards)
if [ -f $R/tools/r06_extra.sh ]; then cd $R; bash $R/tools/r06_extra.sh $T 2>&1 | tee $O/${T}_extra.txt; fi

#!/bin/bash
# run-to-run spread of the driver's command on ONE box (six fresh processes) + one 200-step line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('run $i: ms/step', round(d['ms_per_step'], 3), 'volumes/s', round(d['value'], 2), d['roofline']['kernel'], 'frac', round(d['roofline']['frac'], 4), 'avg launch us', round(d['roofline']['avg_launch_ms'] * 1e3, 1))"
done | tee $O/r06_final_run_to_run.txt
python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('200 steps: ms/step', round(d['ms_per_step'], 3), 'volumes/s', round(d['value'], 2))" | tee -a $O/r06_final_run_to_run.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r06_last_bench.json 2> $O/r06_last_bench.err; tail -3 $O/r06_last_bench.err

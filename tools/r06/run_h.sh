#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_h}
timeout 1500 python -m pytest tests -m gpu -x -q -k "map_branch or medformer" > $O/${T}_gputest_medformer.log 2>&1; tail -5 $O/${T}_gputest_medformer.log
for v in 0 1; do
  CBIM_MAP_KERNELS=$v python bench.py --model medformer --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('CBIM_MAP_KERNELS=$v medformer ms/step', round(d['ms_per_step'], 3), d['config'].get('graph'))"
done | tee $O/${T}_medformer_ab.txt
python tools/aten_sources.py medformer > $O/${T}_aten_medformer.txt 2>&1; grep -v "Warning\|warn" $O/${T}_aten_medformer.txt | head -30
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_m
rocprofv3 --kernel-trace --stats -d /tmp/pf_m -o p -- python $R/bench.py --model medformer --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_m/p_results.db 7 > $O/${T}_medformer_kernels.txt 2>&1
head -45 $O/${T}_medformer_kernels.txt

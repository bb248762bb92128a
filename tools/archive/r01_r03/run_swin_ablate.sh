#!/bin/bash
T=${1:-r02_j}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
tools/ubench/_bin/lds_atomic > $O/${T}_lds_atomic.txt 2>&1; cat $O/${T}_lds_atomic.txt
cd /tmp; export TMPDIR=/tmp
for dbg in 0 1 2 4; do
  rm -rf /tmp/pf_swin
  CBIM_WM_DBG=$dbg rocprofv3 --kernel-trace --stats -d /tmp/pf_swin -o p -- python $R/bench.py --model swin_unetr --steps 3 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  echo "== CBIM_WM_DBG=$dbg"; python $R/tools/rocpd_summary.py /tmp/pf_swin/p_results.db 4 2>&1 | grep "winattn"
done | tee $O/${T}_swin_bwd_ablate.txt

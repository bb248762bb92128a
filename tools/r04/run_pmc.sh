#!/bin/bash
# HBM-side traffic per kernel family of the ResUNet step only (FETCH_SIZE / WRITE_SIZE in separate passes): tools/r04/run_pmc.sh <tag>
T=${1:-r04_v}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PATS=("false>(cbim::R32Params)" "true>(cbim::R32Params)" "k_wgrad_r32<" "k_norm_bwd_apply<" "k_norm_act_fwd<" "k_up_tile<" "k_splitk_finish<")
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  for pat in "${PATS[@]}"; do echo "== $c $pat"; python $R/tools/pmc_query.py /tmp/p_$c/f_results.db "$pat"; done
done > $O/${T}_pmc_hbm_resunet.txt 2>&1
cat $O/${T}_pmc_hbm_resunet.txt

#!/bin/bash
# what bounds k_up_tile? SQ counters in two passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pu_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pu_$i -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  for pat in "k_up_tile<cbim::bf16_tag, 1>" "k_up_tile<cbim::bf16_tag, 2>" "k_norm_act_fwd<"; do echo "== $pat"; python $R/tools/pmc_query.py /tmp/pu_$i/f_results.db "$pat" 100; done
done > $O/r04_w_pmc_up_tile.txt 2>&1
cat $O/r04_w_pmc_up_tile.txt

#!/bin/bash
# PMC counters of the tiled up-path kernels (two passes)
T=${1:-r03_ab}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES"; do
  rm -rf /tmp/pu
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pu -o f -- python $R/tools/stream_bench.py --only up --iters 3 > /dev/null 2>&1
  for pat in "k_up_tile<cbim::bf16_tag, 0>" "k_up_tile<cbim::bf16_tag, 1>" "k_up_tile<cbim::bf16_tag, 2>" "k_lin_adjoint"; do echo "== $pat"; python $R/tools/pmc_query.py /tmp/pu/f_results.db "$pat" 30; done
done > $O/${T}_pmc_up.txt 2>&1
cat $O/${T}_pmc_up.txt

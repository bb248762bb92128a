#!/bin/bash
# PMC passes over the 32->32 @128^3 layer on k_conv3_r32 (forward = transformed input, dgrad = LDS-DMA + mask):
#   gpurun -- bash tools/run_pmc_r32.sh [tag]
R=$GRAFT_REPO_ROOT; T=${1:-r02}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
P4="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  CB_SHAPES=32x32x128 rocprofv3 --kernel-trace --pmc $P -d /tmp/pm$i -o p -- python $R/tools/conv_bench.py bf16 3 fwd,dgrad > /dev/null 2>&1
  for K in "k_conv3_r32<1, true" "k_conv3_r32<1, false"; do echo "## $K"; python $R/tools/pmc_query.py /tmp/pm$i/p_results.db "$K" 30; done
done 2>&1 | tee $O/${T}_pmc_r32.txt

#!/bin/bash
# PMC passes over one conv shape: bash tools/run_pmc_conv.sh 32x32x128 fwd "pattern"
R=$GRAFT_REPO_ROOT; SH=${1:-32x32x128}; W=${2:-fwd}; PAT=${3:-k_conv_igemm}
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  CB_SHAPES=$SH rocprofv3 --kernel-trace --pmc $P -d /tmp/pm$i -o p -- python $R/tools/conv_bench.py bf16 3 $W > /dev/null 2>&1
  python $R/tools/pmc_query.py /tmp/pm$i/p_results.db "$PAT" 30
done

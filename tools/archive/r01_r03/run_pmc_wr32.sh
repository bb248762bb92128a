#!/bin/bash
# PMC passes over k_wgrad_r32 and k_conv3_r32 on the 32->32 @128^3 layer + the ablation tables.
#   gpurun -- bash tools/run_pmc_wr32.sh [tag]
R=$GRAFT_REPO_ROOT; T=${1:-r03_b}; O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/wr32_ablate.py 32x32x128 > $O/${T}_wr32_ablate.txt 2>&1
python tools/wr32_ablate.py 96x64x128 >> $O/${T}_wr32_ablate.txt 2>&1
cat $O/${T}_wr32_ablate.txt
python tools/r32_ablate.py 128 > $O/${T}_r32_ablate.txt 2>&1
cat $O/${T}_r32_ablate.txt
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
P4="SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  CB_SHAPES=32x32x128 timeout 300 rocprofv3 --kernel-trace --pmc $P -d /tmp/pm$i -o p -- python $R/tools/conv_ab.py 2 > /dev/null 2>&1
  for K in "k_wgrad_r32<8" "k_wgrad_r32<4" "k_conv3_r32<0, false, false" "k_conv3_r32<1, false, true"; do echo "## $K"; python $R/tools/pmc_query.py /tmp/pm$i/p_results.db "$K" 30; done
done 2>&1 | tee $O/${T}_pmc_wr32.txt

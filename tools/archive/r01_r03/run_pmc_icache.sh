#!/bin/bash
# Instruction-cache and wait counters of the conv kernels on one layer shape: gpurun -- bash tools/run_pmc_icache.sh [CinxCoutxS]
R=$GRAFT_REPO_ROOT; SH=${1:-192x64x64}; cd /tmp; export TMPDIR=/tmp
P1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
rm -rf /tmp/pmi
CB_SHAPES=$SH rocprofv3 --kernel-trace --pmc $P1 -d /tmp/pmi -o p -- python $R/tools/conv_bench.py bf16 3 fwd > /tmp/pmi.log 2>&1
python $R/tools/pmc_query.py /tmp/pmi/p_results.db k_conv_igemm 30

R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CACHE[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*" | sort -u | head -60
P1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
rm -rf /tmp/pmi
CB_SHAPES=192x64x64 rocprofv3 --kernel-trace --pmc $P1 -d /tmp/pmi -o p -- python $R/tools/conv_bench.py bf16 3 fwd > /tmp/pmi.log 2>&1
tail -3 /tmp/pmi.log
python $R/tools/pmc_query.py /tmp/pmi/p_results.db k_conv_igemm 30

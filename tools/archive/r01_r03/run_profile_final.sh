#!/bin/bash
# Round-1 final measurements: bench JSONs, rocprofv3 kernel tables for the three models and the HBM PMC passes of
# the ResUNet conv kernels.   gpurun -- bash tools/run_profile_final.sh [tag]
T=${1:-r01_j}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python $R/bench.py > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err
python $R/bench.py --model medformer --cpu-size 64 > $O/${T}_medformer_bench.json 2> $O/${T}_medformer_bench.err
python $R/bench.py --model swin_unetr --cpu-size 64 > $O/${T}_swin_bench.json 2> $O/${T}_swin_bench.err
python $R/bench.py --aug 1 --no-cpu-baseline > $O/${T}_resunet_aug_bench.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
for m in resunet medformer swin_unetr; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  for pat in "k_conv_igemm<cbim::bf16_tag, 2, 1," "k_conv_igemm<cbim::bf16_tag, 2, 2," "k_conv_igemm<cbim::bf16_tag, 1, 2," "k_conv_wgrad<"; do
    echo "== $c $pat"; python $R/tools/pmc_query.py /tmp/p_$c/f_results.db "$pat"
  done
done > $O/${T}_pmc_hbm.txt 2>&1
for f in resunet medformer swin resunet_aug; do head -c 420 $O/${T}_${f}_bench.json; echo; done
cat $O/${T}_pmc_hbm.txt

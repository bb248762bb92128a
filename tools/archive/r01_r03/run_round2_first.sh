#!/bin/bash
# FIRST GPU call of the next round: times the changes made after round 1's last GPU minute (explicit fmaf in the Swin
# window attention / depthwise convs / generic attention, tiled online softmax, even-odd histograms, k_mappool_bwd4,
# attn_wide LDS sizing + 4-code trips) against the committed r01_k / r01_l numbers:
#   MedFormer 48.3-49.8 ms, SwinUNETR 44.2-44.8 ms, ResUNet 18.4-19.5 ms (bench.py, hipGraph);
#   ACDC MedFormer 45.2 ms, LiTS MedFormer 80.5 ms (tools/bench_shipped_config.py, eager).
#   gpurun --timeout 600 -- bash tools/run_round2_first.sh [tag]
T=${1:-r02_a}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for m in resunet medformer swin_unetr; do
  python $R/bench.py --model $m --no-cpu-baseline > $O/${T}_${m}_bench.json 2> $O/${T}_${m}_bench.err
  head -c 300 $O/${T}_${m}_bench.json; echo
done
python $R/tools/bench_shipped_config.py acdc/medformer_3d.yaml lits/medformer_3d.yaml bcv/medformer_3d.yaml bcv/swin_unetr_3d.yaml \
  --steps 5 --warmup 2 2>/dev/null | grep config > $O/${T}_shipped_bench.txt
CBIM_MAPPOOL_BWD4=0 python $R/bench.py --model medformer --no-cpu-baseline --no-roofline 2>/dev/null | head -c 300 > $O/${T}_medformer_mappool_old.json
cat $O/${T}_shipped_bench.txt; cat $O/${T}_medformer_mappool_old.json; echo
cd /tmp; export TMPDIR=/tmp
for m in medformer swin_unetr; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
  head -14 $O/${T}_${m}_kernels.txt
done

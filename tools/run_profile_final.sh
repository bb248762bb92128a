#!/bin/bash
# Round-1 final measurements: bench JSONs + rocprofv3 kernel tables for the three models (gpurun -- bash tools/run_profile_final.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python $R/bench.py > $O/r01_f_resunet_bench.json 2> $O/r01_f_resunet_bench.err
python $R/bench.py --model medformer --cpu-size 64 > $O/r01_f_medformer_bench.json 2> $O/r01_f_medformer_bench.err
python $R/bench.py --model swin_unetr --cpu-size 64 > $O/r01_f_swin_bench.json 2> $O/r01_f_swin_bench.err
python $R/bench.py --aug 1 --no-cpu-baseline > $O/r01_f_resunet_aug_bench.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
for m in resunet medformer swin_unetr; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/r01_f_${m}_kernels.txt 2>&1
done
for f in resunet medformer swin resunet_aug; do head -c 420 $O/r01_f_${f}_bench.json; echo; done

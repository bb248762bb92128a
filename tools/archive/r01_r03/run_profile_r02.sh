#!/bin/bash
# Round-2 measurements: bench JSONs, rocprofv3 kernel tables for the three models, and PMC passes (HBM traffic; MFMA busy)
# over the ResUNet conv kernels.   gpurun --timeout 1500 -- bash tools/run_profile_r02.sh [tag]
T=${1:-r02_z}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python $R/bench.py > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err
python $R/bench.py --model medformer --cpu-size 64 > $O/${T}_medformer_bench.json 2> $O/${T}_medformer_bench.err
python $R/bench.py --model swin_unetr --cpu-size 64 > $O/${T}_swin_bench.json 2> $O/${T}_swin_bench.err
cd /tmp; export TMPDIR=/tmp
for m in resunet medformer swin_unetr; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
done
# the default command under the graph (what bench.py's value is measured on): kernel table of the replayed step
rm -rf /tmp/pf_graph
rocprofv3 --kernel-trace --stats -d /tmp/pf_graph -o p -- python $R/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_graph/p_results.db 13 > $O/${T}_resunet_graph_kernels.txt 2>&1
PATS=("k_conv3_r32<" "k_conv_igemm<cbim::bf16_tag, 2, 2," "k_conv_igemm<cbim::bf16_tag, 1, 2," "k_conv_wgrad<")
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  for pat in "${PATS[@]}"; do echo "== $c $pat"; python $R/tools/pmc_query.py /tmp/p_$c/f_results.db "$pat"; done
done > $O/${T}_pmc_hbm.txt 2>&1
rm -rf /tmp/p_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/p_mfma -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
for pat in "${PATS[@]}"; do echo "== $pat"; python $R/tools/pmc_query.py /tmp/p_mfma/f_results.db "$pat" 30; done > $O/${T}_pmc_mfma.txt 2>&1
for f in resunet medformer swin; do head -c 420 $O/${T}_${f}_bench.json; echo; done
cat $O/${T}_pmc_hbm.txt | head -40; cat $O/${T}_pmc_mfma.txt | head -50

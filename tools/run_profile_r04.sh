#!/bin/bash
# Round-4 measurements: bench JSONs of the three models (+ --aug 1), rocprofv3 kernel tables (replayed hipGraph step, eager by
# launch shape), PMC passes (HBM traffic per conv kernel family; MFMA busy), streaming-kernel bandwidth table.
#   gpurun --timeout 2400 -- bash tools/run_profile_r04.sh [tag] [suite]
T=${1:-r04_s}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$2" = "suite" ]; then
  rm -f $O/r04_parity.json
  timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/${T}_gputest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_gputest.log
  tail -4 $O/${T}_gputest.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -3 $O/${T}_smoke.log
fi
python $R/bench.py > $O/${T}_resunet_bench.json 2> $O/${T}_resunet_bench.err
python $R/bench.py --model medformer --cpu-size 64 > $O/${T}_medformer_bench.json 2> $O/${T}_medformer_bench.err
python $R/bench.py --model swin_unetr --cpu-size 64 > $O/${T}_swin_bench.json 2> $O/${T}_swin_bench.err
python $R/bench.py --aug 1 --no-cpu-baseline > $O/${T}_resunet_aug_bench.json 2> $O/${T}_resunet_aug_bench.err
timeout 600 python $R/tools/stream_bench.py --json $O/${T}_stream.json > $O/${T}_stream.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_graph /tmp/pf_eager
rocprofv3 --kernel-trace --stats -d /tmp/pf_graph -o p -- python $R/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_graph/p_results.db 13 > $O/${T}_resunet_graph_kernels.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pf_eager -o p -- python $R/bench.py --steps 4 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_by_grid.py /tmp/pf_eager/p_results.db k_ > $O/${T}_resunet_by_grid.txt 2>&1
for m in medformer swin_unetr; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/${T}_${m}_kernels.txt 2>&1
done
# HBM-side traffic per kernel family (separate --pmc passes; FETCH_SIZE x 2 on gfx950): "k_conv3_r" = k_conv3_r32 + k_conv3_rw
PATS=("false>(cbim::R32Params)" "k_wgrad_r32<" "k_conv_igemm<cbim::bf16_tag, 1, 2," "k_norm_bwd_apply<" "k_norm_act_fwd<" "k_up_tile<" "k_splitk_finish<")
for m in resunet medformer swin_unetr; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o f -- python $R/bench.py --model $m --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
    for pat in "${PATS[@]}" "true>(cbim::R32Params)" "k_conv_igemm<cbim::bf16_tag, 2, 2," "k_conv_wgrad<" "k_dwconv3" "k_winattn"; do echo "== $c $pat"; python $R/tools/pmc_query.py /tmp/p_$c/f_results.db "$pat"; done
  done > $O/${T}_pmc_hbm_$m.txt 2>&1
done
rm -rf /tmp/p_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/p_mfma -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
for pat in "${PATS[@]:0:3}"; do echo "== $pat"; python $R/tools/pmc_query.py /tmp/p_mfma/f_results.db "$pat" 30; done > $O/${T}_pmc_mfma.txt 2>&1
for f in resunet medformer swin resunet_aug; do head -c 330 $O/${T}_${f}_bench.json; echo; done
head -28 $O/${T}_resunet_graph_kernels.txt
cat $O/${T}_pmc_hbm_resunet.txt | head -40

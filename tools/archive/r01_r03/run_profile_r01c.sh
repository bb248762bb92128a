R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/r01_c_bench.json 2> $R/gpurun_out/r01_c_bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r -- python $R/bench.py --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/p1/r_results.db 7 > $R/gpurun_out/r01_c_resunet_kernels.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o f -- python $R/bench.py --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
  for pat in "k_conv_igemm<cbim::bf16_tag, 2, 1," "k_conv_igemm<cbim::bf16_tag, 2, 2," "k_conv_igemm<cbim::bf16_tag, 1, 2," "k_conv_wgrad<"; do
    echo "== $c $pat"; python $R/tools/pmc_query.py /tmp/p_$c/f_results.db "$pat"
  done
done > $R/gpurun_out/r01_c_pmc.txt 2>&1
cp $R/gpurun_out/prof_mf.txt /dev/null 2>&1
head -c 600 $R/gpurun_out/r01_c_bench.json; echo; cat $R/gpurun_out/r01_c_pmc.txt | head -40

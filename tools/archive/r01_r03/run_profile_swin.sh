T=r01_k; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python $R/bench.py --model swin_unetr --cpu-size 64 > $O/${T}_swin_bench.json 2> $O/${T}_swin_bench.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_s
rocprofv3 --kernel-trace --stats -d /tmp/pf_s -o p -- python $R/bench.py --model swin_unetr --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_s/p_results.db 7 > $O/${T}_swin_unetr_kernels.txt 2>&1
head -c 300 $O/${T}_swin_bench.json; echo; head -8 $O/${T}_swin_unetr_kernels.txt

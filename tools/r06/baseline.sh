#!/bin/bash
# round-6 baseline on this round's box: wave->SIMD map, Swin-shape conv layer times on the round-5 kernels, Swin / MedFormer kernel tables
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O2 tools/ubench/simd_map.hip -o /tmp/simd_map 2>/dev/null && /tmp/simd_map > $O/r06_a_simd_map.txt 2>&1
export CB_SHAPES=48x48x128,96x48x128,48x96x128,48x48x64,96x48x64,48x96x64,96x96x32,64x64x128,96x64x128
python tools/conv_bench.py bf16 10 > $O/r06_a_conv_bench_swin_shapes.txt 2>&1
python tools/conv_ab.py 10 > $O/r06_a_conv_ab_swin_shapes.txt 2>&1
unset CB_SHAPES
cd /tmp; export TMPDIR=/tmp
for m in swin_unetr medformer; do
  rm -rf /tmp/pf_$m
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$m -o p -- python $R/bench.py --model $m --steps 5 --warmup 2 --graph 0 --no-roofline --no-cpu-baseline --secondary 0 > $O/r06_a_${m}_bench.json 2>&1
  python $R/tools/rocpd_summary.py /tmp/pf_$m/p_results.db 7 > $O/r06_a_${m}_kernels.txt 2>&1
  for k in k_conv_igemm k_conv_wgrad k_wgrad_reduce k_resnorm k_conv3_rw k_wgrad_r32; do python $R/tools/rocpd_by_grid.py /tmp/pf_$m/p_results.db $k; done > $O/r06_a_${m}_by_grid.txt 2>&1
done
cd $R; python bench.py --no-cpu-baseline --secondary 1 > $O/r06_a_bench.json 2> $O/r06_a_bench.err

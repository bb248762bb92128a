#!/bin/bash
# call r05_a: the in-tree token Linears of the SwinUNETR trunk against the round-4 path (F.linear / hipBLASLt + ATen), same box
T=$1; O=gpurun_out
for tok in 0 1; do
  echo "== CBIM_SWIN_TOKEN_GEMM=$tok"
  CBIM_SWIN_TOKEN_GEMM=$tok python bench.py --model swin_unetr --no-cpu-baseline --no-roofline --secondary 0 --steps 10 --warmup 3 2>&1 | tail -1 | head -c 400; echo
done

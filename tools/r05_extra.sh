#!/bin/bash
# call r05_c: SwinUNETR / MedFormer A/Bs (same box): residual add inside k_layernorm_bwd, strip counts of the weight-gradient kernels
T=$1
run() { echo "== $1 | $2"; env $1 python bench.py --model $2 --no-cpu-baseline --no-roofline --secondary 0 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms')"; }
run "CBIM_SWIN_FUSED_RES=0" swin_unetr
run "CBIM_SWIN_FUSED_RES=1" swin_unetr
run "CBIM_PWG_WGS=256" swin_unetr
run "CBIM_PWG_WGS=384" swin_unetr
run "CBIM_PWG_WGS=1536" swin_unetr
run "CBIM_CWG_WGS=256" swin_unetr
run "CBIM_CWG_WGS=1024" swin_unetr
run "CBIM_PWG_WGS=768" medformer
run "CBIM_PWG_WGS=256" medformer
run "CBIM_PWG_WGS=384" medformer

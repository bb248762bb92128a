#!/bin/bash
# call r05_d: SwinUNETR conv1 | conv3 as one autograd node (A/B), VNet after the packed-slice cache, tests of the last edits
T=$1
run() { echo "== $1 | $2"; env $1 python bench.py --model $2 --no-cpu-baseline --no-roofline --secondary 0 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms')"; }
run "CBIM_SWIN_DUAL_CONV=0" swin_unetr
run "CBIM_SWIN_DUAL_CONV=1" swin_unetr
run "CBIM_SWIN_DUAL_CONV=0" swin_unetr
run "CBIM_SWIN_DUAL_CONV=1" swin_unetr
python tools/r04/vnet_time.py 2 2>&1 | tail -2

#!/bin/bash
# same-box A/Bs: (1) six-step half units of k_conv3_rw48 on / off, (2) the headline with round 5's conv_rw.hip against this round's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06_q}
python - <<'PY' > $O/${T}_rw48_check.txt 2>&1
import sys
sys.path.insert(0, ".")
from tests import op_checks as oc
oc.check_conv_rw48("cuda"); oc.check_conv_rw48("cuda", N=2, Cin=48, Cout=96, dhw=(9, 8, 8), seed=73)
oc.check_conv_rw48("cuda", Cin=48, Cout=48, dhw=(64, 64, 64), seed=75); print("48->48 @64 ok (strips of two tiles: full / half / full / half units)")
oc.check_conv_rw48("cuda", Cin=48, Cout=96, dhw=(32, 64, 64), seed=79); print("48->96 ok")
oc.check_norm_conv_mat48("cuda", dhw=(64, 64, 64), seed=78); print("mat48 ok")
PY
tail -4 $O/${T}_rw48_check.txt
for v in 0 1; do
  echo "CBIM_CONV_RW48_HALF=$v"; CBIM_CONV_RW48_HALF=$v CB_SHAPES=48x48x128,48x48x64 python tools/r06/conv48_ab.py 10 2>/dev/null | grep -v "^#\|wgrad"
  CBIM_CONV_RW48_HALF=$v python bench.py --model swin_unetr --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('   swin_unetr ms/step', round(d['ms_per_step'], 3))"
done 2>&1 | tee $O/${T}_half_ab.txt
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('round-6 conv_rw.hip: resunet ms/step', round(d['ms_per_step'], 3), 'k_conv3_rw avg launch us', round(d['roofline']['avg_launch_ms'] * 1e3, 2))"
done | tee $O/${T}_headline_ab.txt
cp cbim-medical-image-segmentation_amd/csrc/conv_rw.hip /tmp/conv_rw_r06.hip
python - <<'PY'
s = open("tools/r06/conv_rw_r05.hip.txt").read()
# round 5's file needs the two things other files of this tree expect from it
s = s.replace("  p.NC = d->Cin / 32;\n", "  p.NC = d->Cin / 32; p.cin_bytes = d->Cin * 2;\n")
s += '''
bool cbim_conv_rw48_eligible(const cbim_conv_desc*, const void*, int64_t, const void*, int64_t, int, const float*, const void*, const float*) { return false; }
extern "C" int cbim_conv_rw48_enable(int) { return 0; }
extern "C" int cbim_conv_rw48_takes(const cbim_conv_desc*) { return 0; }
'''
open("cbim-medical-image-segmentation_amd/csrc/conv_rw.hip", "w").write(s)
PY
make -C cbim-medical-image-segmentation_amd/csrc 2>&1 | tail -1
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('round-5 conv_rw.hip: resunet ms/step', round(d['ms_per_step'], 3), 'k_conv3_rw avg launch us', round(d['roofline']['avg_launch_ms'] * 1e3, 2))"
done | tee -a $O/${T}_headline_ab.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python - <<'PY' > $O/r06_b_rw48_check.txt 2>&1
import sys
sys.path.insert(0, ".")
from tests import op_checks as oc
oc.check_conv_rw48("cuda"); print("48->48 lrelu ok")
oc.check_conv_rw48("cuda", Cin=96, Cout=48, dhw=(8, 8, 8), act="relu", seed=72); print("96->48 ok")
oc.check_conv_rw48("cuda", N=2, Cin=48, Cout=96, dhw=(9, 8, 8), seed=73); print("48->96 ok")
oc.check_conv_rw48("cuda", Cin=8, Cout=48, dhw=(8, 8, 16), seed=74); print("8->48 ok")
oc.check_conv_rw48("cuda", Cin=48, Cout=48, dhw=(64, 64, 64), seed=75); print("48->48 @64 ok")
oc.check_conv_rw(dev="cuda"); oc.check_conv_rw("cuda", N=1, Cin=96, Cout=64, dhw=(8, 9, 8), x_split=32, wide=2); print("rw regression ok")
PY
tail -8 $O/r06_b_rw48_check.txt
python tools/r06/conv48_ab.py 10 > $O/r06_b_conv48_ab.txt 2>&1; cat $O/r06_b_conv48_ab.txt

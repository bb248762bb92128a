"""The reference's own trainer driven through the drop-in seam: /root/reference/
train.py is imported UNMODIFIED with `model.utils.get_model` resolving to `cbim_amd.model.utils.get_model`; its
`init_network` and `train_epoch` (train.py:138-233) then run on the engine (host-side kernel executor).  Skipped where
the reference tree is not mounted (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isfile("/root/reference/train.py"), reason="reference tree not mounted")
def test_reference_train_epoch_runs_on_the_engine():
    if not os.environ.get("CBIM_HIP_LIBRARY"):
        pytest.skip("needs the host-side kernel executor (CPU suite)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_trainer_seam.py")], capture_output=True, text=True,
                       timeout=1200, env=dict(os.environ))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("SEAM ")][-1]
    out = json.loads(line[5:])
    assert out["net"] == "cbim_amd.model.dim3.unet.UNet"
    assert out["iters"] == 2                                  # iter_per_epoch + 1 iterations (train.py:227-230)
    assert out["finite"] and all(l == l and 0.0 < l < 20.0 for l in out["losses"])
    assert out["moved"] > 0.0 and out["ema_moved"] > 0.0      # optimizer.step() and update_ema_variables took effect

import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import cbim_amd
from cbim_amd import ops, _lib
from cbim_amd.training import augmentation as A
from tests import aug_checks
print(_lib.backend())
mode = sys.argv[1]
if mode == "run":
    print(aug_checks.run("cuda"))
elif mode == "seedfirst":
    img = torch.randn(1,1,20,24,28).cuda(); lab = torch.zeros(1,1,20,24,28, dtype=torch.int8).cuda()
    np.random.seed(1); torch.manual_seed(1)
    print(A.random_scale_rotate_translate_3d(img, lab, [0.3]*3, [30]*3, [0]*3)[0].shape)
elif mode == "affinefirst":
    img = torch.randn(1,1,20,24,28).cuda(); lab = torch.zeros(1,1,20,24,28, dtype=torch.int8).cuda()
    print(A.random_scale_rotate_translate_3d(img, lab, [0.3]*3, [30]*3, [0]*3)[0].shape)

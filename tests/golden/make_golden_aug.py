"""Golden fixtures of the augmentation ops: the REAL reference functions
(/root/reference/training/augmentation.py) run under fixed seeds in the build container.
    python tests/golden/make_golden_aug.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 4242
SHAPE = (20, 24, 28)


def inputs():
    g = torch.Generator().manual_seed(99)
    img = torch.randn((1, 1) + SHAPE, generator=g) * 1.7 + 0.4
    coarse = torch.randint(0, 6, (1, 1, 5, 6, 7), generator=g)
    lab = torch.nn.functional.interpolate(coarse.float(), size=SHAPE, mode="nearest").to(torch.int8)
    return img, lab


def seeded(fn):
    np.random.seed(SEED)
    torch.manual_seed(SEED)
    return fn()


def main():
    sys.path.insert(0, REF)
    for missing in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    from training import augmentation as A
    img, lab = inputs()
    out = {"img": img.numpy(), "lab": lab.numpy()}
    oi, ol = seeded(lambda: A.random_scale_rotate_translate_3d(img, lab, [0.3, 0.3, 0.3], [30, 30, 30], [0, 0, 0]))
    out["affine_img"], out["affine_lab"] = oi.numpy(), ol.numpy()
    ci, cl = seeded(lambda: A.crop_3d(img, lab, [12, 16, 20], mode="random"))
    out["crop_img"], out["crop_lab"] = ci.numpy(), cl.numpy()
    out["bmul"] = seeded(lambda: A.brightness_multiply(img, multiply_range=[0.7, 1.3])).numpy()
    out["badd"] = seeded(lambda: A.brightness_additive(img, std=0.1)).numpy()
    out["gamma"] = seeded(lambda: A.gamma(img.clone(), gamma_range=[0.7, 1.5])).numpy()
    out["contrast"] = seeded(lambda: A.contrast(img.clone(), contrast_range=[0.7, 1.3])).numpy()
    out["contrast_free"] = seeded(lambda: A.contrast(img.clone(), contrast_range=[1.4, 1.8], preserve_range=False)).numpy()
    out["blur"] = seeded(lambda: A.gaussian_blur(img, sigma_range=[0.5, 1.5])).numpy()
    out["noise"] = seeded(lambda: A.gaussian_noise(img, std=0.05)).numpy()
    out["mirror1"] = A.mirror(img, axis=1).numpy()
    np.savez_compressed(os.path.join(HERE, "aug_1x1x20x24x28.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

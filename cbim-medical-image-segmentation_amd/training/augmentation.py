"""On-device 3-D augmentation with the reference's function names, signatures and RANDOM DRAW ORDER
(/root/reference/training/augmentation.py), executed by HIP kernels (csrc/augment_kernels.hip).

tensor_img: [1, C, D, H, W] float32, tensor_lab: [1, 1, D, H, W] int8 or int64, both on the GPU.
Random parameters come from the host RNGs exactly as in the reference (np.random for the affine
parameters and crop offsets, CPU torch RNG for gamma / contrast / blur / brightness / noise), so a
seeded run draws the same parameters as the reference.  Only 5-D (volume) inputs are built.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .._lib import check
from ..ops import _dev_ok, _p, _stream


def _need5(t):
    if t.dim() != 5:
        raise ValueError("Invalid input tensor dimension, should be 5d for volume image")  # 2-D images: not built
    if t.shape[0] != 1:
        raise ValueError("augmentation works on one sample: tensor_img must be [1, C, D, H, W]")


def _lab_bytes(lab):
    if lab.dtype == torch.int8:
        return 1
    if lab.dtype == torch.int64:
        return 8
    raise TypeError(f"cbim_amd: label dtype {lab.dtype} unsupported (int8 or int64)")


def _stats(x):
    """float32 [C,4] = (min, max, mean, unbiased std) of x[0, c]."""
    _, Cc, D, H, W = x.shape
    S = D * H * W
    L = _lib.lib()
    n = L.cbim_chan_stats_workspace(Cc, S)
    ws = torch.empty((n,), dtype=torch.uint8, device=x.device)
    st = torch.empty((Cc, 4), dtype=torch.float32, device=x.device)
    check(L.cbim_chan_stats(_p(x), Cc, S, _p(st), _p(ws), n, _stream(x)), "chan_stats")
    return st


def _intensity(x, mode, prm, prm_pc, st=None, st2=None, st_pc=False, noise=None):
    _dev_ok(x, prm, st, st2, noise)
    _, Cc, D, H, W = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().cbim_intensity(_p(x), _p(y), Cc, D * H * W, mode, _p(prm), int(prm_pc), _p(st), _p(st2),
                                    int(st_pc), _p(noise), _stream(x)), "intensity")
    return y


def _prm(vals_a, vals_b, device):
    """[n,2] float32 device tensor from two host sequences."""
    a = torch.as_tensor(vals_a, dtype=torch.float32).reshape(-1)
    b = torch.as_tensor(vals_b, dtype=torch.float32).reshape(-1).expand_as(a)
    return torch.stack([a, b], 1).contiguous().to(device)


# ---- intensity ops --------------------------------------------------------------------------------------

def gaussian_noise(tensor_img, std, mean=0):
    """augmentation.py:15-17 — the noise is drawn on the CPU like the reference, then added on device."""
    _need5(tensor_img)
    noise = torch.randn(tensor_img.shape).to(tensor_img.device)
    Cc = tensor_img.shape[1]
    return _intensity(tensor_img.contiguous(), 4, _prm([std] * Cc, [mean] * Cc, tensor_img.device), True, noise=noise)


def brightness_additive(tensor_img, std, mean=0, per_channel=False):
    """augmentation.py:67-82."""
    _need5(tensor_img)
    Cc = tensor_img.shape[1] if per_channel else 1
    rb = torch.normal(mean, std, size=(1, Cc, 1, 1, 1)).reshape(-1)
    return _intensity(tensor_img.contiguous(), 0, _prm([1.0] * Cc, rb, tensor_img.device), per_channel)


def brightness_multiply(tensor_img, multiply_range=[0.7, 1.3], per_channel=False):
    """augmentation.py:84-101."""
    _need5(tensor_img)
    Cc = tensor_img.shape[1] if per_channel else 1
    assert multiply_range[1] > multiply_range[0], "Invalid range"
    span = multiply_range[1] - multiply_range[0]
    rb = (torch.rand(size=(1, Cc, 1, 1, 1)) * span + multiply_range[0]).reshape(-1)
    return _intensity(tensor_img.contiguous(), 0, _prm(rb, [0.0] * Cc, tensor_img.device), per_channel)


def _flat_view(x, per_channel):
    """the reference's tensor_img.view(tmp_C, -1): per_channel=False treats all channels as one row"""
    _, Cc, D, H, W = x.shape
    return x if per_channel else x.reshape(1, 1, Cc * D, H, W)


def gamma(tensor_img, gamma_range=(0.5, 2), per_channel=False, retain_stats=True):
    """augmentation.py:104-136 (the reference draws torch.rand(C,1) even when per_channel=False and only
    works for C == 1 then; the same restriction applies here)."""
    _need5(tensor_img)
    _, Cc, D, H, W = tensor_img.shape
    g = torch.rand(Cc, 1) * (gamma_range[1] - gamma_range[0]) + gamma_range[0]
    if not per_channel and Cc != 1:
        raise RuntimeError("gamma(per_channel=False) is only defined for single-channel input (as in the reference)")
    x = _flat_view(tensor_img.contiguous(), per_channel)
    st = _stats(x)
    y = _intensity(x, 1, _prm(g.reshape(-1), [0.0] * Cc, x.device), True, st=st, st_pc=True)
    if retain_stats:
        st_y = _stats(y)
        y = _intensity(y, 2, _prm([0.0] * Cc, [0.0] * Cc, x.device), True, st=st_y, st2=st, st_pc=True)
    return y.view(1, Cc, D, H, W)


def contrast(tensor_img, contrast_range=(0.65, 1.5), per_channel=False, preserve_range=True):
    """augmentation.py:138-167."""
    _need5(tensor_img)
    _, Cc, D, H, W = tensor_img.shape
    f = torch.rand(Cc, 1) * (contrast_range[1] - contrast_range[0]) + contrast_range[0]
    if not per_channel and Cc != 1:
        raise RuntimeError("contrast(per_channel=False) is only defined for single-channel input (as in the reference)")
    x = _flat_view(tensor_img.contiguous(), per_channel)
    st = _stats(x)
    y = _intensity(x, 3, _prm(f.reshape(-1), [0.0 if preserve_range else 1.0] * Cc, x.device), True, st=st, st_pc=True)
    return y.view(1, Cc, D, H, W)


def gaussian_blur(tensor_img, sigma_range=[0.5, 1.0]):
    """augmentation.py:46-64: dense normalised k^3 Gaussian, k = 2*ceil(3*sigma)+1, zero padding; executed
    as three 1-D passes (the kernel is an outer product).  Single-channel like the reference."""
    _need5(tensor_img)
    sigma = torch.rand(1) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    kernel_size = 2 * math.ceil(3 * sigma) + 1
    _, Cc, D, H, W = tensor_img.shape
    if Cc != 1:
        raise RuntimeError("gaussian_blur is only defined for single-channel input (as in the reference)")
    xs = torch.arange(-kernel_size // 2 + 1, kernel_size // 2 + 1, dtype=torch.float32)
    g = torch.exp(-(xs ** 2) / (2 * sigma ** 2))
    g = (g / g.sum()).contiguous()
    x = tensor_img.contiguous()
    _dev_ok(x)
    y, tmp = torch.empty_like(x), torch.empty_like(x)
    garr = (C.c_float * kernel_size)(*[float(v) for v in g])
    check(_lib.lib().cbim_gaussian_blur3d(_p(x), _p(y), _p(tmp), Cc, D, H, W, C.cast(garr, C.c_void_p), kernel_size,
                                          _stream(x)), "gaussian_blur3d")
    return y


def mirror(tensor_img, axis=0):
    """augmentation.py:169-189 — a flip is a strided view + copy (torch plumbing, no arithmetic)."""
    _need5(tensor_img)
    assert axis in [0, 1, 2], "axis should be either 0, 1 or 2 for volume images"
    return torch.flip(tensor_img, dims=[2 + axis])


# ---- geometry ------------------------------------------------------------------------------------------------

def _affine_theta_3d(scale, rotate, translate, shear):
    """the reference's parameter draws and matrix product, in its order (augmentation.py:234-281)."""
    def three(v):
        return [v] * 3 if isinstance(v, (float, int)) else v
    scale, translate, rotate, shear = three(scale), three(translate), three(rotate), three(shear)
    scale_x = np.random.uniform(low=1 - scale[0], high=1 / (1 - scale[0]))
    scale_y = np.random.uniform(low=1 - scale[1], high=1 / (1 - scale[1]))
    scale_z = np.random.uniform(low=1 - scale[2], high=1 / (1 - scale[2]))
    shear_xy = np.random.uniform(-shear[0], shear[0])
    shear_xz = np.random.uniform(-shear[0], shear[0])
    shear_yx = np.random.uniform(-shear[1], shear[1])
    shear_yz = np.random.uniform(-shear[1], shear[1])
    shear_zx = np.random.uniform(-shear[2], shear[2])
    shear_zy = np.random.uniform(-shear[2], shear[2])
    translate_x = np.random.uniform(-translate[0], translate[0])
    translate_y = np.random.uniform(-translate[1], translate[1])
    translate_z = np.random.uniform(-translate[2], translate[2])
    theta_scale = torch.tensor([[scale_x, shear_xy, shear_xz, translate_x],
                                [shear_yx, scale_y, shear_yz, translate_y],
                                [shear_zx, shear_zy, scale_z, translate_z],
                                [0, 0, 0, 1]]).float()
    angle_x = (float(np.random.randint(-rotate[0], max(rotate[0], 1))) / 180.) * math.pi
    angle_y = (float(np.random.randint(-rotate[1], max(rotate[1], 1))) / 180.) * math.pi
    angle_z = (float(np.random.randint(-rotate[2], max(rotate[2], 1))) / 180.) * math.pi
    rx = torch.tensor([[1, 0, 0, 0], [0, math.cos(angle_x), -math.sin(angle_x), 0],
                       [0, math.sin(angle_x), math.cos(angle_x), 0], [0, 0, 0, 1]]).float()
    ry = torch.tensor([[math.cos(angle_y), 0, -math.sin(angle_y), 0], [0, 1, 0, 0],
                       [math.sin(angle_y), 0, math.cos(angle_y), 0], [0, 0, 0, 1]]).float()
    rz = torch.tensor([[math.cos(angle_z), -math.sin(angle_z), 0, 0], [math.sin(angle_z), math.cos(angle_z), 0, 0],
                       [0, 0, 1, 0], [0, 0, 0, 1]]).float()
    theta = torch.mm(torch.mm(torch.mm(rx, ry), rz), theta_scale)[0:3, :]
    return theta.contiguous()


def affine_sample_3d(tensor_img, tensor_lab, theta, out_size=None, offset=None):
    """grid_sample of image (trilinear) and label (nearest) under the 3x4 `theta`; out_size/offset select
    a window of the full-size result (fused crop)."""
    _need5(tensor_img)
    _dev_ok(tensor_img, tensor_lab)
    _, Cc, D, H, W = tensor_img.shape
    Do, Ho, Wo = (D, H, W) if out_size is None else [int(v) for v in out_size]
    d0, h0, w0 = (0, 0, 0) if offset is None else [int(v) for v in offset]
    img = tensor_img.contiguous()
    oimg = torch.empty((1, Cc, Do, Ho, Wo), dtype=torch.float32, device=img.device)
    olab = None
    lb = 0
    lab = None
    if tensor_lab is not None:
        lab = tensor_lab.contiguous()
        lb = _lab_bytes(lab)
        olab = torch.empty((1, 1, Do, Ho, Wo), dtype=torch.int64, device=img.device)
    th = (C.c_float * 12)(*[float(v) for v in theta.reshape(-1)])
    check(_lib.lib().cbim_affine_sample3d(_p(img), _p(lab), lb, C.cast(th, C.c_void_p), _p(oimg), _p(olab), Cc, D, H, W,
                                          Do, Ho, Wo, d0, h0, w0, _stream(img)), "affine_sample3d")
    return oimg, olab


def random_scale_rotate_translate_3d(tensor_img, tensor_lab, scale=0.3, rotate=45, translate=0.1, shear=0.05):
    """augmentation.py:226-291."""
    theta = _affine_theta_3d(scale, rotate, translate, shear)
    return affine_sample_3d(tensor_img, tensor_lab, theta)


def random_affine_center_crop_3d(tensor_img, tensor_lab, crop_size, scale=0.3, rotate=45, translate=0.1, shear=0.05):
    """random_scale_rotate_translate_3d followed by crop_3d(mode='center') (dataset_amos_ct.py:131-132) as ONE
    kernel: only the cropped window of the resampled volume is ever computed or written."""
    theta = _affine_theta_3d(scale, rotate, translate, shear)
    if isinstance(crop_size, int):
        crop_size = [crop_size] * 3
    _, _, D, H, W = tensor_img.shape
    off = [(D - crop_size[0]) // 2, (H - crop_size[1]) // 2, (W - crop_size[2]) // 2]
    return affine_sample_3d(tensor_img, tensor_lab, theta, out_size=crop_size, offset=off)


def crop_3d(tensor_img, tensor_lab, crop_size, mode):
    """augmentation.py:320-343."""
    assert mode in ["random", "center"], "Invalid Mode, should be 'random' or 'center'"
    if isinstance(crop_size, int):
        crop_size = [crop_size] * 3
    _need5(tensor_img)
    _dev_ok(tensor_img, tensor_lab)
    _, Cc, D, H, W = tensor_img.shape
    diff_D, diff_H, diff_W = D - crop_size[0], H - crop_size[1], W - crop_size[2]
    if mode == "random":
        rand_z = np.random.randint(0, max(diff_D, 1))
        rand_y = np.random.randint(0, max(diff_H, 1))
        rand_x = np.random.randint(0, max(diff_W, 1))
    else:
        rand_z, rand_y, rand_x = diff_D // 2, diff_H // 2, diff_W // 2
    return _crop_at(tensor_img, tensor_lab, crop_size, rand_z, rand_y, rand_x)


def _crop_at(tensor_img, tensor_lab, crop_size, z0, y0, x0):
    _, Cc, D, H, W = tensor_img.shape
    if min(z0, y0, x0) < 0 or z0 + crop_size[0] > D or y0 + crop_size[1] > H or x0 + crop_size[2] > W:
        raise ValueError(f"cbim_amd: crop window {(z0, y0, x0)}+{tuple(crop_size)} leaves the {(D, H, W)} volume")
    img, lab = tensor_img.contiguous(), tensor_lab.contiguous()
    oimg = torch.empty((1, Cc) + tuple(crop_size), dtype=torch.float32, device=img.device)
    olab = torch.empty((1, 1) + tuple(crop_size), dtype=lab.dtype, device=img.device)
    check(_lib.lib().cbim_crop3d(_p(img), _p(lab), _lab_bytes(lab), _p(oimg), _p(olab), Cc, D, H, W, *crop_size,
                                 z0, y0, x0, _stream(img)), "crop3d")
    return oimg, olab


def crop_around_coordinate_3d(tensor_img, tensor_lab, crop_size, coordinate, mode):
    """augmentation.py:346-382: a window of crop_size near (z, y, x) — "random": origin drawn (np.random.randint, z then y
    then x) from [max(0, c - size), min(dim - size, c + size)); "center": origin max(0, c - ceil(size/2)) clamped to the
    volume."""
    assert mode in ["random", "center"], "Invalid Mode, should be 'random' or 'center'"
    if isinstance(crop_size, int):
        crop_size = [crop_size] * 3
    _need5(tensor_img)
    _dev_ok(tensor_img, tensor_lab)
    z, y, x = (int(c) for c in coordinate)
    _, _, D, H, W = tensor_img.shape
    dims, cs = (D, H, W), tuple(int(c) for c in crop_size)
    if mode == "random":
        org = [np.random.randint(max(0, c - s), min(d - s, c + s)) for c, s, d in zip((z, y, x), cs, dims)]
    else:
        org = [min(max(0, c - math.ceil(s / 2)), d - s) for c, s, d in zip((z, y, x), cs, dims)]
    return _crop_at(tensor_img, tensor_lab, list(cs), *org)

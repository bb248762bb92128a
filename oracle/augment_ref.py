"""Oracle: 3-D augmentation ops of the reference, restated with stock torch CPU ops.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Follows
``/root/reference/training/augmentation.py`` (line numbers below); every function draws its random
parameters from the same host RNGs in the same order as the reference, so seeding np.random / torch
reproduces the reference's output (pinned by tests/golden/aug_*.npz produced by the real functions).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def gaussian_noise(x, std, mean=0):                                  # :15-17
    return x + torch.randn(x.shape) * std + mean


def brightness_additive(x, std, mean=0, per_channel=False):          # :67-82
    C = x.shape[1] if per_channel else 1
    return x + torch.normal(mean, std, size=(1, C, 1, 1, 1))


def brightness_multiply(x, multiply_range=(0.7, 1.3), per_channel=False):   # :84-101
    C = x.shape[1] if per_channel else 1
    span = multiply_range[1] - multiply_range[0]
    return x * (torch.rand(size=(1, C, 1, 1, 1)) * span + multiply_range[0])


def gamma(x, gamma_range=(0.5, 2), per_channel=False, retain_stats=True):   # :104-136
    _, C, D, H, W = x.shape
    t = x.reshape(C if per_channel else 1, -1)
    minm = t.min(dim=1)[0].unsqueeze(1)
    maxm = t.max(dim=1)[0].unsqueeze(1)
    rng = maxm - minm
    mean = t.mean(dim=1).unsqueeze(1)
    std = t.std(dim=1).unsqueeze(1)
    g = torch.rand(C, 1) * (gamma_range[1] - gamma_range[0]) + gamma_range[0]
    t = torch.pow((t - minm) / rng, g) * rng + minm
    if retain_stats:
        t = t - t.mean(dim=1).unsqueeze(1)
        t = t / t.std(dim=1).unsqueeze(1) * std + mean
    return t.view(1, C, D, H, W)


def contrast(x, contrast_range=(0.65, 1.5), per_channel=False, preserve_range=True):   # :138-167
    _, C, D, H, W = x.shape
    t = x.reshape(C if per_channel else 1, -1)
    minm = t.min(dim=1)[0].unsqueeze(1)
    maxm = t.max(dim=1)[0].unsqueeze(1)
    mean = t.mean(dim=1).unsqueeze(1)
    f = torch.rand(C, 1) * (contrast_range[1] - contrast_range[0]) + contrast_range[0]
    t = (t - mean) * f + mean
    if preserve_range:
        t = torch.clamp(t, min=minm, max=maxm)
    return t.view(1, C, D, H, W)


def gaussian_blur(x, sigma_range=(0.5, 1.0)):                        # :46-64 with :32-44
    sigma = torch.rand(1) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    k = 2 * math.ceil(3 * sigma) + 1
    r = torch.arange(-k // 2 + 1, k // 2 + 1, dtype=torch.float32)
    xx, yy, zz = torch.meshgrid(r, r, r, indexing="ij")
    ker = torch.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / (2 * sigma ** 2))
    ker = ker / (2 * math.pi * sigma ** 2) ** 1.5
    ker = (ker / ker.sum()).unsqueeze(0).unsqueeze(0)
    return F.conv3d(x, ker, padding=[k // 2] * 3)


def affine_theta_3d(scale=0.3, rotate=45, translate=0.1, shear=0.05):   # :234-281
    def three(v):
        return [v] * 3 if isinstance(v, (float, int)) else v
    scale, translate, rotate, shear = three(scale), three(translate), three(rotate), three(shear)
    sx = np.random.uniform(low=1 - scale[0], high=1 / (1 - scale[0]))
    sy = np.random.uniform(low=1 - scale[1], high=1 / (1 - scale[1]))
    sz = np.random.uniform(low=1 - scale[2], high=1 / (1 - scale[2]))
    sh = [np.random.uniform(-shear[i // 2], shear[i // 2]) for i in range(6)]   # xy, xz, yx, yz, zx, zy
    tr = [np.random.uniform(-translate[i], translate[i]) for i in range(3)]
    ts = torch.tensor([[sx, sh[0], sh[1], tr[0]], [sh[2], sy, sh[3], tr[1]], [sh[4], sh[5], sz, tr[2]],
                       [0, 0, 0, 1]]).float()
    ax = (float(np.random.randint(-rotate[0], max(rotate[0], 1))) / 180.) * math.pi
    ay = (float(np.random.randint(-rotate[1], max(rotate[1], 1))) / 180.) * math.pi
    az = (float(np.random.randint(-rotate[2], max(rotate[2], 1))) / 180.) * math.pi
    rx = torch.tensor([[1, 0, 0, 0], [0, math.cos(ax), -math.sin(ax), 0], [0, math.sin(ax), math.cos(ax), 0],
                       [0, 0, 0, 1]]).float()
    ry = torch.tensor([[math.cos(ay), 0, -math.sin(ay), 0], [0, 1, 0, 0], [math.sin(ay), 0, math.cos(ay), 0],
                       [0, 0, 0, 1]]).float()
    rz = torch.tensor([[math.cos(az), -math.sin(az), 0, 0], [math.sin(az), math.cos(az), 0, 0], [0, 0, 1, 0],
                       [0, 0, 0, 1]]).float()
    return torch.mm(torch.mm(torch.mm(rx, ry), rz), ts)[0:3, :]


def affine_sample_3d(img, lab, theta):                               # :282-289
    grid = F.affine_grid(theta.unsqueeze(0), img.size(), align_corners=True)
    oi = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    ol = F.grid_sample(lab.float(), grid, mode="nearest", padding_mode="zeros", align_corners=True).long()
    return oi, ol


def random_scale_rotate_translate_3d(img, lab, scale=0.3, rotate=45, translate=0.1, shear=0.05):   # :226-291
    return affine_sample_3d(img, lab, affine_theta_3d(scale, rotate, translate, shear))


def crop_3d(img, lab, crop_size, mode):                              # :320-343
    if isinstance(crop_size, int):
        crop_size = [crop_size] * 3
    _, _, D, H, W = img.shape
    dD, dH, dW = D - crop_size[0], H - crop_size[1], W - crop_size[2]
    if mode == "random":
        z, y, x = np.random.randint(0, max(dD, 1)), np.random.randint(0, max(dH, 1)), np.random.randint(0, max(dW, 1))
    else:
        z, y, x = dD // 2, dH // 2, dW // 2
    sl = (slice(None), slice(None), slice(z, z + crop_size[0]), slice(y, y + crop_size[1]), slice(x, x + crop_size[2]))
    return img[sl].contiguous(), lab[sl].contiguous()


def crop_around_coordinate_3d(img, lab, crop_size, coordinate, mode):   # :346-382
    if isinstance(crop_size, int):
        crop_size = [crop_size] * 3
    z, y, x = coordinate
    _, _, D, H, W = img.shape
    dD, dH, dW = D - crop_size[0], H - crop_size[1], W - crop_size[2]
    if mode == "random":
        rz = np.random.randint(max(0, z - crop_size[0]), min(dD, z + crop_size[0]))
        ry = np.random.randint(max(0, y - crop_size[1]), min(dH, y + crop_size[1]))
        rx = np.random.randint(max(0, x - crop_size[2]), min(dW, x + crop_size[2]))
    else:
        rz = min(max(0, z - math.ceil(crop_size[0] / 2)), D - crop_size[0])
        ry = min(max(0, y - math.ceil(crop_size[1] / 2)), H - crop_size[1])
        rx = min(max(0, x - math.ceil(crop_size[2] / 2)), W - crop_size[2])
    sl = (slice(None), slice(None), slice(rz, rz + crop_size[0]), slice(ry, ry + crop_size[1]), slice(rx, rx + crop_size[2]))
    return img[sl].contiguous(), lab[sl].contiguous()


def amos_train_sample(img, lab, training_size, affine_pad_size, scale, rotate, translate):
    """AMOSDataset.__getitem__, train mode (/root/reference/training/dataset/dim3/dataset_amos_ct.py:105-165) on one
    volume: img float [C,D,H,W], lab int8 [1,D,H,W]; random draws in the reference's order."""
    import numpy as np
    x, y = img.unsqueeze(0), lab.unsqueeze(0)
    _, _, d, h, w = x.shape
    if np.random.random() < 0.5:                                                                  # :124
        crop_size = [min(i + j, k) for i, j, k in zip(training_size, affine_pad_size, [d, h, w])]
        x, y = crop_3d(x, y, crop_size, mode="random")
        x, y = random_scale_rotate_translate_3d(x, y, scale, rotate, translate)
        x, y = crop_3d(x, y, training_size, mode="center")
    else:
        x, y = crop_3d(x, y, training_size, mode="random")
    x, y = x.contiguous(), y.contiguous()
    if np.random.random() < 0.2:
        x = brightness_multiply(x, multiply_range=[0.7, 1.3])
    if np.random.random() < 0.2:
        x = brightness_additive(x, std=0.1)
    if np.random.random() < 0.2:
        x = gamma(x, gamma_range=[0.7, 1.5])
    if np.random.random() < 0.2:
        x = contrast(x, contrast_range=[0.7, 1.3])
    if np.random.random() < 0.2:
        x = gaussian_blur(x, sigma_range=[0.5, 1.5])
    if np.random.random() < 0.2:
        std = np.random.random() * 0.1
        x = gaussian_noise(x, std=std)
    return x.squeeze(0), y.squeeze(0)

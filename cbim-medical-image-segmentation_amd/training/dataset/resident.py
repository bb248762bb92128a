"""HBM-resident training set + on-device sample pipeline.

The reference keeps every volume in host RAM and builds a sample per DataLoader worker on the CPU, or ships the
cropped sub-volume to the GPU for the affine step (/root/reference/training/dataset/dim3/dataset_amos_ct.py:105-165).
With 288 GB of HBM per MI355X a whole training set (AMOS-CT: 240 volumes, ~60 GB as fp32) stays resident;
``ResidentVolumeDataset.__getitem__`` reproduces the reference's train-mode sample recipe — same numpy/torch random
draws in the same order, so a seed gives the same sample — entirely with the HIP augmentation kernels
(``cbim_amd.training.augmentation``): random crop with affine padding, affine resample fused with the centre
crop, then each intensity op with probability 0.2.  ``DevicePrefetcher`` prepares the next sample on a side
stream while the current step computes.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import augmentation


class ResidentVolumeDataset:
    """images: list of float32 [D,H,W] (or [C,D,H,W]) device tensors; labels: list of int8 [D,H,W] device tensors.
    args needs training_size, affine_pad_size, scale, rotate, translate (dataset_amos_ct.py:124-153)."""

    def __init__(self, images, labels, args, mode="train"):
        if mode != "train":
            raise NotImplementedError("cbim_amd: ResidentVolumeDataset builds training samples only")
        if len(images) != len(labels) or not images:
            raise ValueError("images and labels must be non-empty lists of equal length")
        self.img_list = [i if i.dim() == 4 else i.unsqueeze(0) for i in images]
        self.lab_list = [l if l.dim() == 4 else l.unsqueeze(0) for l in labels]
        for i, l in zip(self.img_list, self.lab_list):
            if i.dtype != torch.float32 or l.dtype != torch.int8 or i.shape[1:] != l.shape[1:]:
                raise ValueError("volumes must be float32 images with int8 labels of the same spatial shape")
        self.args, self.mode = args, mode

    def __len__(self):
        return len(self.img_list)

    def __getitem__(self, idx):
        a = self.args
        idx = idx % len(self.img_list)
        tensor_img = self.img_list[idx].unsqueeze(0)      # 1, C, D, H, W
        tensor_lab = self.lab_list[idx].unsqueeze(0)
        _, _, d, h, w = tensor_img.shape
        if np.random.random() < 0.5:                       # "crop trick", dataset_amos_ct.py:124-135
            crop_size = [min(i + j, k) for i, j, k in zip(a.training_size, a.affine_pad_size, [d, h, w])]
            tensor_img, tensor_lab = augmentation.crop_3d(tensor_img, tensor_lab, crop_size, mode="random")
            # random_scale_rotate_translate_3d + crop_3d(center) as one kernel (same random draws)
            tensor_img, tensor_lab = augmentation.random_affine_center_crop_3d(
                tensor_img, tensor_lab, list(a.training_size), a.scale, a.rotate, a.translate)   # label comes back int64
        else:
            tensor_img, tensor_lab = augmentation.crop_3d(tensor_img, tensor_lab, list(a.training_size), mode="random")
        if np.random.random() < 0.2:
            tensor_img = augmentation.brightness_multiply(tensor_img, multiply_range=[0.7, 1.3])
        if np.random.random() < 0.2:
            tensor_img = augmentation.brightness_additive(tensor_img, std=0.1)
        if np.random.random() < 0.2:
            tensor_img = augmentation.gamma(tensor_img, gamma_range=[0.7, 1.5])
        if np.random.random() < 0.2:
            tensor_img = augmentation.contrast(tensor_img, contrast_range=[0.7, 1.3])
        if np.random.random() < 0.2:
            tensor_img = augmentation.gaussian_blur(tensor_img, sigma_range=[0.5, 1.5])
        if np.random.random() < 0.2:
            std = np.random.random() * 0.1
            tensor_img = augmentation.gaussian_noise(tensor_img, std=std)
        return tensor_img.squeeze(0), tensor_lab.squeeze(0)


class DevicePrefetcher:
    """Builds sample k+1 on a side stream while step k runs (one process per GPU, no DataLoader workers).
    ``next()`` returns (img [1,C,D,H,W] float32, label [1,1,D,H,W] int64) ready on the current stream."""

    def __init__(self, dataset, order=None):
        self.ds, self.k = dataset, 0
        self.order = order
        dev = dataset.img_list[0].device
        self.cuda = dev.type == "cuda"
        self.stream = torch.cuda.Stream(dev) if self.cuda else None
        self._pending = None
        self._launch()

    def _index(self):
        i = self.k if self.order is None else self.order[self.k % len(self.order)]
        self.k += 1
        return i

    def _launch(self):
        i = self._index()
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                img, lab = self.ds[i]
                out = (img.unsqueeze(0), lab.unsqueeze(0).long())
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self._pending = (out, ev)
        else:
            img, lab = self.ds[i]
            self._pending = ((img.unsqueeze(0), lab.unsqueeze(0).long()), None)

    def next(self):
        out, ev = self._pending
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            for t in out:
                t.record_stream(torch.cuda.current_stream())
        self._launch()
        return out

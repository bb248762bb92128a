#!/usr/bin/env python
"""Compulsory-traffic model of one SwinUNETR step (BASELINE.md §3 left it "to be derived"), by the rule SURVEY.md §8d uses for the
other two models: every convolution / Linear / attention core reads its input once and writes its output once, everything else
(norms, activations, residual adds, concatenation, depth-to-space) fused; the backward touches each of them twice more
(input gradient + weight gradient)  =>  bytes(step) = 3 x 2 B x sum over layers of (in + out) elements.
Layer list = /root/reference/model/dim3/swin_unetr.py:129-228 (monai blocks), :467-490, :552, :640-643, :707-731 (trunk) for
in_chan 4, feature_size 48, depths (2, 2, 2, 0), 4 classes at 128^3.      python tools/r06/swin_bytes.py"""
S, F, IN, CLS = 128, 48, 4, 4
V = [(S >> i) ** 3 for i in range(6)]            # voxels at 128^3 ... 4^3
rows = []


def add(name, v, cin, cout, n=1):
    rows.append((name, n * v * (cin + cout)))


def res_block(name, v, cin, cout):               # UnetResBlock: conv1, conv2 (+ 1x1 conv3 when the channel count changes)
    add(name + ".conv1", v, cin, cout)
    add(name + ".conv2", v, cout, cout)
    if cin != cout:
        add(name + ".conv3", v, cin, cout)


res_block("encoder1", V[0], IN, F)
res_block("encoder2", V[1], F, F)
res_block("encoder3", V[2], 2 * F, 2 * F)
res_block("encoder4", V[3], 4 * F, 4 * F)
res_block("encoder10", V[5], 16 * F, 16 * F)
for i, (lvl, cin) in enumerate(((4, 16 * F), (3, 8 * F), (2, 4 * F), (1, 2 * F), (0, F))):
    cout = cin // 2 if lvl else F
    rows.append((f"decoder{5 - i}.transp_conv", V[lvl + 1] * cin + V[lvl] * cout))
    res_block(f"decoder{5 - i}.conv_block", V[lvl], 2 * cout, cout)
add("out", V[0], F, CLS)
conv = sum(r[1] for r in rows)
trunk = []
trunk.append(("patch_embed", V[0] * IN + V[1] * F))
for st, (c, depth) in enumerate(((F, 2), (2 * F, 2), (4 * F, 2), (8 * F, 0))):
    t = V[st + 1]
    per_block = t * ((c + 3 * c) + (3 * c + c) + (c + c) + (c + 4 * c) + (4 * c + c))   # qkv, attention core, proj, fc1, fc2
    trunk.append((f"layers{st + 1} ({depth} blocks)", depth * per_block))
    trunk.append((f"layers{st + 1}.downsample", V[st + 2] * (8 * c + 2 * c)))
tr = sum(r[1] for r in trunk)
tot = conv + tr
for name, e in rows + trunk:
    if e > 0.01 * tot:
        print(f"{name:32s} {e / 1e6:9.1f} Me")
print(f"conv blocks {conv / 1e6:.1f} Me, transformer trunk {tr / 1e6:.1f} Me, sum(in + out) = {tot / 1e6:.1f} Me")
print(f"forward {2 * tot / 1e9:.2f} GB (bf16), step (x3) {6 * tot / 1e9:.2f} GB  -> HBM roofline {6 * tot / 8e12 * 1e3:.2f} ms at 8 TB/s")

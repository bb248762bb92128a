"""Per-kernel parity checks against plain torch fp32 references of the same op.

Shared by tests/test_ops_emu.py (CPU: kernels run on the host-side executor in tests/emu) and
tests/test_gpu_ops.py (-m gpu: the real gfx950 library through the C ABI).  bf16 kernels are
compared with the fp32 op evaluated on bf16-rounded inputs; tolerances are written per check.
"""
import torch
import torch.nn.functional as F

from cbim_amd import ops


def to_cl(x, dtype):  # NCDHW fp32 -> channels-last dtype
    return x.permute(0, 2, 3, 4, 1).contiguous().to(dtype)


def from_cl(x):
    return x.float().permute(0, 4, 1, 2, 3)


def tol(dtype, f32=1e-5, bf16=1e-2):
    return bf16 if dtype == torch.bfloat16 else f32


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-20))


def act_ref(name):
    """torch reference of an activation code (reference get_act, /root/reference/model/dim3/utils.py:23-30)"""
    return {"relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.01), "gelu": F.gelu, "swish": F.silu, "silu": F.silu,
            "none": lambda t: t}[name]


def check_instnorm(dev, dtype, N=2, C=24, dhw=(5, 7, 9), act="relu"):
    torch.manual_seed(1)
    x = torch.randn(N, C, *dhw) * 2 + 3.0           # |mean| >> 0 exercises the centred moments
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu())
    st = ops.instnorm_stats(xl).cpu()
    m = xr.mean((2, 3, 4))
    v = xr.var((2, 3, 4), unbiased=False)
    assert relerr(st[..., 0], m) < 1e-5
    assert relerr(st[..., 1], 1 / torch.sqrt(v + 1e-4)) < 1e-5
    fa = act_ref(act)
    y = ops.norm_act_fwd(xl, st.to(dev), ops.ACT[act])
    ref = fa(F.instance_norm(xr, eps=1e-4))
    assert relerr(from_cl(y.cpu()), ref) < tol(dtype, 1e-5, 8e-3)
    # backward of act(IN(x)) wrt x
    # (contiguous NCDHW copies for the reference: torch's CPU batch-norm backward mis-reads a permuted grad_output when
    #  no activation sits between it and the norm)
    xr2 = xr.contiguous().clone().requires_grad_(True)
    g = torch.randn_like(xr2)
    gl = to_cl(g, dtype).to(dev)
    fa(F.instance_norm(xr2, eps=1e-4)).backward(from_cl(gl.cpu()).contiguous())
    sums = ops.norm_bwd_sums(gl, xl, st.to(dev), ops.ACT[act], True)
    dx = ops.norm_bwd_apply(gl, xl, st.to(dev), sums, ops.ACT[act], True)
    assert relerr(from_cl(dx.cpu()), xr2.grad) < tol(dtype, 2e-5, 1e-2)


def check_maxpool(dev, dtype, N=1, C=16, dhw=(5, 8, 6), scale=(2, 2, 2)):
    torch.manual_seed(2)
    x = torch.randn(N, C, *dhw)
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu()).requires_grad_(True)
    y, idx = ops.maxpool_fwd(xl, scale)
    ref = F.max_pool3d(xr, scale)
    assert torch.equal(from_cl(y.cpu()), ref.detach())
    g = torch.randn_like(ref)
    gl = to_cl(g, dtype).to(dev)
    ref.backward(from_cl(gl.cpu()))
    dx = ops.maxpool_bwd(gl, idx, tuple(xl.shape), scale)
    assert torch.equal(from_cl(dx.cpu()), xr.grad)


def check_upcat(dev, dtype, N=1, Cl=16, Cs=8, low=(3, 4, 5), hi=(6, 8, 9), skip_first=True):
    torch.manual_seed(3)
    lo = torch.randn(N, Cl, *low)
    sk = torch.randn(N, Cs, *hi)
    lol, skl = to_cl(lo, dtype).to(dev), to_cl(sk, dtype).to(dev)
    lor = from_cl(lol.cpu()).requires_grad_(True)
    skr = from_cl(skl.cpu()).requires_grad_(True)
    up = F.interpolate(lor, size=hi, mode="trilinear", align_corners=True)
    ref = torch.cat([skr, up] if skip_first else [up, skr], 1)
    out = ops.upcat_fwd(lol, skl, skip_first)
    assert relerr(from_cl(out.cpu()), ref.detach()) < tol(dtype, 2e-6, 8e-3)
    # the fused variant: same tensor bit for bit + the InstanceNorm statistics of what was stored
    out2, st2 = ops.upcat_fwd_stats(lol, skl, skip_first)
    assert torch.equal(out2.cpu(), out.cpu())
    st = ops.instnorm_stats(out)
    assert relerr(st2.cpu()[..., 0], st.cpu()[..., 0]) < 1e-5 and relerr(st2.cpu()[..., 1], st.cpu()[..., 1]) < 1e-5
    g = torch.randn_like(ref)
    gl = to_cl(g, dtype).to(dev)
    ref.backward(from_cl(gl.cpu()))
    dlow, dskip = ops.upcat_bwd(gl, tuple(lol.shape), Cs, skip_first)
    assert relerr(from_cl(dlow.cpu()), lor.grad) < tol(dtype, 5e-6, 1e-2)
    assert torch.equal(from_cl(dskip.cpu()), skr.grad)


def check_up_gram_stats(dev, dtype, N=2, Cl=16, low=(3, 4, 5), hi=(6, 8, 9), offset=0.2):
    """Statistics of the virtual up-sampled tensor from the coarse grid (column sums + the tridiagonal Gram matrices of the
    three 1-D interpolations) against F.interpolate + instance statistics in float64, and against the kernel that forms the
    up-sampled values."""
    torch.manual_seed(7)
    lo = torch.randn(N, Cl, *low) * 1.5 + offset
    lol = to_cl(lo, dtype).to(dev)
    ref = F.interpolate(from_cl(lol.cpu()).double(), size=hi, mode="trilinear", align_corners=True)
    mean = ref.mean(dim=(2, 3, 4))
    rstd = 1.0 / torch.sqrt(ref.var(dim=(2, 3, 4), unbiased=False) + 1e-4)
    keep = ops.UP_GRAM
    try:
        ops.UP_GRAM = True
        st = ops.up_stats(lol, hi, 1e-4).cpu().double()
        ops.UP_GRAM = False
        st0 = ops.up_stats(lol, hi, 1e-4).cpu().double()
    finally:
        ops.UP_GRAM = keep
    assert relerr(st[..., 0], mean) < 2e-6 + 1e-6 * abs(offset) and relerr(st[..., 1], rstd) < 1e-5
    # the other kernel measures the values ROUNDED to the storage type
    assert relerr(st[..., 0], st0[..., 0]) < tol(dtype, 1e-5, 1e-3) and relerr(st[..., 1], st0[..., 1]) < tol(dtype, 1e-5, 1e-3)


def check_up_adjoint(dev, dtype, N=2, Cl=16, Cs=8, low=(3, 4, 5), hi=(6, 8, 9), skip_first=True):
    """dup -> dlow as three 1-D reductions (cbim_lin_adjoint_axis) against the one-pass 64-candidate gather, for a
    strided slice of the concatenation's gradient and for float32 NCDHW planes."""
    torch.manual_seed(6)
    g = to_cl(torch.randn(N, Cs + Cl, *hi), dtype).to(dev)
    low_shape = (N,) + tuple(low) + (Cl,)
    keep = ops.UP_SEPARABLE
    try:
        ops.UP_SEPARABLE = False
        ref, rskip = ops.upcat_bwd(g, low_shape, Cs, skip_first)
        pref = ops.trilinear_planes_bwd(from_cl(g.float()).contiguous(), (N, Cs + Cl) + tuple(low))
        ops.UP_SEPARABLE = True
        got, gskip = ops.upcat_bwd(g, low_shape, Cs, skip_first)
        pgot = ops.trilinear_planes_bwd(from_cl(g.float()).contiguous(), (N, Cs + Cl) + tuple(low))
    finally:
        ops.UP_SEPARABLE = keep
    assert got.shape == ref.shape and torch.equal(gskip.cpu(), rskip.cpu())
    assert relerr(got.cpu().float(), ref.cpu().float()) < tol(dtype, 1e-6, 8e-3)
    assert relerr(pgot.cpu(), pref.cpu()) < 1e-6


def check_upcat_fused(dev, dtype, N=2, Cl=16, Cs=8, low=(3, 4, 5), hi=(6, 8, 9), skip_first=True, tiles=True):
    """The decoder level's first block without the stored concatenation: statistics of the virtual up-sampled tensor,
    a = relu(IN([skip | up(low)])) in one pass, and the InstanceNorm backward split into (dlow, dskip) — against the
    materialising kernels (bit for bit where the arithmetic is the same) and torch autograd."""
    torch.manual_seed(4)
    lo = torch.randn(N, Cl, *low) * 1.5 + 0.2
    sk = torch.randn(N, Cs, *hi) * 0.7 - 0.1
    lol, skl = to_cl(lo, dtype).to(dev), to_cl(sk, dtype).to(dev)
    cat, st_cat = ops.upcat_fwd_stats(lol, skl, skip_first)                # round-2 path: stored concatenation
    if tiles:
        from cbim_amd import _lib
        _lib.lib().cbim_up_tile_min_tiles(0)                       # (small test volumes: take the tiled kernels anyway)
        # the LDS-tiled kernels against the gather kernels: same arithmetic in the same order -> same bits
        g0 = to_cl(torch.randn(N, Cs + Cl, *hi), dtype).to(dev)
        sums0 = ops.norm_bwd_sums(g0, cat, st_cat, 0, masked=False)
        ref, got = [], []
        for flag, dst in ((False, ref), (True, got)):
            ops.UP_TILES = flag
            dst.append(ops.up_stats(lol, hi))
            dst.append(ops.upcat_act_fwd(lol, skl, st_cat, ops.ACT["relu"], skip_first))
            dst.extend(ops.upcat_norm_bwd(g0, lol, skl, st_cat, sums0, skip_first))
        assert relerr(got[0].cpu(), ref[0].cpu()) < 1e-5          # statistics: other partition of the voxels
        for a_, b_ in zip(got[1:], ref[1:]):
            assert torch.equal(a_.cpu(), b_.cpu())
    ops.UP_TILES = tiles
    st_up = ops.up_stats(lol, hi)
    up_slice = slice(Cs, None) if skip_first else slice(0, Cl)
    # (bf16: the coarse-grid statistics are those of the fp32 interpolation, st_cat measured the values rounded to bf16)
    assert relerr(st_up.cpu()[..., 0], st_cat.cpu()[:, up_slice, 0]) < tol(dtype, 1e-5, 1e-3)
    assert relerr(st_up.cpu()[..., 1], st_cat.cpu()[:, up_slice, 1]) < tol(dtype, 1e-5, 1e-3)
    a = ops.upcat_act_fwd(lol, skl, st_cat, ops.ACT["relu"], skip_first)
    # (the stored concatenation holds the up-sampled values rounded to the storage type, the fused pass normalises the fp32
    #  interpolation: bit-identical in float32, within the bf16 rounding otherwise)
    a_cat = ops.norm_act_fwd(cat, st_cat, ops.ACT["relu"])
    assert torch.equal(a.cpu(), a_cat.cpu()) if dtype == torch.float32 else relerr(a.cpu().float(), a_cat.cpu().float()) < 2e-2
    # backward: g -> IN backward over the concatenation -> (trilinear adjoint, slice)
    g = torch.randn(N, Cs + Cl, *hi)
    gl = to_cl(g, dtype).to(dev)
    sums = ops.norm_bwd_sums(gl, cat, st_cat, 0, masked=False)
    dlow, dskip = ops.upcat_norm_bwd(gl, lol, skl, st_cat, sums, skip_first)
    dcat = ops.norm_bwd_apply(gl, cat, st_cat, sums, 0, masked=False)
    dlow_r, dskip_r = ops.upcat_bwd(dcat, tuple(lol.shape), Cs, skip_first)
    assert torch.equal(dskip.cpu(), dskip_r.cpu())
    assert relerr(from_cl(dlow.cpu()), from_cl(dlow_r.cpu())) < tol(dtype, 1e-6, 1e-2)
    # torch: d/d(low, skip) of sum(g * IN(cat([skip, up(low)])))
    lor = from_cl(lol.cpu()).requires_grad_(True)
    skr = from_cl(skl.cpu()).requires_grad_(True)
    up = F.interpolate(lor, size=hi, mode="trilinear", align_corners=True)
    ref = F.instance_norm(torch.cat([skr, up] if skip_first else [up, skr], 1), eps=1e-4)
    ref.backward(from_cl(gl.cpu()).contiguous())   # (a permuted view with N = 1 sends torch's CPU batch-norm backward down its channels-last path)
    assert relerr(from_cl(dlow.cpu()), lor.grad) < tol(dtype, 2e-4, 3e-2)
    assert relerr(from_cl(dskip.cpu()), skr.grad) < tol(dtype, 2e-4, 3e-2)


def check_conv(dev, dtype, N, Cin, Cout, dhw, k, seed=0, act="relu"):
    """conv(relu(IN(x))) forward (+epilogue statistics, +residual), dgrad (+mask, +IN-backward sums,
    +accumulate) and wgrad of one ConvNormAct (reference conv_layers.py:48-49)."""
    torch.manual_seed(seed)
    pad = [i // 2 for i in k]
    x = torch.randn(N, Cin, *dhw) + 0.5
    w = torch.randn(Cout, Cin, *k) * 0.1
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu())
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT[act])
    wdev = w.to(dev)
    wp = ops.pack_weights(wdev, geom, 0)
    st = ops.instnorm_stats(xl)
    xh = F.instance_norm(xr, eps=1e-4)
    xh_g = xh.clone().requires_grad_(True)      # for act'(xh) of the masked dgrad
    a = act_ref(act)(xh).clone()
    wr = w
    if dtype == torch.bfloat16:
        a = a.bfloat16().float()
        wr = w.bfloat16().float()
    a.requires_grad_(True)
    wr = wr.clone().requires_grad_(True)
    yr = F.conv3d(a, wr, None, 1, pad)
    res = torch.randn_like(yr)
    resl = to_cl(res, dtype).to(dev)
    y, ys = ops.conv_fwd(xl, wp, geom, in_stats=st, res=resl, want_stats=True)
    ysum = yr.detach() + from_cl(resl.cpu())
    t = tol(dtype, 2e-5, 1e-2)
    assert relerr(from_cl(y.cpu()), ysum) < t, "fwd"
    assert relerr(ys[..., 0].cpu(), ysum.mean((2, 3, 4))) < 1e-3 * (10 if dtype == torch.bfloat16 else 1) + 1e-4
    rs = 1 / torch.sqrt(ysum.var((2, 3, 4), unbiased=False) + 1e-4)
    assert relerr(ys[..., 1].cpu(), rs) < 1e-4
    # backward
    dy = torch.randn_like(yr)
    dyl = to_cl(dy, dtype).to(dev)
    yr.backward(from_cl(dyl.cpu()))
    wpd = ops.pack_weights(wdev, geom, 1)
    g, _ = ops.conv_dgrad(dyl, wpd, geom)
    assert relerr(from_cl(g.cpu()), a.grad) < t, "dgrad"
    dw = ops.conv_wgrad(xl, st, dyl, geom)
    # fp32 result of bf16 MFMA over up to ~10^5 voxels: accumulation order differs from torch's
    e_w = relerr(dw.cpu(), wr.grad)
    assert e_w < tol(dtype, 5e-5, 1e-3), f"wgrad {e_w:.3e}"
    accl = to_cl(torch.randn_like(a), dtype).to(dev)
    g2, sums = ops.conv_dgrad(dyl, wpd, geom, mask_x=xl, mask_stats=st, accumulate=accl)
    act_ref(act)(xh_g).sum().backward()
    gm = (a.grad + from_cl(accl.cpu())) * xh_g.grad                      # x act'(xh)
    assert relerr(from_cl(g2.cpu()), gm) < t, "masked dgrad"
    assert float((sums[..., 0].cpu() - gm.mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())
    assert float((sums[..., 1].cpu() - (gm * xh).mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())


def check_conv_r32(dev, N=2, Cout=32, dhw=(9, 16, 11), act="relu", seed=21, tile_depth=8, Cin=32, dy_split=0):
    """The 'weights in registers' kernel (conv_r32.hip: bf16, 3x3x3, channels in multiples of 32, Cout <= 32 or a multiple
    of 32) against torch AND against k_conv_igemm on the same inputs: transformed input + residual + statistics
    (forward), raw input (LDS-DMA path), plain dgrad, masked dgrad with the two InstanceNorm-backward sums.
    dy_split > 0: the dgrad input is the virtual concatenation [dy[..., :dy_split] | dy[..., dy_split:]] (conv1 +
    shortcut as one GEMM)."""
    from cbim_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    x = torch.randn(N, Cin, *dhw) + 0.5
    w = torch.randn(Cout, Cin, *k) * 0.1
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu())
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT[act])
    wdev = w.to(dev)
    wp, wpd = ops.pack_weights(wdev, geom, 0), ops.pack_weights(wdev, geom, 1)
    st = ops.instnorm_stats(xl)
    xh = F.instance_norm(xr, eps=1e-4)
    a = (F.relu(xh) if act == "relu" else xh.clone()).bfloat16().float()
    wr = w.bfloat16().float()
    res = torch.randn(N, Cout, *dhw)
    resl = to_cl(res, dtype).to(dev)
    dy = torch.randn(N, Cout, *dhw)
    dyl = to_cl(dy, dtype).to(dev)
    # dgrad of this layer = a convolution Cout -> Cin: the r32 shape when Cout is a multiple of 32
    dgrad_r32 = Cout % 32 == 0
    mk = torch.randn(N, Cin, *dhw) * 1.3 + 0.2
    mkl = to_cl(mk, dtype).to(dev)
    mst = ops.instnorm_stats(mkl)
    if dy_split:
        dya, dyb = dyl[..., :dy_split].contiguous(), dyl[..., dy_split:].contiguous()

    def run():
        y, ys = ops.conv_fwd(xl, wp, geom, in_stats=st, res=resl, want_stats=True)
        y0, ys0 = ops.conv_fwd(xl, wp, geom, want_stats=True)
        y1, _ = ops.conv_fwd(xl, wp, geom)
        out = [y, ys, y0, ys0, y1]
        if dgrad_r32:
            g, _ = ops.conv_dgrad(dyl, wpd, geom)
            if dy_split:
                g2, sums = ops.conv_dgrad(dya, wpd, geom, mask_x=mkl, mask_stats=mst, dy2=dyb)
            else:
                g2, sums = ops.conv_dgrad(dyl, wpd, geom, mask_x=mkl, mask_stats=mst)
            out += [g, g2, sums]
        return [o.float().cpu() for o in out]

    old, old_td = L.cbim_conv_r32_min_voxels(-1), L.cbim_conv_r32_tile_depth(tile_depth)
    try:
        L.cbim_conv_r32_min_voxels(1 << 40)
        ref = run()                       # k_conv_igemm
        L.cbim_conv_r32_min_voxels(0)
        got = run()                       # k_conv3_r32
    finally:
        L.cbim_conv_r32_min_voxels(old)
        L.cbim_conv_r32_tile_depth(old_td)
    names = ["fwd+res", "fwd stats", "raw fwd", "raw stats", "raw fwd (no stats)", "dgrad", "masked dgrad", "bwd sums"]
    for nm, r, g_ in zip(names, ref, got):
        lim = 2e-3 if nm in ("fwd stats", "raw stats", "bwd sums") else 1e-2
        assert relerr(g_, r) < lim, f"r32 vs igemm: {nm} {relerr(g_, r):.3e}"
    # and against torch
    yr = F.conv3d(a, wr, None, 1, pad) + from_cl(resl.cpu())
    assert relerr(from_cl(got[0]), yr) < 1e-2, "fwd vs torch"
    assert relerr(got[1][..., 0], yr.mean((2, 3, 4))) < 1e-2 + 1e-4
    assert relerr(from_cl(got[2]), F.conv3d(xr, wr, None, 1, pad)) < 1e-2, "raw fwd vs torch"
    if dgrad_r32:
        dyr = from_cl(dyl.cpu())
        gr = F.conv_transpose3d(dyr, wr, None, 1, pad)
        assert relerr(from_cl(got[5]), gr) < 1e-2, "dgrad vs torch"
        mh = F.instance_norm(from_cl(mkl.cpu()), eps=1e-4)
        gm = gr * ((mh > 0) if act == "relu" else 1.0)
        assert relerr(from_cl(got[6]), gm) < 1e-2, "masked dgrad vs torch"
        assert float((got[7][..., 0] - gm.mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())
        assert float((got[7][..., 1] - (gm * mh).mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())


def check_conv_rw(dev, N=2, Cin=32, Cout=32, dhw=(9, 16, 11), wide=0, x_split=0, seed=61):
    """conv_rw.hip (round 4: plane-major buffer-addressed LDS-DMA, statistics sums in LDS, optional 64-cout workgroups) against
    k_conv3_r32 on the same inputs and against torch: the raw forward (+ residual, + statistics), the plain dgrad, the dgrad
    masked by the activated tensor with the two InstanceNorm-backward sums.  wide: 0 narrow, 2 forced wide.  x_split > 0: the
    forward input is the virtual concatenation [x[..., :x_split] | x[..., x_split:]]."""
    from cbim_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    a = torch.relu(torch.randn(N, Cin, *dhw) + 0.3)
    al = to_cl(a, dtype).to(dev)
    w = torch.randn(Cout, Cin, *k) * 0.1
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT["relu"])
    wdev = w.to(dev)
    wp, wpd = ops.pack_weights(wdev, geom, 0), ops.pack_weights(wdev, geom, 1)
    resl = to_cl(torch.randn(N, Cout, *dhw) + 2.0, dtype).to(dev)
    dyl = to_cl(torch.randn(N, Cout, *dhw), dtype).to(dev)
    if x_split:
        xa, xb = al[..., :x_split].contiguous(), al[..., x_split:].contiguous()
    kern = []

    def run():
        if x_split:
            y, part = ops.conv_igemm(geom.fwd, xa, wp, tuple(resl.shape), res=resl, want_partials=True, x2=xb)
            S = dhw[0] * dhw[1] * dhw[2]
            ys = ops.stats_finalize(part, S, ops.IN_EPS, 0)
        else:
            y, ys = ops.conv_fwd(al, wp, geom, res=resl, want_stats=True)
        kern.append(L.cbim_conv3d_last_kernel())
        y0, ys0 = ops.conv_fwd(al, wp, geom, want_stats=True)
        y1, _ = ops.conv_fwd(al, wp, geom)
        out = [y, ys, y0, ys0, y1]
        if Cout % 32 == 0:
            g, _ = ops.conv_dgrad(dyl, wpd, geom)
            g2, sums = ops.conv_dgrad(dyl, wpd, geom, mask_x=al, mask_stats=None)
            kern.append(L.cbim_conv3d_last_kernel())
            out += [g, g2, sums]
        return [o.float().cpu() for o in out]

    old_mv = L.cbim_conv_r32_min_voxels(0)
    old = L.cbim_conv_rw_enable(0, wide)
    try:
        ref = run()                       # k_conv3_r32
        L.cbim_conv_rw_enable(1, wide)
        got = run()                       # k_conv3_rw
    finally:
        L.cbim_conv_r32_min_voxels(old_mv)
        L.cbim_conv_rw_enable(old & 1, old >> 1)
    n = len(kern) // 2
    assert all(kk == 1 for kk in kern[:n]) and all(kk == 2 for kk in kern[n:]), f"kernels selected: {kern}"
    names = ["fwd+res", "fwd stats", "raw fwd", "raw stats", "raw fwd (no stats)", "dgrad", "masked dgrad", "bwd sums"]
    for nm, r, g_ in zip(names, ref, got):
        if nm in ("fwd stats", "raw stats", "bwd sums"):
            assert relerr(g_, r) < 1e-4, f"rw vs r32: {nm} {relerr(g_, r):.3e}"
        else:     # the same MFMA sequence per output: the same bits
            assert torch.equal(g_, r), f"rw vs r32: {nm} differs, rel {relerr(g_, r):.3e}"
    ar = from_cl(al.cpu())
    wr = w.bfloat16().float()
    yr = F.conv3d(ar, wr, None, 1, pad)
    assert relerr(from_cl(got[2]), yr) < 1e-2, "raw fwd vs torch"
    ysum = yr + from_cl(resl.cpu())
    assert relerr(got[1][..., 0], ysum.mean((2, 3, 4))) < 1e-3
    assert relerr(got[1][..., 1], 1 / torch.sqrt(ysum.var((2, 3, 4), unbiased=False) + 1e-4)) < 2e-3
    if Cout % 32 == 0:
        gr = F.conv_transpose3d(from_cl(dyl.cpu()), wr, None, 1, pad)
        assert relerr(from_cl(got[5]), gr) < 1e-2, "dgrad vs torch"
        gm = gr * (ar > 0)
        assert relerr(from_cl(got[6]), gm) < 1e-2, "masked dgrad vs torch"
        assert float((got[7][..., 0] - gm.mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())
        assert float((got[7][..., 1] - (gm * ar).mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())


def check_conv_rw48(dev, N=1, Cin=48, Cout=48, dhw=(8, 9, 16), act="lrelu", seed=71):
    """k_conv3_rw48 (round 6: 48 output channels per workgroup, two h-pairs on waves 0..3 and one on waves 4..7, a zero-filled last
    Cin chunk) against k_conv_igemm on the same calls and against torch: raw forward (+ residual, + statistics), plain dgrad, the
    dgrad masked by the activated tensor a = act(xh) with the two InstanceNorm-backward sums — act relu or lrelu (monai's
    UnetResBlock, /root/reference/model/dim3/swin_unetr.py:129-228).  The layer (Cin -> Cout) and its input-gradient layer
    (Cout -> Cin) are both exercised when both qualify."""
    from cbim_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    slope = 0.01
    xh = torch.randn(N, Cin, *dhw) + 0.3
    a = torch.relu(xh) if act == "relu" else F.leaky_relu(xh, slope)
    al = to_cl(a, dtype).to(dev)
    w = torch.randn(Cout, Cin, *k) * 0.1
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT[act])
    wdev = w.to(dev)
    wp, wpd = ops.pack_weights(wdev, geom, 0), ops.pack_weights(wdev, geom, 1)
    resl = to_cl(torch.randn(N, Cout, *dhw) + 2.0, dtype).to(dev)
    dyl = to_cl(torch.randn(N, Cout, *dhw), dtype).to(dev)
    fwd48 = Cout % 48 == 0
    bwd48 = Cin % 48 == 0          # the dgrad launch has Cout' = Cin
    kern = []

    def run(masked):
        out = []
        y, ys = ops.conv_fwd(al, wp, geom, res=resl, want_stats=True)
        kern.append(L.cbim_conv3d_last_kernel())
        y1, _ = ops.conv_fwd(al, wp, geom)
        out += [y, ys, y1]
        g, _ = ops.conv_dgrad(dyl, wpd, geom)
        kern.append(L.cbim_conv3d_last_kernel())
        out += [g]
        if masked:
            g2, sums = ops.conv_dgrad(dyl, wpd, geom, mask_x=al, mask_stats=None)
            out += [g2, sums]
        return [o.float().cpu() for o in out]

    old = L.cbim_conv_rw48_enable(0)
    try:
        ref = run(False)                  # k_conv_igemm (it has no activated-mask epilogue for LeakyReLU: torch is the reference there)
        L.cbim_conv_rw48_enable(2)
        got = run(bwd48)
    finally:
        L.cbim_conv_rw48_enable(old)
    assert kern[0] != 5 and kern[1] != 5, f"kernels selected with the switch off: {kern}"
    assert kern[2] == (5 if fwd48 else kern[0]) and kern[3] == (5 if bwd48 else kern[1]), f"kernels selected: {kern}"
    for nm, r, g_ in zip(["fwd+res", "fwd stats", "raw fwd", "dgrad"], ref, got):
        lim = 1e-4 if nm == "fwd stats" else 1e-2       # (bf16 outputs of two different summation orders)
        assert relerr(g_, r) < lim, f"rw48 vs igemm: {nm} {relerr(g_, r):.3e}"
    ar = from_cl(al.cpu())
    wr = w.bfloat16().float()
    yr = F.conv3d(ar, wr, None, 1, pad)
    assert relerr(from_cl(got[2]), yr) < 1e-2, "raw fwd vs torch"
    ysum = yr + from_cl(resl.cpu())
    assert relerr(from_cl(got[0]), ysum) < 1e-2, "fwd + residual vs torch"
    assert relerr(got[1][..., 0], ysum.mean((2, 3, 4))) < 1e-3
    assert relerr(got[1][..., 1], 1 / torch.sqrt(ysum.var((2, 3, 4), unbiased=False) + 1e-4)) < 2e-3
    gr = F.conv_transpose3d(from_cl(dyl.cpu()), wr, None, 1, pad)
    assert relerr(from_cl(got[3]), gr) < 1e-2, "dgrad vs torch"
    if bwd48:
        pos = ar > 0
        gm = gr * pos if act == "relu" else torch.where(pos, gr, slope * gr)
        xhr = ar if act == "relu" else torch.where(pos, ar, ar / slope)
        assert relerr(from_cl(got[4]), gm) < 1e-2, "masked dgrad vs torch"
        assert float((got[5][..., 0] - gm.mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())
        assert float((got[5][..., 1] - (gm * xhr).mean((2, 3, 4))).abs().max()) < 5e-3 * float((gm * xhr).abs().max())


def check_norm_conv_mat48(dev, N=1, C=48, dhw=(8, 8, 16), seed=76):
    """functional.NormConvFn on a layer k_conv3_rw48 takes: y = conv(lrelu(IN(z))) as monai's UnetResBlock.conv2 runs it in
    SwinUNETR (/root/reference/model/dim3/swin_unetr.py:129-228) — the materialised path (a = lrelu(IN(z)) written once; raw
    forward, raw weight gradient, dgrad masked by `a`) against the normalise-on-load path of k_conv_igemm (same function, one
    bf16 rounding of `a` apart) and against torch autograd in fp32: y, its statistics, dz, dw."""
    from cbim_amd import _lib, functional as Fn
    L = _lib.lib()
    torch.manual_seed(seed)
    eps = 1e-5
    z = torch.randn(N, C, *dhw) * 1.5 + 0.2
    w = torch.randn(C, C, 3, 3, 3) * 0.08
    gy = torch.randn(N, C, *dhw)
    zl = to_cl(z, torch.bfloat16).to(dev)
    gl = to_cl(gy, torch.bfloat16).to(dev)

    def run(force):
        old = L.cbim_conv_rw48_enable(2 if force else 0)
        try:
            zz = zl.clone().requires_grad_(True)
            ww = w.detach().clone().to(dev).requires_grad_(True)
            st = ops.instnorm_stats(zz.detach(), eps)
            y, so = Fn.NormConvFn.apply(zz, st, ww, ops.ACT["lrelu"], None, True, None, eps)
            kern = L.cbim_conv3d_last_kernel()
            y.backward(gl)
            return [t.detach().float().cpu() for t in (y, so, zz.grad, ww.grad)], kern
        finally:
            L.cbim_conv_rw48_enable(old)

    ref, k0 = run(False)
    got, k1 = run(True)
    assert k0 != 5 and k1 == 5, (k0, k1)
    for nm, r, g_ in zip(["y", "stats", "dz", "dw"], ref, got):
        assert relerr(g_, r) < (2e-2 if nm in ("y", "dz") else 1e-2), f"materialised vs normalise-on-load: {nm} {relerr(g_, r):.3e}"
    # torch fp32 on the bf16-rounded operands
    zr = from_cl(zl.cpu()).requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    a = F.leaky_relu(F.instance_norm(zr, eps=eps), 0.01)
    yr = F.conv3d(a, wr, None, 1, 1)
    yr.backward(from_cl(gl.cpu()))
    assert relerr(from_cl(got[0]), yr.detach()) < 2e-2, "y vs torch"
    assert relerr(from_cl(got[2]), zr.grad) < 3e-2, f"dz vs torch {relerr(from_cl(got[2]), zr.grad):.3e}"
    assert relerr(got[3], wr.grad) < 2e-2, f"dw vs torch {relerr(got[3], wr.grad):.3e}"


def check_map_branch(dev, B=1, C=40, I=48, M=64, seed=81):
    """functional.MapQVFn / MapOutFn (k_map_gemm, round 6) against the torch ops they replace in BidirectionAttentionBlock
    (/root/reference/model/dim3/medformer_utils.py:36,40,66,93,113,127,137): InstanceNorm of the semantic map over its M positions
    -> map_qv -> [stand-in for the attention: any differentiable function of (mq, mv)] -> map_out -> + semantic_map; outputs and
    every gradient (map, both weights) in float32."""
    from cbim_amd import functional as Fn
    torch.manual_seed(seed)
    smap = torch.randn(B, C, M) * 1.3 + 0.4
    wqv = torch.randn(2 * I, C) * 0.2
    wout = torch.randn(C, I) * 0.2
    mix = torch.randn(M, M) * 0.3                  # stand-in for the attention: mo = mix @ (mq * tanh(mv))
    g = torch.randn(B, C, M)

    def tail(mq, mv, mixm):
        return torch.matmul(mixm, mq * torch.tanh(mv))

    # torch reference
    sr, wq, wo = smap.clone().requires_grad_(True), wqv.clone().requires_grad_(True), wout.clone().requires_grad_(True)
    mapp = F.layer_norm(sr, (M,), None, None, 1e-5)
    mqv = torch.matmul(wq, mapp).transpose(1, 2)                       # [B, M, 2I]
    mo = tail(mqv[..., :I], mqv[..., I:], mix)
    out = torch.matmul(wo, mo.transpose(1, 2)) + sr
    out.backward(g)
    # engine
    se, we, woe = smap.clone().to(dev).requires_grad_(True), wqv.clone().to(dev).requires_grad_(True), wout.clone().to(dev).requires_grad_(True)
    mq, mv = Fn.MapQVFn.apply(se, we, 1e-5)
    moe = tail(mq, mv, mix.to(dev))
    oute = Fn.MapOutFn.apply(moe, woe, se)
    oute.backward(g.to(dev))
    assert relerr(mq.detach().cpu(), mqv[..., :I].detach()) < 2e-5 and relerr(mv.detach().cpu(), mqv[..., I:].detach()) < 2e-5
    assert relerr(oute.detach().cpu(), out.detach()) < 2e-5, relerr(oute.detach().cpu(), out.detach())
    for nm, a, b in (("d map", se.grad, sr.grad), ("d w_qv", we.grad, wq.grad), ("d w_out", woe.grad, wo.grad)):
        assert relerr(a.cpu(), b) < 5e-5, f"{nm}: {relerr(a.cpu(), b):.3e}"


def check_upcat_skip(dev, dtype, N=2, Cu=16, Cs=24, dhw=(2, 3, 2), seed=91):
    """functional.UpCatSkipFn (round 6: the scatter half of ConvTranspose3d(k = s = 2) written straight into the concatenated tensor of
    monai's UnetrUpBlock, /root/reference/model/dim3/swin_unetr.py:176-228) against torch.cat((DepthToSpaceFn(t), skip)): bit-identical
    forward, bit-identical gradients (pure data movement)."""
    from cbim_amd import functional as Fn
    torch.manual_seed(seed)
    D, H, W = dhw
    t0 = torch.randn(N, D, H, W, 8 * Cu).to(dtype)
    sk0 = torch.randn(N, 2 * D, 2 * H, 2 * W, Cs).to(dtype)
    g = torch.randn(N, 2 * D, 2 * H, 2 * W, Cu + Cs).to(dtype).to(dev)
    outs = []
    for fused in (False, True):
        t, sk = t0.clone().to(dev).requires_grad_(True), sk0.clone().to(dev).requires_grad_(True)
        y = Fn.UpCatSkipFn.apply(t, sk, (2, 2, 2)) if fused else torch.cat((Fn.DepthToSpaceFn.apply(t, (2, 2, 2)), sk), -1)
        y.backward(g)
        outs.append((y.detach().float().cpu(), t.grad.float().cpu(), sk.grad.float().cpu()))
    for a, b, nm in zip(outs[0], outs[1], ("forward", "d t", "d skip")):
        assert torch.equal(a, b), f"UpCatSkipFn {nm} differs"


def check_se_gate(dev, N=1, C=40, H=10, seed=95):
    """functional.SEGateFn (k_se_rows / k_se_cols / k_se_outer, round 6) against the torch ops of SEBlock.excitation
    (/root/reference/model/dim3/conv_layers.py:159-175): gate, and the gradients of the means, both weights and both biases."""
    from cbim_amd import functional as Fn
    torch.manual_seed(seed)
    m = torch.randn(N, C) * 0.8 + 0.3
    w1, b1 = torch.randn(H, C) * 0.3, torch.randn(H) * 0.2
    w2, b2 = torch.randn(C, H) * 0.3, torch.randn(C) * 0.2
    gout = torch.randn(N, C)
    ref = [t.clone().requires_grad_(True) for t in (m, w1, b1, w2, b2)]
    r = torch.sigmoid(F.linear(F.relu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4]))
    r.backward(gout)
    eng = [t.clone().to(dev).requires_grad_(True) for t in (m, w1, b1, w2, b2)]
    e = Fn.SEGateFn.apply(*eng)
    e.backward(gout.to(dev))
    assert relerr(e.detach().cpu(), r.detach()) < 1e-5, "se gate"
    for nm, a, b in zip(("d mean", "d w1", "d b1", "d w2", "d b2"), eng, ref):
        assert relerr(a.grad.cpu(), b.grad) < 2e-5, f"se gate {nm}: {relerr(a.grad.cpu(), b.grad):.3e}"


def check_conv_rw_split(dev, N=1, Cin=64, Cout=64, dhw=(8, 8, 8), seed=63):
    """Low-resolution layers: k_conv3_rw over slices of the Cin chunks (blockIdx.z) + k_splitk_finish, against k_conv_igemm's
    split-K on the same call (the same finish pass: residual, activated mask, statistics) and against torch."""
    from cbim_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    a = torch.relu(torch.randn(N, Cin, *dhw) + 0.3)
    al = to_cl(a, dtype).to(dev)
    w = torch.randn(Cout, Cin, *k) * 0.1
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT["relu"])
    wdev = w.to(dev)
    wp, wpd = ops.pack_weights(wdev, geom, 0), ops.pack_weights(wdev, geom, 1)
    resl = to_cl(torch.randn(N, Cout, *dhw) + 2.0, dtype).to(dev)
    dyl = to_cl(torch.randn(N, Cout, *dhw), dtype).to(dev)
    kern = []

    def run():
        y, ys = ops.conv_fwd(al, wp, geom, res=resl, want_stats=True)
        kern.append(L.cbim_conv3d_last_kernel())
        y1, _ = ops.conv_fwd(al, wp, geom)
        out = [y, ys, y1]
        if Cout >= 64:                      # the dgrad (Cout -> Cin) has several chunks to share out
            g2, sums = ops.conv_dgrad(dyl, wpd, geom, mask_x=al, mask_stats=None)
            kern.append(L.cbim_conv3d_last_kernel())
            out += [g2, sums]
        return [o.float().cpu() for o in out]

    old = L.cbim_conv_rw_enable(0, -1)
    try:
        ref = run()                       # k_conv_igemm (split-K)
        L.cbim_conv_rw_enable(1, -1)
        got = run()                       # k_conv3_rw split-K
    finally:
        L.cbim_conv_rw_enable(old & 1, old >> 1)
    n = len(kern) // 2
    assert all(kk == 0 for kk in kern[:n]) and all(kk == 3 for kk in kern[n:]), f"kernels selected: {kern}"
    for nm, r, g_ in zip(["fwd+res", "fwd stats", "raw fwd", "masked dgrad", "bwd sums"], ref, got):
        lim = 1e-4 if nm in ("fwd stats", "bwd sums") else 1e-2        # (statistics / sums: fp32 records)
        assert relerr(g_, r) < lim, f"rw split-K vs igemm: {nm} {relerr(g_, r):.3e}"
    ar = from_cl(al.cpu())
    wr = w.bfloat16().float()
    yr = F.conv3d(ar, wr, None, 1, pad)
    assert relerr(from_cl(got[2]), yr) < 1e-2, "raw fwd vs torch"
    ysum = yr + from_cl(resl.cpu())
    assert relerr(got[1][..., 0], ysum.mean((2, 3, 4))) < 1e-3
    if Cout >= 64:
        gm = F.conv_transpose3d(from_cl(dyl.cpu()), wr, None, 1, pad) * (ar > 0)
        assert relerr(from_cl(got[3]), gm) < 1e-2, "masked dgrad vs torch"
        assert float((got[4][..., 0] - gm.mean((2, 3, 4))).abs().max()) < 5e-3 * float(gm.abs().max())


def check_wgrad_large(dev, N=2, Cin=64, Cout=32, dhw=(8, 64, 64), act="relu", raw=False, split=0, seed=5):
    """Weight gradient of a bf16 3x3x3 convolution at >= 32768 voxels (the 4x8x8-tile, fully unrolled path of
    k_conv_wgrad) against torch.  raw: no input transform (SingleConv's wgrad); split > 0: dy as two tensors (conv1 +
    shortcut as one GEMM)."""
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    x = torch.randn(N, Cin, *dhw) * 1.2 + 0.3
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu())
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT[act])
    st = None if raw else ops.instnorm_stats(xl)
    if raw:
        a = xr
    else:
        xh = F.instance_norm(xr, eps=1e-4)
        a = (F.relu(xh) if act == "relu" else xh).bfloat16().float()
    dy = torch.randn(N, Cout, *dhw)
    dyl = to_cl(dy, dtype).to(dev)
    if split:
        dw = ops.conv_wgrad(xl, st, dyl[..., :split].contiguous(), geom, dy2=dyl[..., split:].contiguous())
    else:
        dw = ops.conv_wgrad(xl, st, dyl, geom)
    w = torch.zeros(Cout, Cin, *k, requires_grad=True)
    F.conv3d(a, w, None, 1, pad).backward(from_cl(dyl.cpu()))
    e = relerr(dw.cpu(), w.grad)
    assert e < 1e-3, f"wgrad vs torch {e:.3e}"


def check_stem_head(dev, dtype, N=1, Cin=2, base=8, K=5, dhw=(6, 9, 10), k=(3, 3, 3)):
    torch.manual_seed(4)
    pad = [i // 2 for i in k]
    x = torch.randn(N, Cin, *dhw)
    w = (torch.randn(base, Cin, *k) * 0.2).requires_grad_(True)
    yr = F.conv3d(x, w, None, 1, pad)
    y = ops.stem_fwd(x.to(dev), w.detach().to(dev), pad, dtype)
    assert relerr(from_cl(y.cpu()), yr.detach()) < tol(dtype, 2e-6, 8e-3)
    dy = torch.randn_like(yr)
    dyl = to_cl(dy, dtype).to(dev)
    yr.backward(from_cl(dyl.cpu()))
    dw = ops.stem_wgrad(x.to(dev), dyl, tuple(w.shape), pad)
    assert relerr(dw.cpu(), w.grad) < 2e-5
    # head
    h = torch.randn(N, base, *dhw)
    hl = to_cl(h, dtype).to(dev)
    hr = from_cl(hl.cpu()).requires_grad_(True)
    wh = (torch.randn(K, base, 1, 1, 1) * 0.3).requires_grad_(True)
    bh = torch.randn(K).requires_grad_(True)
    zr = F.conv3d(hr, wh, bh)
    z = ops.head_fwd(hl, wh.detach().reshape(K, base).contiguous().to(dev), bh.detach().to(dev))
    assert relerr(z.cpu(), zr.detach()) < 5e-6
    dz = torch.randn(zr.shape)
    zr.backward(dz)
    dx, dwh, dbh = ops.head_bwd(hl, wh.detach().reshape(K, base).contiguous().to(dev), dz.to(dev))
    assert relerr(from_cl(dx.cpu()), hr.grad) < tol(dtype, 5e-6, 8e-3)
    assert relerr(dwh.cpu(), wh.grad.reshape(K, base)) < 2e-5
    assert relerr(dbh.cpu(), bh.grad) < 2e-5


def check_stem_mfma(dev, N=2, base=32, dhw=(6, 9, 10), k=(3, 3, 3)):
    """The matrix-core stem kernels (one input channel, bf16 rows of 32..128 channels) against torch fp32 and the VALU kernels."""
    from cbim_amd import _lib
    torch.manual_seed(45)
    BF = torch.bfloat16
    pad = [i // 2 for i in k]
    x = torch.randn(N, 1, *dhw) * 3.0 + 100.0            # CT-like offsets: the bf16 hi + lo split of the image
    w = (torch.randn(base, 1, *k) * 0.2).requires_grad_(True)
    yr = F.conv3d(x, w, None, 1, pad)
    dyl = to_cl(torch.randn_like(yr), BF).to(dev)
    yr.backward(from_cl(dyl.cpu()))
    L = _lib.lib()
    old = L.cbim_stem_mfma_enable(1)
    try:
        y = ops.stem_fwd(x.to(dev), w.detach().to(dev), pad, BF)
        dw = ops.stem_wgrad(x.to(dev), dyl, tuple(w.shape), pad)
        L.cbim_stem_mfma_enable(0)
        y0 = ops.stem_fwd(x.to(dev), w.detach().to(dev), pad, BF)
        dw0 = ops.stem_wgrad(x.to(dev), dyl, tuple(w.shape), pad)
    finally:
        L.cbim_stem_mfma_enable(old)
    assert relerr(from_cl(y.cpu()), yr.detach()) < 4e-3
    # same fp32 value up to ~2^-17 before the bf16 rounding: the two kernels agree except for a few last-bit flips
    diff = (y.cpu().float() - y0.cpu().float()).abs()
    assert float((diff > 0).float().mean()) < 0.02 and relerr(y.cpu().float(), y0.cpu().float()) < 8e-3
    assert relerr(dw.cpu(), w.grad) < 2e-5 and relerr(dw.cpu(), dw0.cpu()) < 2e-5


def check_head_mfma(dev, N=2, base=32, K=5, dhw=(4, 8, 8), need_dx=True):
    """The matrix-core head backward (k_head_bwd_mfma: bf16 rows of 32..128 channels, K <= 16, S % 32 == 0) against torch
    and against the VALU kernel it replaces."""
    from cbim_amd import _lib
    torch.manual_seed(44)
    BF = torch.bfloat16
    hl = to_cl(torch.randn(N, base, *dhw), BF).to(dev)
    hr = from_cl(hl.cpu()).requires_grad_(True)
    wh = (torch.randn(K, base, 1, 1, 1) * 0.3).requires_grad_(True)
    bh = torch.randn(K).requires_grad_(True)
    zr = F.conv3d(hr, wh, bh)
    dz = torch.randn(zr.shape) * 0.01
    dz[:, :, :1] *= 300.0                       # a wide range of magnitudes: the bf16 hi + lo split of dz
    zr.backward(dz)
    w2 = wh.detach().reshape(K, base).contiguous().to(dev)
    old = _lib.lib().cbim_head_mfma_enable(1)
    try:
        dx, dwh, dbh = ops.head_bwd(hl, w2, dz.to(dev), need_dx=need_dx)
        _lib.lib().cbim_head_mfma_enable(0)
        dx0, dwh0, dbh0 = ops.head_bwd(hl, w2, dz.to(dev))
    finally:
        _lib.lib().cbim_head_mfma_enable(old)
    if need_dx:
        assert relerr(from_cl(dx.cpu()), hr.grad) < 8e-3 and relerr(dx.cpu().float(), dx0.cpu().float()) < 8e-3
    assert relerr(dwh.cpu(), wh.grad.reshape(K, base)) < 2e-5 and relerr(dwh.cpu(), dwh0.cpu()) < 2e-5
    assert relerr(dbh.cpu(), bh.grad) < 2e-5


def check_loss(dev, N=2, C=6, dhw=(6, 7, 8), weighted=True, seed=5):
    from oracle import loss_ref
    torch.manual_seed(seed)
    z = (torch.randn(N, C, *dhw) * 2).requires_grad_(True)
    lab = torch.randint(0, C, (N, 1) + dhw)
    lab[:, :, :2] = 0          # unbalanced: some alphas clamp, some do not
    w = torch.rand(C) + 0.5 if weighted else None
    ce = loss_ref.cross_entropy(z, lab.squeeze(1), w)
    dl = loss_ref.dice_loss(z, lab)
    (0.7 * ce + 1.3 * dl).backward()
    out, coef = ops.dice_ce_fwd(z.detach().to(dev), lab.to(dev), None if w is None else w.to(dev))
    assert abs(float(out[0]) - float(ce)) < 2e-6 * max(1.0, abs(float(ce)))
    assert abs(float(out[1]) - float(dl)) < 2e-6
    g2 = torch.tensor([0.7, 1.3], device=dev)
    dz = ops.dice_ce_bwd(z.detach().to(dev), lab.to(dev), None if w is None else w.to(dev), coef, g2)
    assert relerr(dz.cpu(), z.grad) < 2e-5
    assert float(out[3]) == 0.0
    # a label outside [0, C) is counted (out[3]) and never used as an index: same result as dropping those voxels
    # from CE / the one-hot target (the reference raises instead)
    lab2 = lab.clone()
    lab2[0, 0, -1, -1, -3:] = C + 5
    lab2[-1, 0, 0, 0, 0] = -1
    out2, coef2 = ops.dice_ce_fwd(z.detach().to(dev), lab2.to(dev), None if w is None else w.to(dev))
    assert float(out2[3]) == 4.0 and all(bool(torch.isfinite(t).all()) for t in (out2, coef2))
    dz2 = ops.dice_ce_bwd(z.detach().to(dev), lab2.to(dev), None if w is None else w.to(dev), coef2, g2)
    assert bool(torch.isfinite(dz2).all())


def check_dice_reductions(dev, N=2, C=5, dhw=(6, 7, 8)):
    """DiceLoss(reduce=False) / DiceLoss(size_average=False) (losses.py:48-56) against the oracle, values and gradients (a
    per-class upstream gradient through the clamp-aware alpha coefficients)."""
    from cbim_amd.training.losses import DiceLoss
    from oracle import loss_ref
    torch.manual_seed(9)
    z = (torch.randn(N, C, *dhw) * 2)
    lab = torch.randint(0, C, (N, 1) + dhw)
    lab[:, :, :2] = 0
    gw = torch.rand(C) + 0.25
    zr = z.clone().requires_grad_(True)
    vr = loss_ref.dice_loss(zr, lab, reduce=False)
    (vr * gw).sum().backward()
    ze = z.clone().to(dev).requires_grad_(True)
    ve = DiceLoss(reduce=False)(ze, lab.to(dev))
    (ve * gw.to(dev)).sum().backward()
    assert tuple(ve.shape) == (C,) and relerr(ve.detach().cpu(), vr.detach()) < 2e-6
    assert relerr(ze.grad.cpu(), zr.grad) < 2e-5
    zs = z.clone().requires_grad_(True)
    ls = loss_ref.dice_loss(zs, lab, size_average=False)
    ls.backward()
    zt = z.clone().to(dev).requires_grad_(True)
    lt = DiceLoss(size_average=False)(zt, lab.to(dev))
    lt.backward()
    assert abs(float(lt) - float(ls)) < 2e-6 * C and relerr(zt.grad.cpu(), zs.grad) < 2e-5


def check_fused_block(dev, dtype, N=1, Cin=64, Cout=32, dhw=(4, 8, 8)):
    """BasicBlock with a conv shortcut: the fused path (conv1+shortcut as one Cout-/K-concatenated GEMM,
    strided channel-slice views) must agree with the unfused kernel sequence and with torch."""
    from cbim_amd import functional as Fn
    torch.manual_seed(11)
    k = (3, 3, 3)
    x = torch.randn(N, Cin, *dhw) * 1.5 + 0.3
    w1 = torch.randn(Cout, Cin, *k) * 0.08
    w2 = torch.randn(Cout, Cout, *k) * 0.08
    wsc = torch.randn(Cout, Cin, *k) * 0.08
    xl = to_cl(x, dtype).to(dev)
    outs = {}
    for fused in (True, False):
        orig = Fn._fusable
        Fn._fusable = (lambda x_, c_: orig(x_, c_)) if fused else (lambda x_, c_: False)
        try:
            xs = xl.clone().requires_grad_(True)
            ws = [w.to(dev).clone().requires_grad_(True) for w in (w1, w2, wsc)]
            st = ops.instnorm_stats(xs.detach())
            out, so = Fn.BasicBlockFn.apply(xs, st, ws[0], ws[1], ws[2], 1, True)
            g = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(dtype).to(dev)
            out.backward(g)
            outs[fused] = (out.detach().float().cpu(), so.cpu(), xs.grad.float().cpu(), [w.grad.cpu() for w in ws])
        finally:
            Fn._fusable = orig
    assert Fn._fusable(xl, Cout), "test shape must take the fused path"
    f, u = outs[True], outs[False]
    t = tol(dtype, 2e-5, 2e-2)
    assert relerr(f[0], u[0]) < t and relerr(f[1], u[1]) < 1e-3 and relerr(f[2], u[2]) < t
    for a, b in zip(f[3], u[3]):
        assert relerr(a, b) < tol(dtype, 5e-5, 2e-2)
    # torch reference of the block (fp32 on the rounded input)
    xr = from_cl(xl.cpu()).requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in (w1, w2, wsc)]
    a = F.relu(F.instance_norm(xr, eps=1e-4))
    y1 = F.conv3d(a, wr[0], None, 1, 1)
    ref = F.conv3d(F.relu(F.instance_norm(y1, eps=1e-4)), wr[1], None, 1, 1) + F.conv3d(a, wr[2], None, 1, 1)
    ref.backward(from_cl(g.cpu()))
    # bf16: dy1 is zero-mean per channel (InstanceNorm backward), so dw1 is a heavily cancelling sum and the
    # bf16-rounded chain differs from the fp32 chain by tens of percent of max|dw1|; the fused-vs-unfused
    # agreement above is the sharp check, this one only guards against gross errors
    t2 = tol(dtype, 5e-5, 0.5)
    assert relerr(from_cl(f[0]), ref.detach()) < t2
    assert relerr(from_cl(f[2]), xr.grad) < t2
    for a_, b_ in zip(f[3], wr):
        assert relerr(a_, b_.grad) < t2


# ---- MedFormer pieces -----------------------------------------------------------------------------

def check_dwconv(dev, dtype, N=2, C=16, dhw=(5, 6, 7), k=(3, 3, 3), act="relu", seed=11):
    """depthwise conv of act(IN(x)) (MBConv.depthwise, conv_layers.py:211) + data/weight gradients."""
    torch.manual_seed(seed)
    pad = [i // 2 for i in k]
    x = torch.randn(N, C, *dhw) + 0.3
    w = torch.randn(C, 1, *k) * 0.3
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu())
    st = ops.instnorm_stats(xl, 1e-5)
    xh = F.instance_norm(xr, eps=1e-5)
    a = (F.relu(xh) if act == "relu" else xh.clone()).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(a, wr, None, 1, pad, 1, C)
    w2d = w.reshape(C, -1).contiguous().to(dev)
    y = ops.dwconv(xl, w2d, k, in_stats=st, act=ops.ACT[act])
    t = tol(dtype, 2e-5, 1e-2)
    assert relerr(from_cl(y.cpu()), yr.detach()) < t, "dw fwd"
    yraw = ops.dwconv(xl, w2d, k)
    assert relerr(from_cl(yraw.cpu()), F.conv3d(xr, w, None, 1, pad, 1, C)) < t, "dw raw fwd"
    dy = torch.randn_like(yr)
    dyl = to_cl(dy, dtype).to(dev)
    bias = torch.randn(N, C) * 0.1
    yr.backward(from_cl(dyl.cpu()) + bias[:, :, None, None, None])
    g = ops.dwconv(dyl, w2d, k, bias=bias.to(dev), flip=True)
    assert relerr(from_cl(g.cpu()), a.grad) < t, "dw dgrad"
    dw = ops.dwconv_wgrad(xl, st, ops.ACT[act], dyl, k, dy_bias=bias.to(dev))
    assert relerr(dw.cpu(), wr.grad.reshape(C, -1)) < tol(dtype, 5e-5, 1e-2), "dw wgrad"


def check_dwconv_wgrad_mfma(dev, N=1, C=64, dhw=(8, 8, 16), act="relu", with_stats=True, with_bias=True, seed=13):
    """Depthwise 3x3x3 weight gradient on the matrix cores (k_wgrad_r32, diagonal of the 32-channel groups) against torch and
    against the vector-ALU kernels; act(IN(x)) and dy + bias are materialised first when given."""
    torch.manual_seed(seed)
    BF = torch.bfloat16
    x = torch.randn(N, C, *dhw) + 0.3
    xl = to_cl(x, BF).to(dev)
    xr = from_cl(xl.cpu())
    st = ops.instnorm_stats(xl, 1e-5) if with_stats else None
    a = xr
    if with_stats:
        xh = F.instance_norm(xr, eps=1e-5)
        a = F.relu(xh) if act == "relu" else xh
    a = a.clone().requires_grad_(True)
    wr = (torch.randn(C, 1, 3, 3, 3) * 0.3).requires_grad_(True)
    yr = F.conv3d(a, wr, None, 1, 1, 1, C)
    dyl = to_cl(torch.randn_like(yr), BF).to(dev)
    bias = torch.randn(N, C) * 0.1 if with_bias else None
    yr.backward(from_cl(dyl.cpu()) + (bias[:, :, None, None, None] if with_bias else 0.0))
    keep = ops.DW_WGRAD_MFMA
    try:
        ops.DW_WGRAD_MFMA = True
        assert ops.dwconv_wgrad_on_matrix_cores(xl, (3, 3, 3))
        dw = ops.dwconv_wgrad(xl, st, ops.ACT[act], dyl, (3, 3, 3), dy_bias=None if bias is None else bias.to(dev))
        ops.DW_WGRAD_MFMA = False
        dw0 = ops.dwconv_wgrad(xl, st, ops.ACT[act], dyl, (3, 3, 3), dy_bias=None if bias is None else bias.to(dev))
    finally:
        ops.DW_WGRAD_MFMA = keep
    # raw operands: same bf16 products, fp32 sums -> agreement to summation order; materialised operands are rounded to bf16 once
    t = 2e-5 if not (with_stats or with_bias) else 1e-2          # (the bf16 bar of check_dwconv)
    assert relerr(dw.cpu(), wr.grad.reshape(C, -1)) < t, "dw wgrad (matrix cores) vs torch"
    assert relerr(dw.cpu(), dw0.cpu()) < t, "dw wgrad (matrix cores) vs vector ALU"


def check_space_to_depth(dev, dtype, N=2, C=8, dhw=(4, 6, 8), scale=(2, 2, 2)):
    torch.manual_seed(12)
    x = torch.randn(N, C, *dhw)
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu())
    parts = [xr[:, :, i::scale[0], j::scale[1], k::scale[2]]
             for i in range(scale[0]) for j in range(scale[1]) for k in range(scale[2])]  # medformer_utils.py:163-171
    ref = torch.cat(parts, 1)
    y = ops.space_to_depth(xl, scale)
    assert torch.equal(from_cl(y.cpu()), ref)
    back = ops.depth_to_space(y, tuple(xl.shape), scale)
    assert torch.equal(back.cpu(), xl.cpu())


def _attn_ref(q, v, mq, mv, heads, scale):
    """medformer_utils.py:63-97 on [B, inner, L] / [B, inner, M] tensors, head split '(dim_head heads)'."""
    B, inner, L = q.shape
    dh = inner // heads
    r1 = lambda t: t.view(B, dh, heads, -1).permute(0, 2, 3, 1)       # b heads l dh
    fq, fv, mq_, mv_ = r1(q), r1(v), r1(mq), r1(mv)
    attn = torch.einsum('bhid,bhjd->bhij', fq, mq_) * scale
    p1 = F.softmax(attn, -1)
    p2 = F.softmax(attn, -2)
    fo = torch.einsum('bhij,bhjd->bhid', p1, mv_)
    mo = torch.einsum('bhji,bhjd->bhid', p2, fv)
    r2 = lambda t: t.permute(0, 3, 1, 2).reshape(B, inner, -1)
    return r2(fo), r2(mo)


def check_attn(dev, dtype, N=2, heads=3, dh=8, dhw=(5, 6, 7), M=8, seed=13):
    torch.manual_seed(seed)
    inner = heads * dh
    L = dhw[0] * dhw[1] * dhw[2]
    qv = torch.randn(N, 2 * inner, *dhw)
    qvl = to_cl(qv, dtype).to(dev)
    qvr = from_cl(qvl.cpu()).reshape(N, 2 * inner, L).clone().requires_grad_(True)
    mq = torch.randn(N, inner, M).requires_grad_(True)
    mv = torch.randn(N, inner, M).requires_grad_(True)
    scale = dh ** -0.5
    fo_r, mo_r = _attn_ref(qvr[:, :inner], qvr[:, inner:], mq, mv, heads, scale)
    mql = mq.detach().permute(0, 2, 1).contiguous().to(dev)
    mvl = mv.detach().permute(0, 2, 1).contiguous().to(dev)
    fo, mo, cs = ops.bidir_attn_fwd(qvl, mql, mvl, heads, scale)
    t = tol(dtype, 2e-5, 1e-2)
    assert relerr(from_cl(fo.cpu()).reshape(N, inner, L), fo_r.detach()) < t, "attn feat_out"
    # bf16 with d_head 32 / 64 codes runs on the matrix cores: map operands and probabilities are rounded to bf16
    assert relerr(mo.cpu().permute(0, 2, 1), mo_r.detach()) < tol(dtype, 2e-5, 1e-2), "attn map_out"
    dfo = torch.randn(N, inner, *dhw)
    dfol = to_cl(dfo, dtype).to(dev)
    dmo = torch.randn(N, inner, M)
    (fo_r * from_cl(dfol.cpu()).reshape(N, inner, L)).sum().backward(retain_graph=True)
    (mo_r * dmo).sum().backward()
    dqv, dmq, dmv = ops.bidir_attn_bwd(qvl, mql, mvl, cs, mo, dfol, dmo.permute(0, 2, 1).contiguous().to(dev), heads, scale)
    assert relerr(from_cl(dqv.cpu()).reshape(N, 2 * inner, L), qvr.grad) < tol(dtype, 2e-5, 3e-2), "attn dqv"
    assert relerr(dmq.cpu().permute(0, 2, 1), mq.grad) < tol(dtype, 5e-5, 3e-2), "attn dmq"
    assert relerr(dmv.cpu().permute(0, 2, 1), mv.grad) < tol(dtype, 5e-5, 3e-2), "attn dmv"


def check_attn_gemm(dev, N=2, dh=64, dhw=(8, 8, 10), M=64, seed=21, heads=1):
    """BidirectionAttention core of ONE wide head as matrix products (round 6: ops.bidir_attn_gemm_fwd / _bwd + csrc/attn_gemm_kernels.hip;
    config/lits/medformer_3d.yaml) — bf16 rows: against torch on the same rounded operands, and against the vector-ALU kernel of
    attn_wide.hip it replaces (fp32 accumulation of the same bf16 inputs): both softmaxes, all four products, every gradient."""
    torch.manual_seed(seed)
    dtype, inner = torch.bfloat16, heads * dh
    L = dhw[0] * dhw[1] * dhw[2]
    qv = torch.randn(N, 2 * inner, *dhw)
    qvl = to_cl(qv, dtype).to(dev)
    qvr = from_cl(qvl.cpu()).reshape(N, 2 * inner, L).clone().requires_grad_(True)
    mq = torch.randn(N, inner, M).requires_grad_(True)
    mv = torch.randn(N, inner, M).requires_grad_(True)
    scale = dh ** -0.5
    fo_r, mo_r = _attn_ref(qvr[:, :inner], qvr[:, inner:], mq, mv, heads, scale)
    mql = mq.detach().permute(0, 2, 1).contiguous().to(dev)
    mvl = mv.detach().permute(0, 2, 1).contiguous().to(dev)
    assert ops.awg_eligible(qvl, mql, heads) or L < 512
    fo, mo, P, Cs = ops.bidir_attn_gemm_fwd(qvl, mql, mvl, heads, scale)
    fo0, mo0, cs0 = ops.bidir_attn_fwd(qvl, mql, mvl, heads, scale)
    # P, C: rows / columns of probabilities (bf16)
    assert float((P.float().view(N, L, heads, M).sum(-1) - 1).abs().max()) < 2e-2 and float((Cs.float().sum(1) - 1).abs().max()) < 2e-2
    for name, got, ref in (("feat_out", from_cl(fo.cpu()).reshape(N, inner, L), fo_r.detach()),
                           ("map_out", mo.cpu().permute(0, 2, 1), mo_r.detach())):
        assert relerr(got, ref) < 2e-2, "attn gemm " + name + " vs torch"
    assert relerr(from_cl(fo.cpu()), from_cl(fo0.cpu())) < 2e-2 and relerr(mo.cpu(), mo0.cpu()) < 2e-2, "attn gemm vs attn_wide"
    dfo = torch.randn(N, inner, *dhw)
    dfol = to_cl(dfo, dtype).to(dev)
    dmo = torch.randn(N, inner, M)
    (fo_r * from_cl(dfol.cpu()).reshape(N, inner, L)).sum().backward(retain_graph=True)
    (mo_r * dmo).sum().backward()
    dmol = dmo.permute(0, 2, 1).contiguous().to(dev)
    dqv, dmq, dmv = ops.bidir_attn_gemm_bwd(qvl, mql, mvl, P, Cs, mo, dfol, dmol, heads, scale)
    dqv0, dmq0, dmv0 = ops.bidir_attn_bwd(qvl, mql, mvl, cs0, mo0, dfol, dmol, heads, scale)
    assert relerr(from_cl(dqv.cpu()).reshape(N, 2 * inner, L), qvr.grad) < 4e-2, "attn gemm dqv"
    assert relerr(dmq.cpu().permute(0, 2, 1), mq.grad) < 4e-2, "attn gemm dmq"
    assert relerr(dmv.cpu().permute(0, 2, 1), mv.grad) < 4e-2, "attn gemm dmv"
    assert relerr(from_cl(dqv.cpu()), from_cl(dqv0.cpu())) < 4e-2 and relerr(dmq.cpu(), dmq0.cpu()) < 4e-2 and relerr(dmv.cpu(), dmv0.cpu()) < 4e-2
    # through the autograd Function (what the model calls)
    from cbim_amd import functional as Fn
    q2 = qvl.clone().requires_grad_(True)
    a, b = mql.clone().requires_grad_(True), mvl.clone().requires_grad_(True)
    f2, m2 = Fn.BidirAttnFn.apply(q2, a, b, heads, scale)
    ((f2.float() * dfol.float()).sum() + (m2 * dmol).sum()).backward()
    assert relerr(q2.grad.cpu(), dqv.cpu()) == 0.0 and relerr(a.grad.cpu(), dmq.cpu()) == 0.0 and relerr(b.grad.cpu(), dmv.cpu()) == 0.0


def check_wgrad_133(dev, N=1, Cin=32, Cout=32, dhw=(8, 16, 16), seed=31):
    """The weight gradient of a (1, 3, 3) convolution (config/acdc: kernel_size [[1,3,3],[1,3,3],...]) taken as the centre plane of the
    3x3x3 weight gradient on k_wgrad_r32 (round 6, ops.WGRAD_EMBED_133): against torch and against k_conv_wgrad's nine-tap form."""
    torch.manual_seed(seed)
    dtype = torch.bfloat16
    x = torch.randn(N, Cin, *dhw)
    dy = torch.randn(N, Cout, *dhw) * 0.1
    xl, dyl = to_cl(x, dtype).to(dev), to_cl(dy, dtype).to(dev)
    xr, dyr = from_cl(xl.cpu()), from_cl(dyl.cpu())
    w = torch.zeros(Cout, Cin, 1, 3, 3, requires_grad=True)
    F.conv3d(xr, w, None, 1, (0, 1, 1)).backward(dyr)
    g = ops.ConvGeom(dtype, N, dhw, Cin, Cout, (1, 3, 3), (0, 1, 1), 0)
    keep = ops.WGRAD_EMBED_133
    try:
        ops.WGRAD_EMBED_133 = True
        dw1 = ops.conv_wgrad(xl, None, dyl, g)
        slot = torch.empty_like(dw1)
        assert ops.conv_wgrad(xl, None, dyl, g, out=slot) is slot and torch.equal(slot, dw1)
        ops.WGRAD_EMBED_133 = False
        dw0 = ops.conv_wgrad(xl, None, dyl, g)
    finally:
        ops.WGRAD_EMBED_133 = keep
    assert tuple(dw1.shape) == (Cout, Cin, 1, 3, 3) and dw1.is_contiguous()
    assert relerr(dw1.cpu(), w.grad) < 2e-5 and relerr(dw0.cpu(), w.grad) < 2e-5, (relerr(dw1.cpu(), w.grad), relerr(dw0.cpu(), w.grad))


def check_mappool(dev, dtype, N=2, C=24, M=8, dhw=(5, 6, 7), seed=14):
    """SemanticMapGeneration tail, medformer_utils.py:218-228."""
    torch.manual_seed(seed)
    L = dhw[0] * dhw[1] * dhw[2]
    fw = torch.randn(N, C + M, *dhw)
    fwl = to_cl(fw, dtype).to(dev)
    fwr = from_cl(fwl.cpu()).reshape(N, C + M, L).clone().requires_grad_(True)
    wm = F.softmax(fwr[:, C:], dim=2)
    ref = torch.einsum('bij,bkj->bik', fwr[:, :C], wm)
    mp, cs = ops.colsoftmax_pool_fwd(fwl, C)
    assert relerr(mp.cpu(), ref.detach()) < 2e-5, "mappool fwd"
    dmap = torch.randn(N, C, M)
    ref.backward(dmap)
    keep = ops.MAPPOOL_BWD_GEMM
    try:
        for flag in (True, False):          # two library GEMMs (small volumes) | k_mappool_bwd / k_mappool_bwd4
            ops.MAPPOOL_BWD_GEMM = flag
            dfw = ops.colsoftmax_pool_bwd(fwl, C, mp, cs, dmap.to(dev))
            assert dfw.shape == fwl.shape and dfw.dtype == fwl.dtype
            assert relerr(from_cl(dfw.cpu()).reshape(N, C + M, L), fwr.grad) < tol(dtype, 2e-5, 1e-2), "mappool bwd"
    finally:
        ops.MAPPOOL_BWD_GEMM = keep


def check_trilinear_planes(dev, N=1, C=3, lo=(3, 4, 5), hi=(7, 8, 9)):
    torch.manual_seed(15)
    x = torch.randn(N, C, *lo, requires_grad=True)
    ref = F.interpolate(x, size=hi, mode="trilinear", align_corners=True)
    y = ops.trilinear_planes_fwd(x.detach().to(dev), hi)
    assert relerr(y.cpu(), ref.detach()) < 2e-6
    g = torch.randn_like(ref)
    ref.backward(g)
    dx = ops.trilinear_planes_bwd(g.to(dev), tuple(x.shape))
    assert relerr(dx.cpu(), x.grad) < 2e-6


# ---- SwinUNETR pieces -----------------------------------------------------------------------------

def check_window_attn(dev, dtype, B=1, dhw=(9, 8, 7), C=24, heads=3, window=(7, 7, 7), shift=(3, 3, 3), seed=21):
    """pad + roll + partition + WindowAttention core + reverse + roll back + crop (swin_unetr.py:467-490,554-606)
    against the oracle's stock-torch restatement of the same lines."""
    from oracle import swin_unetr_ref as R
    torch.manual_seed(seed)
    D, H, W = dhw
    ws, ss = R.effective_window(dhw, window, shift)
    x = torch.randn(B, D, H, W, C)                                   # plays norm1's output
    wq = torch.randn(3 * C, C) * 0.3
    bq = torch.randn(3 * C) * 0.3
    table = torch.randn((2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1), heads) * 0.5
    rel = R.relative_position_index(window)
    # reference: qkv of padded tokens is the bias (linear of zeros)
    xr = x.clone().requires_grad_(True)
    bqr = bq.clone().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    pd, ph, pw = (ws[0] - D % ws[0]) % ws[0], (ws[1] - H % ws[1]) % ws[1], (ws[2] - W % ws[2]) % ws[2]
    xp = F.pad(xr, (0, 0, 0, pw, 0, ph, 0, pd))
    dims = xp.shape[1:4]
    shifted = any(s > 0 for s in ss)
    mask = None
    if shifted:
        xp = torch.roll(xp, shifts=tuple(-s for s in ss), dims=(1, 2, 3))
        mask = R.region_mask(dims, ws, ss, torch.float32)
    qkv_w = F.linear(R.partition(xp, ws), wq, bqr)
    o = R.window_attention_core(qkv_w, tr, rel, heads, mask)
    o = R.unpartition(o, ws, (B,) + tuple(dims))
    if shifted:
        o = torch.roll(o, shifts=ss, dims=(1, 2, 3))
    ref = o[:, :D, :H, :W, :]
    # kernel: qkv computed on the real tokens only
    qkv = F.linear(x, wq, bq).to(dtype).to(dev)
    out, lse = ops.window_attn_fwd(qkv, bq.to(dev), table.to(dev), heads, ws, ss, window)
    # bf16 I/O variant: q/k/v rounded to bf16 feed softmax logits of magnitude ~10 -> percent-level output changes
    # (the product runs the transformer trunk in fp32; bf16 here only exercises the 2-byte load/store path)
    t = tol(dtype, 2e-5, 1e-1)
    e = relerr(out.float().cpu(), ref.detach())
    assert e < t, f"window attention fwd {e:.3e}"
    if dtype != torch.float32:
        return
    g = torch.randn_like(ref)
    qkv_leaf = F.linear(xr, wq, bqr)      # for d(qkv) of real tokens: differentiate through a fresh graph
    ref.backward(g)
    dqkv, dtable, dbias = ops.window_attn_bwd(qkv, bq.to(dev), table.to(dev), out, g.to(dev), lse, heads, ws, ss, window)
    # d x = dqkv @ Wq;  d bias = sum over real tokens of dqkv + gradient through padded keys
    dx = dqkv.float().cpu() @ wq
    assert relerr(dx, xr.grad) < 5e-5, "window attention dqkv"
    assert relerr(dtable.cpu(), tr.grad) < 5e-5, "window attention dtable"
    db = dqkv.float().cpu().reshape(-1, 3 * C).sum(0) + dbias.cpu()
    assert relerr(db, bqr.grad) < 5e-5, "window attention dbias"


def check_resnorm(dev, dtype, N=2, C=16, dhw=(4, 5, 6), with_b_stats=True, seed=22):
    """monai UnetResBlock tail: lrelu(IN(a) + (IN(b) | b)), eps 1e-5."""
    torch.manual_seed(seed)
    a = torch.randn(N, C, *dhw) * 1.5 + 0.3
    b = torch.randn(N, C, *dhw)
    al, bl = to_cl(a, dtype).to(dev), to_cl(b, dtype).to(dev)
    ar = from_cl(al.cpu()).requires_grad_(True)
    br = from_cl(bl.cpu()).requires_grad_(True)
    sa = ops.instnorm_stats(al, 1e-5)
    sb = ops.instnorm_stats(bl, 1e-5) if with_b_stats else None
    ref = F.leaky_relu(F.instance_norm(ar, eps=1e-5) + (F.instance_norm(br, eps=1e-5) if with_b_stats else br), 0.01)
    y = ops.resnorm_fwd(al, sa, bl, sb, ops.ACT["lrelu"])
    t = tol(dtype, 2e-5, 1e-2)
    assert relerr(from_cl(y.cpu()), ref.detach()) < t, "resnorm fwd"
    g = torch.randn_like(ref)
    gl = to_cl(g, dtype).to(dev)
    ref.backward(from_cl(gl.cpu()))
    da, db = ops.resnorm_bwd(gl, al, sa, bl, sb, ops.ACT["lrelu"])
    assert relerr(from_cl(da.cpu()), ar.grad) < tol(dtype, 5e-5, 2e-2), "resnorm da"
    assert relerr(from_cl(db.cpu()), br.grad) < tol(dtype, 5e-5, 2e-2), "resnorm db"


def check_window_attn_mfma(dev, B=1, dhw=(9, 8, 7), C=48, heads=3, window=(7, 7, 7), shift=(3, 3, 3), seed=23):
    """The matrix-core window attention (swin_mfma.hip: bf16, d_head 16) against the vector-ALU fp32 kernels of
    swin_kernels.hip fed with the SAME bf16-rounded q/k/v (and, for the backward, the same bf16-rounded upstream gradient):
    the only differences left are the bf16 rounding of the probabilities / score gradients that feed the second GEMMs."""
    from oracle import swin_unetr_ref as R
    torch.manual_seed(seed)
    D, H, W = dhw
    ws, ss = R.effective_window(dhw, window, shift)
    bq = (torch.randn(3 * C) * 0.3).to(dev)
    table = (torch.randn((2 * window[0] - 1) * (2 * window[1] - 1) * (2 * window[2] - 1), heads) * 0.5).to(dev)
    qkv16 = (torch.randn(B, D, H, W, 3 * C) * 0.7).bfloat16().to(dev)
    qkv32 = qkv16.float()
    o32, lse32 = ops.window_attn_fwd(qkv32, bq, table, heads, ws, ss, window)
    o16, lse16 = ops.window_attn_fwd(qkv16, bq, table, heads, ws, ss, window)
    e = relerr(o16.float().cpu(), o32.cpu())
    assert e < 1.5e-2, f"window attention (mfma) fwd {e:.3e}"
    n = ws[0] * ws[1] * ws[2]
    assert relerr(lse16.cpu()[..., :n], lse32.cpu()[..., :n]) < 1e-3, "window attention (mfma) lse"
    g16 = torch.randn(B, D, H, W, C).bfloat16().to(dev)
    dq32, dt32, db32 = ops.window_attn_bwd(qkv32, bq, table, o32, g16.float(), lse32, heads, ws, ss, window)
    dq16, dt16, db16 = ops.window_attn_bwd(qkv16, bq, table, o16, g16, lse16, heads, ws, ss, window)
    assert relerr(dq16.float().cpu(), dq32.cpu()) < 3e-2, f"window attention (mfma) dqkv {relerr(dq16.float().cpu(), dq32.cpu()):.3e}"
    assert relerr(dt16.cpu(), dt32.cpu()) < 3e-2, f"window attention (mfma) dtable {relerr(dt16.cpu(), dt32.cpu()):.3e}"
    assert relerr(db16.cpu(), db32.cpu()) < 3e-2 + 1e-6, f"window attention (mfma) dbias {relerr(db16.cpu(), db32.cpu()):.3e}"


def check_wgrad_r32(dev, N=1, Cin=32, Cout=32, dhw=(8, 16, 8), split=0, xsplit=0, seed=7):
    """Weight gradient of a bf16 3x3x3 convolution on a RAW input (the caller materialised act(IN(x))) through
    k_wgrad_r32 (8x8x8 tiles, LDS-DMA double buffering, 27 tap accumulators in registers) against torch.
    split > 0: dy as two tensors (conv1 + shortcut as one GEMM); xsplit > 0: the input as two tensors (virtual concat)."""
    from cbim_amd import _lib
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    x = torch.relu(torch.randn(N, Cin, *dhw) * 1.2 + 0.3)
    xl = to_cl(x, dtype).to(dev)
    a = from_cl(xl.cpu())
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT["relu"])
    dy = torch.randn(N, Cout, *dhw)
    dyl = to_cl(dy, dtype).to(dev)
    kw = {}
    if split:
        kw["dy2"] = dyl[..., split:].contiguous()
    if xsplit:
        kw["x2"] = xl[..., xsplit:]                      # a channel-range view of the wider tensor (row stride Cin)
    dw = ops.conv_wgrad(xl[..., :xsplit].contiguous() if xsplit else xl, None, dyl[..., :split].contiguous() if split else dyl,
                        geom, **kw)
    assert _lib.lib().cbim_conv3d_wgrad_last_kernel() == 1, "k_wgrad_r32 was not selected"
    w = torch.zeros(Cout, Cin, *k, requires_grad=True)
    F.conv3d(a, w, None, 1, pad).backward(from_cl(dyl.cpu()))
    e = relerr(dw.cpu(), w.grad)
    assert e < 1e-3, f"wgrad r32 vs torch {e:.3e}"


def check_layernorm(dev, rows=(3, 5, 7), C=48, affine=True, out_bf16=False, seed=31):
    """cbim_layernorm_fwd / _bwd against torch's layer_norm on the same fp32 rows (the SwinUNETR trunk's LayerNorms)."""
    from cbim_amd import functional as Fn
    torch.manual_seed(seed)
    x = (torch.randn(*rows, C) * 1.7 + 0.4)
    w = (torch.randn(C) * 0.3 + 1.0) if affine else None
    b = (torch.randn(C) * 0.2) if affine else None
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True) if affine else None
    br = b.clone().requires_grad_(True) if affine else None
    ref = F.layer_norm(xr, (C,), wr, br, 1e-5)
    g = torch.randn_like(ref)
    xe = x.clone().to(dev).requires_grad_(True)
    we = w.clone().to(dev).requires_grad_(True) if affine else None
    be = b.clone().to(dev).requires_grad_(True) if affine else None
    od = torch.bfloat16 if out_bf16 else torch.float32
    y = Fn.layer_norm(xe, we, be, 1e-5, od)
    assert y.dtype == od
    assert relerr(y.float().cpu(), ref.detach()) < (8e-3 if out_bf16 else 2e-6)
    gd = g.to(dev).to(od)
    ref.backward(gd.float().cpu())
    y.backward(gd)
    assert relerr(xe.grad.cpu(), xr.grad) < 2e-5
    if affine:
        assert relerr(we.grad.cpu(), wr.grad) < 2e-5 and relerr(be.grad.cpu(), br.grad) < 2e-5
    # the two-output form (LN(x), x) of a pre-norm residual branch: out = x + 0.7 * LN(x) * v, both gradients in one kernel
    v = torch.randn(C)
    xr2 = x.clone().requires_grad_(True)
    (xr2 + 0.7 * F.layer_norm(xr2, (C,), w, b, 1e-5) * v).backward(g)
    xe2 = x.clone().to(dev).requires_grad_(True)
    y2, xp = Fn.LayerNormResFn.apply(xe2, we.detach() if affine else None, be.detach() if affine else None, 1e-5, torch.float32)
    (xp + 0.7 * y2 * v.to(dev)).backward(g.to(dev))
    assert relerr(xe2.grad.cpu(), xr2.grad) < 2e-5
    xe3 = x.clone().to(dev).requires_grad_(True)                       # only the pass-through used: the gradient is handed on as it is
    _, xp3 = Fn.LayerNormResFn.apply(xe3, None, None, 1e-5, torch.float32)
    (xp3 * 2.0).sum().backward()
    assert torch.equal(xe3.grad.cpu(), torch.full_like(x, 2.0))


def check_dgrad_mask_by_activated(dev, dtype, N=1, Cin=32, Cout=32, dhw=(8, 16, 8), seed=41):
    """Masked dgrad whose mask tensor is the activated a = relu(IN(x)) itself (mask_stats None) against the same launch
    with identity statistics (mean 0, rstd 1) — the same arithmetic, so the same bits — and against torch."""
    torch.manual_seed(seed)
    k, pad = (3, 3, 3), (1, 1, 1)
    a = torch.relu(torch.randn(N, Cin, *dhw))
    al = to_cl(a, dtype).to(dev)
    w = torch.randn(Cout, Cin, *k) * 0.1
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT["relu"])
    wd = ops.pack_weights(w.to(dev), geom, 1)
    dy = torch.randn(N, Cout, *dhw)
    dyl = to_cl(dy, dtype).to(dev)
    ident = torch.zeros(N, Cin, 2)
    ident[..., 1] = 1.0
    from cbim_amd import _lib
    L = _lib.lib()
    old = L.cbim_conv_r32_min_voxels(0)                   # bf16 multiples of 32: force k_conv3_r32 on this small volume
    try:
        g0, s0 = ops.conv_dgrad(dyl, wd, geom, mask_x=al, mask_stats=ident.to(dev))
        k0 = L.cbim_conv3d_last_kernel()
        g1, s1 = ops.conv_dgrad(dyl, wd, geom, mask_x=al, mask_stats=None)
        k1 = L.cbim_conv3d_last_kernel()
        if dtype == torch.bfloat16 and Cin % 32 == 0 and Cout % 32 == 0:
            assert k0 == 1 and k1 in (1, 2)       # k_conv3_r32; the activated-mask call may run on its round-4 form k_conv3_rw
    finally:
        L.cbim_conv_r32_min_voxels(old)
    assert torch.equal(g0.cpu(), g1.cpu())
    if k0 == k1:
        assert torch.equal(s0.cpu(), s1.cpu())
    else:                                         # the two kernels sum the lanes in different orders
        assert relerr(s1.cpu(), s0.cpu()) < 1e-4
    ar = from_cl(al.cpu())
    wr = w.to(dtype).float() if dtype == torch.bfloat16 else w
    xin = torch.zeros(N, Cin, *dhw, requires_grad=True)
    F.conv3d(xin, wr, None, 1, pad).backward(from_cl(dyl.cpu()).contiguous())
    ref = xin.grad * (ar > 0)
    assert relerr(from_cl(g1.cpu()), ref) < tol(dtype, 1e-5, 1e-2)
    assert relerr(s1.cpu()[..., 0], ref.mean((2, 3, 4))) < tol(dtype, 1e-4, 2e-2)
    assert relerr(s1.cpu()[..., 1], (ref * ar).mean((2, 3, 4))) < tol(dtype, 1e-4, 2e-2)


def check_colsum(dev, dtype, rows=1000, C=48, seed=51):
    torch.manual_seed(seed)
    x = torch.randn(rows, C).to(dtype)
    got = ops.colsum(x.to(dev))
    ref = x.float().sum(0)
    assert relerr(got.cpu(), ref) < 1e-5


def check_conv_pw(dev, N=2, Cin=64, Cout=96, dhw=(5, 6, 7), act="relu", seed=71):
    """conv_pw.hip (bf16 1x1x1 convolutions as a row GEMM without operand staging) against k_conv_igemm on the same calls and
    against torch: forward with InstanceNorm + activation on load, residual and output statistics; raw forward; plain dgrad;
    dgrad masked through the pre-activation (statistics given) with the two InstanceNorm-backward sums; dgrad masked by an
    activated tensor."""
    from cbim_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    torch.manual_seed(seed)
    k, pad = (1, 1, 1), (0, 0, 0)
    S = dhw[0] * dhw[1] * dhw[2]
    x = torch.randn(N, Cin, *dhw) * 1.5 + 0.4
    xl = to_cl(x, dtype).to(dev)
    xs = ops.instnorm_stats(xl, ops.IN_EPS)
    w = torch.randn(Cout, Cin, *k) * 0.1
    geom = ops.ConvGeom(dtype, N, dhw, Cin, Cout, k, pad, ops.ACT[act])
    wdev = w.to(dev)
    wp, wpd = ops.pack_weights(wdev, geom, 0), ops.pack_weights(wdev, geom, 1)
    resl = to_cl(torch.randn(N, Cout, *dhw) + 2.0, dtype).to(dev)
    dyl = to_cl(torch.randn(N, Cout, *dhw), dtype).to(dev)
    accl = to_cl(torch.randn(N, Cin, *dhw), dtype).to(dev)
    al = ops.norm_act_fwd(xl, xs, ops.ACT[act])
    kern = []

    def run():
        y, ys = ops.conv_fwd(xl, wp, geom, in_stats=xs, res=resl, want_stats=True)
        kern.append(L.cbim_conv3d_last_kernel())
        y0, ys0 = ops.conv_fwd(xl, wp, geom, want_stats=True)
        y1, _ = ops.conv_fwd(xl, wp, geom, in_stats=xs)
        g, _ = ops.conv_dgrad(dyl, wpd, geom)
        g1, sums1 = ops.conv_dgrad(dyl, wpd, geom, mask_x=xl, mask_stats=xs, accumulate=accl)
        kern.append(L.cbim_conv3d_last_kernel())
        out = [y, ys, y0, ys0, y1, g, g1, sums1]
        if act == "relu":
            g2, sums2 = ops.conv_dgrad(dyl, wpd, geom, mask_x=al, mask_stats=None)
            out += [g2, sums2]
        else:
            out += [g, sums1]
        dw = ops.conv_wgrad(xl, xs, dyl, geom)                   # k_pw_wgrad + reduce | k_conv_wgrad + reduce
        kern.append(10 + L.cbim_conv3d_wgrad_last_kernel())
        dw0 = ops.conv_wgrad(xl, None, dyl, geom)
        out += [dw, dw0]
        return [o.float().cpu() for o in out]

    old = L.cbim_conv_pw_enable(0)
    try:
        ref = run()                       # k_conv_igemm
        L.cbim_conv_pw_enable(1)
        got = run()                       # k_conv_pw
    finally:
        L.cbim_conv_pw_enable(old)
    n = len(kern) // 2
    assert all(kk not in (4, 12) for kk in kern[:n]) and all(kk in (4, 12) for kk in kern[n:]), f"kernels selected: {kern}"
    names = ["fwd+IN+res", "fwd stats", "raw fwd", "raw stats", "fwd+IN (no stats)", "dgrad", "masked dgrad+acc", "bwd sums",
             "dgrad masked by a", "bwd sums (a)", "wgrad (IN + act on load)", "wgrad (raw input)"]
    for nm, r, g_ in zip(names, ref, got):
        tol = 1e-4 if nm in ("fwd stats", "raw stats", "bwd sums", "bwd sums (a)") or nm.startswith("wgrad") else 4e-3   # bf16 outputs: one rounding apart
        scale = float(r.abs().max()) + 1e-12
        err = float((g_ - r).abs().max()) / scale
        assert err < tol, f"pw vs igemm: {nm} {err:.3e}"
    # against torch in fp32 on the bf16-rounded operands
    xr, wr = from_cl(xl.float().cpu()), w.bfloat16().float()
    xh = F.instance_norm(xr, eps=ops.IN_EPS)
    a = {"relu": torch.relu, "none": lambda t: t, "gelu": F.gelu}[act](xh).bfloat16().float()
    yr = F.conv3d(a, wr) + from_cl(resl.float().cpu())
    assert relerr(from_cl(got[0]), yr) < 1e-2, "fwd vs torch"
    assert relerr(got[1][..., 0], yr.mean((2, 3, 4))) < 2e-3
    assert relerr(from_cl(got[2]), F.conv3d(xr, wr)) < 1e-2, "raw fwd vs torch"
    gr = F.conv_transpose3d(from_cl(dyl.float().cpu()), wr)
    assert relerr(from_cl(got[5]), gr) < 1e-2, "dgrad vs torch"
    dyr = from_cl(dyl.float().cpu())
    dwr = torch.einsum("ncdhw,nkdhw->ck", dyr, a)
    assert relerr(got[10].reshape(Cout, Cin), dwr) < 2e-3, "wgrad vs torch"
    assert relerr(got[11].reshape(Cout, Cin), torch.einsum("ncdhw,nkdhw->ck", dyr, xr)) < 2e-3, "raw wgrad vs torch"


def check_token_linear(dev, rows=300, Cin=48, Cout=144, seed=91):
    """cbim_token_linear / cbim_token_linear_wgrad (conv_pw.hip, token mode) against torch in fp32 on the bf16-rounded
    operands: plain Linear + bias (bf16 and fp32 rows in, bf16 and fp32 out), GELU on load + fp32 residual (the MLP's second
    Linear), the input gradient with the GELU' mask at the stored pre-activation, the weight gradient with either operand in
    fp32 and with GELU on load; and the autograd Function end to end against F.linear / F.gelu."""
    import torch.nn.functional as F
    from cbim_amd import functional as Fn
    BF = torch.bfloat16
    torch.manual_seed(seed)
    rt = lambda t: t.to(BF).float()                      # what the kernel sees of an fp32 operand
    x32 = torch.randn(rows, Cin) * 1.3
    xb = x32.to(BF)
    w = torch.randn(Cout, Cin) * 0.2
    b = torch.randn(Cout)
    res = torch.randn(rows, Cout) * 3.0
    h = (torch.randn(rows, Cout) * 1.5).to(BF)          # a pre-activation tensor for the mask
    g = ops.linear_geom(Cin, Cout)
    wdev = w.to(dev)
    wp, wpd = ops.pack_weights(wdev.view(Cout, Cin, 1, 1, 1), g, 0), ops.pack_weights(wdev.view(Cout, Cin, 1, 1, 1), g, 1)
    wr = rt(w)

    def close(got, ref, tol, what):
        err = float((got.float().cpu() - ref).abs().max() / ref.abs().max())
        assert err < tol, f"token_linear {what}: {err:.3e} (rows {rows}, {Cin} -> {Cout})"

    # forward: bf16 rows, bias, bf16 out / fp32 out
    ref = xb.float() @ wr.t() + b
    close(ops.token_linear(xb.to(dev), wp, b.to(dev), Cout), ref, 6e-3, "bf16 -> bf16")
    close(ops.token_linear(xb.to(dev), wp, b.to(dev), Cout, out_dtype=torch.float32), ref, 2e-5, "bf16 -> fp32")
    close(ops.token_linear(xb.to(dev), wp, None, Cout, out_dtype=torch.float32), xb.float() @ wr.t(), 2e-5, "no bias")
    # fp32 rows (split into bf16 hi + lo in registers: ~2^-17 on the row side)
    close(ops.token_linear(x32.to(dev), wp, b.to(dev), Cout, out_dtype=torch.float32), x32 @ wr.t() + b, 3e-5, "fp32 rows (hi + lo)")
    # fp32 rows + the residue weight image: the product at fp32 accuracy (three MFMAs per fragment pair)
    wl0 = torch.empty_like(wp)
    wl1 = torch.empty_like(wpd)
    from cbim_amd import _lib as _L
    import ctypes as _C
    _L.check(_L.lib().cbim_conv3d_pack_weights_lo(_C.byref(g.fwd), _C.c_void_p(wdev.data_ptr()), _C.c_void_p(wl0.data_ptr()),
                                                  _C.c_void_p(wl1.data_ptr()), None if dev == "cpu" else _C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack lo")
    close(ops.token_linear(x32.to(dev), wp, b.to(dev), Cout, out_dtype=torch.float32, w_lo=wl0), x32 @ w.t() + b, 3e-5, "exact: fp32 rows, hi + lo weights")
    dye = torch.randn(rows, Cout)
    close(ops.token_linear(dye.to(dev), wpd, None, Cin, out_dtype=torch.float32, w_lo=wl1), dye @ w, 3e-5, "exact: input gradient")
    # GELU on load + fp32 residual (the activated tensor is rounded to bf16 like any MFMA operand)
    ref = rt(F.gelu(xb.float())) @ wr.t() + b + res
    close(ops.token_linear(xb.to(dev), wp, b.to(dev), Cout, act_in=ops.ACT["gelu"], res=res.to(dev), out_dtype=torch.float32), ref, 2e-5,
          "gelu on load + residual")
    # input gradient: dy (fp32 rows) @ W, times gelu'(mask) — the dgrad image of the FORWARD weight, roles of Cin / Cout swapped
    dy32 = torch.randn(rows, Cout)
    hin = (torch.randn(rows, Cin) * 1.5).to(BF)
    hh = hin.float().requires_grad_(True)
    F.gelu(hh).backward(torch.ones_like(hh))
    ref = (dy32 @ wr) * hh.grad
    close(ops.token_linear(dy32.to(dev), wpd, None, Cin, mask=hin.to(dev), mask_act=ops.ACT["gelu"]), ref, 6e-3, "dgrad * gelu'(h)")
    close(ops.token_linear(dy32.to(BF).to(dev), wpd, None, Cin), rt(dy32) @ wr, 6e-3, "dgrad bf16 rows")
    # weight gradient: operands in bf16 / fp32, GELU on load
    dyb = dy32.to(BF)
    close(ops.token_linear_wgrad(xb.to(dev), dyb.to(dev)), dyb.float().t() @ xb.float(), 2e-5, "wgrad bf16 / bf16")
    close(ops.token_linear_wgrad(xb.to(dev), dy32.to(dev)), rt(dy32).t() @ xb.float(), 2e-5, "wgrad bf16 x, fp32 dy")
    close(ops.token_linear_wgrad(x32.to(dev), dy32.to(dev)), rt(dy32).t() @ rt(x32), 2e-5, "wgrad fp32 / fp32")
    close(ops.token_linear_wgrad(xb.to(dev), dy32.to(dev), act_in=ops.ACT["gelu"]), rt(dy32).t() @ rt(F.gelu(xb.float())), 2e-5,
          "wgrad gelu on load")
    # the autograd Function: an MLP half block  x + W2 gelu(W1 y + b1) + b2  end to end against torch
    hid = 2 * Cout
    w1 = torch.nn.Parameter((torch.randn(hid, Cin) * 0.2).to(dev))
    b1 = torch.nn.Parameter(torch.randn(hid).to(dev))
    w2 = torch.nn.Parameter((torch.randn(Cin, hid) * 0.2).to(dev))
    b2 = torch.nn.Parameter(torch.randn(Cin).to(dev))
    yb = xb.detach().clone().to(dev).requires_grad_(True)
    stream = (torch.randn(rows, Cin) * 2).to(dev).requires_grad_(True)
    hpre = Fn.token_linear(yb, w1, b1)
    out = Fn.token_linear(hpre, w2, b2, act_in=ops.ACT["gelu"], res=stream, out_dtype=torch.float32)
    gout = torch.randn(rows, Cin)
    out.backward(gout.to(dev))
    yr = xb.detach().float().requires_grad_(True)
    sr = stream.detach().cpu().clone().requires_grad_(True)
    p = [t.detach().cpu().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    hr = rt((yr @ rt(p[0]).t() + p[1]).detach())                                  # the stored bf16 pre-activation
    hr = hr + ((yr @ rt(p[0]).t() + p[1]) - (yr @ rt(p[0]).t() + p[1]).detach())  # (straight-through: gradients as in fp32)
    outr = sr + F.gelu(hr) @ rt(p[2]).t() + p[3]
    outr.backward(gout)
    close(out.detach(), outr.detach(), 2e-2, "MLP half block forward")
    for nm, got, refg, tol in (("dW1", w1.grad, p[0].grad, 2e-2), ("db1", b1.grad, p[1].grad, 2e-2), ("dW2", w2.grad, p[2].grad, 2e-2),
                               ("db2", b2.grad, p[3].grad, 1e-4), ("d stream", stream.grad, sr.grad, 1e-6), ("dy", yb.grad, yr.grad, 3e-2)):
        close(got, refg, tol, "MLP half block " + nm)


def check_batchnorm_affine(dev, dtype, N=2, C=16, dhw=(4, 6, 8), act="elu"):
    """functional.BatchNormActFn on the affine norm kernels (round 5) against F.batch_norm + activation: batch statistics with
    gamma in {1.2, 1e-3, -0.5, 0} and beta != 0 (the folded form divided by gamma), running-statistics update,
    and the running-statistics (eval) mode of nn.BatchNorm3d."""
    from cbim_amd import functional as Fn
    torch.manual_seed(21)
    x = torch.randn(N, C, *dhw) * 1.7 + 0.6
    xl = to_cl(x, dtype).to(dev)
    xr = from_cl(xl.cpu()).requires_grad_(True)
    gam = torch.tensor(([1.2, 1e-3, -0.5, 0.0] * ((C + 3) // 4))[:C])
    bet = torch.linspace(-0.8, 0.9, C)
    fa = {"elu": F.elu, "relu": F.relu, "none": lambda t: t}[act]
    code = ops.ACT[act if act != "none" else None]
    g = torch.randn(N, C, *dhw)
    gl = to_cl(g, dtype).to(dev)
    for batch in (True, False):
        rm, rv = torch.linspace(-0.3, 0.4, C), torch.linspace(0.5, 2.0, C)
        rm_e, rv_e = rm.clone().to(dev), rv.clone().to(dev)
        wr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        if xr.grad is not None:
            xr.grad = None
        ref = fa(F.batch_norm(xr, rm, rv, wr, br, batch, 0.1, 1e-5))
        ref.backward(from_cl(gl.cpu()))
        we, be = gam.clone().to(dev).requires_grad_(True), bet.clone().to(dev).requires_grad_(True)
        xe = xl.clone().requires_grad_(True)
        y = Fn.BatchNormActFn.apply(xe, we, be, rm_e, rv_e, 0.1, 1e-5, code, batch)
        y.backward(gl)
        t = tol(dtype, 3e-5, 1.5e-2)
        assert relerr(from_cl(y.detach().cpu()), ref.detach()) < t, (batch, relerr(from_cl(y.detach().cpu()), ref.detach()))
        assert relerr(from_cl(xe.grad.cpu()), xr.grad) < tol(dtype, 1e-4, 2e-2), (batch, relerr(from_cl(xe.grad.cpu()), xr.grad))
        assert relerr(we.grad.cpu(), wr.grad) < tol(dtype, 1e-4, 2e-2) and relerr(be.grad.cpu(), br.grad) < tol(dtype, 1e-4, 2e-2)
        assert relerr(rm_e.cpu(), rm) < 1e-5 and relerr(rv_e.cpu(), rv) < tol(dtype, 1e-5, 1e-2)
        assert bool(torch.isfinite(xe.grad).all())


def check_dual_raw_conv(dev, dtype, N=1, Cin=16, Cout=8, dhw=(4, 8, 8)):
    """functional.DualRawConvFn (conv1 k^3 | conv3 1x1x1 of a monai UnetResBlock reading the same tensor, the two input
    gradients accumulated in the kernel) against the two separate NormConvFn nodes it replaces and against torch."""
    from cbim_amd import functional as Fn
    torch.manual_seed(61)
    x = torch.randn(N, Cin, *dhw)
    w1 = torch.nn.Parameter((torch.randn(Cout, Cin, 3, 3, 3) * 0.1).to(dev))
    w3 = torch.nn.Parameter((torch.randn(Cout, Cin, 1, 1, 1) * 0.3).to(dev))
    g1, g3 = torch.randn(N, Cout, *dhw), torch.randn(N, Cout, *dhw)
    xl = to_cl(x, dtype).to(dev)
    res = []
    for dual in (True, False):
        xe = xl.clone().requires_grad_(True)
        for p in (w1, w3):
            p.grad = None
        if dual:
            z1, s1, r, s3 = Fn.DualRawConvFn.apply(xe, w1, w3, 1e-5)
        else:
            z1, s1 = Fn.NormConvFn.apply(xe, None, w1, 0, None, True, None, 1e-5)
            r, s3 = Fn.NormConvFn.apply(xe, None, w3, 0, None, True, None, 1e-5)
        (z1.float() * to_cl(g1, dtype).to(dev).float()).sum().backward(retain_graph=True)
        (r.float() * to_cl(g3, dtype).to(dev).float()).sum().backward()
        res.append([t.detach().float().cpu() for t in (z1, s1, r, s3, xe.grad, w1.grad, w3.grad)])
    for a, b, nm in zip(res[0], res[1], ("z1", "stats1", "r", "stats3", "dx", "dw1", "dw3")):
        assert relerr(a, b) < tol(dtype, 1e-6, 8e-3), (nm, relerr(a, b))       # (dx: one bf16 rounding of the sum instead of two)
    xr = from_cl(xl.cpu()).requires_grad_(True)
    w1r, w3r = w1.detach().cpu().clone().requires_grad_(True), w3.detach().cpu().clone().requires_grad_(True)
    if dtype == torch.bfloat16:
        w1q, w3q = w1r.bfloat16().float(), w3r.bfloat16().float()
    else:
        w1q, w3q = w1r, w3r
    (F.conv3d(xr, w1q, None, 1, 1) * from_cl(to_cl(g1, dtype))).sum().backward()
    (F.conv3d(xr, w3q, None, 1, 0) * from_cl(to_cl(g3, dtype))).sum().backward()
    assert relerr(from_cl(res[0][4].to(dtype) if False else res[0][4]), xr.grad) < tol(dtype, 2e-5, 1e-2)

"""Op-level parity of every HIP entry point (through the C-ABI) against stock torch CPU fp32 ops.

torch is the platform here, not the reference: these checks pin each kernel's arithmetic
 before the model/step-level parity tests use the oracle + golden fixtures.
Tolerances: 1e-3 is the north-star bar for logits/losses; op-level checks use tighter bounds
(fp32 with a different summation order: ~1e-5 relative).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from mis_hip import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _as5(t):
    return t if t.dim() == 5 else t.unsqueeze(2)


def _close(a, b, rtol=2e-4, atol=2e-5):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref scale {ref:.3e}"


CONV_CASES = [
    # N, Cin, Cout, D, H, W, k
    (2, 1, 16, 1, 32, 32, (3, 3)),       # first layer of the 2-D UNet: conv_fwd_cin1_kernel<1>, wgrad_cin1_kernel<1>
    (3, 1, 16, 1, 64, 96, (3, 3)),
    (2, 1, 16, 1, 40, 32, (3, 3)),       # H not a multiple of 32: the generic kernels
    (2, 16, 16, 1, 64, 64, (3, 3)),
    (1, 32, 64, 1, 16, 16, (3, 3)),
    (2, 48, 32, 1, 32, 48, (3, 3)),
    (1, 16, 4, 1, 32, 32, (3, 3)),
    (3, 16, 4, 1, 72, 100, (3, 3)),      # Cout <= 4: the 4x4x1-MFMA kernel, ragged tiles in both directions
    (2, 12, 3, 1, 40, 64, (3, 3)),
    (2, 4, 2, 1, 16, 130, (3, 3)),       # ... also taken by the data gradient (2 -> 4 channels)
    (2, 64, 32, 1, 16, 16, (1, 1)),
    (1, 256, 128, 1, 16, 16, (1, 1)),
    (1, 1, 16, 16, 16, 16, (3, 3, 3)),
    (2, 16, 16, 16, 16, 16, (3, 3, 3)),
    (1, 48, 16, 8, 16, 32, (3, 3, 3)),
    (1, 32, 64, 8, 8, 8, (3, 3, 3)),
    (1, 64, 32, 12, 12, 12, (3, 3, 3)),
    (1, 128, 48, 6, 6, 6, (3, 3, 3)),
    (1, 16, 2, 16, 16, 16, (1, 1, 1)),
    (1, 20, 24, 4, 10, 12, (3, 3, 3)),   # ragged everything
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    ops = _ops()
    N, Cin, Cout, D, H, W, k = case
    three_d = len(k) == 3
    x = _rand(N, Cin, D, H, W, seed=1) if three_d else _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, *k, seed=2, scale=0.2)
    b = _rand(Cout, seed=3)
    pad = tuple(kk // 2 for kk in k)
    x.requires_grad_(True)
    w.requires_grad_(True)
    conv = F.conv3d if three_d else F.conv2d
    y_ref = conv(x, w, b, padding=pad)
    dy = _rand(*y_ref.shape, seed=4)
    y_ref.backward(dy)

    xd = _as5(x.detach()).cuda().contiguous()
    wd = w.detach().cuda().contiguous()
    bd = b.cuda()
    dyd = _as5(dy).cuda().contiguous()
    y = torch.empty(N, Cout, D, H, W, device="cuda")
    wp = ops.conv_pack(wd, 0)
    ops.conv_fwd(xd, wp, bd, y, Cin, Cout, k)
    _close(y, _as5(y_ref))

    # data gradient = conv of dy with the flipped/transposed pack
    wpd = ops.conv_pack(wd, 1)
    dx = torch.empty(N, Cin, D, H, W, device="cuda")
    ops.conv_fwd(dyd, wpd, None, dx, Cout, Cin, k)
    _close(dx, _as5(x.grad))

    dw = torch.empty_like(wd)
    ops.conv_wgrad(xd, dyd, dw, k)
    _close(dw, w.grad, rtol=3e-4, atol=1e-4)


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,k,per_sample", [
    (3, 16, 32, 1, 32, 64, (3, 3), False), (2, 8, 16, 1, 128, 128, (3, 3), False),
    (2, 16, 16, 8, 16, 16, (3, 3, 3), True), (2, 32, 32, 8, 16, 32, (3, 3, 3), True),
    (2, 1, 16, 8, 16, 32, (3, 3, 3), True),       # first layer: conv_fwd_cin1_kernel<3>
    (3, 1, 16, 1, 64, 48, (3, 3), False)])        # ... of the 2-D UNet: conv_fwd_cin1_kernel<1>
def test_conv_fused_statistics(N, Cin, Cout, D, H, W, k, per_sample):
    """mis_conv_fwd_stats + mis_norm_stats_finalize == conv followed by the statistics of BatchNorm / InstanceNorm."""
    ops = _ops()
    T = ops.conv_stat_tiles(N, Cin, Cout, D, H, W, k)
    assert T > 0
    x = _rand(N, Cin, D, H, W, seed=21)
    w = _rand(Cout, Cin, *k, seed=22, scale=0.3)
    b = _rand(Cout, seed=23)
    conv = F.conv3d if len(k) == 3 else F.conv2d
    y_ref = conv(x if len(k) == 3 else x[:, :, 0], w, b, padding=tuple(kk // 2 for kk in k))
    y_ref = _as5(y_ref)
    part = torch.full((Cout * N * T, 2), float("nan"), device="cuda")
    y = torch.empty(N, Cout, D, H, W, device="cuda")
    strides = (T, Cout * T) if per_sample else (N * T, T)
    ops.conv_fwd(x.cuda(), ops.conv_pack(w.cuda(), 0), b.cuda(), y, Cin, Cout, k, stat=(part, *strides))
    _close(y, y_ref)
    G = N * Cout if per_sample else Cout
    mean = torch.empty(G, device="cuda"); rstd = torch.empty(G, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    ops.norm_stats_finalize(part, N, Cout, D * H * W, T, per_sample, 1e-5, mean, rstd,
                            None if per_sample else rm, None if per_sample else rv, None if per_sample else nbt)
    dims = (2, 3, 4) if per_sample else (0, 2, 3, 4)
    m_ref = y_ref.double().mean(dim=dims).flatten()
    v_ref = y_ref.double().var(dim=dims, unbiased=False).flatten()
    _close(mean, m_ref, rtol=1e-5, atol=1e-6)
    _close(rstd, 1.0 / torch.sqrt(v_ref + 1e-5), rtol=1e-5, atol=1e-6)
    if not per_sample:
        n = N * D * H * W
        _close(rm, 0.1 * m_ref, rtol=1e-5, atol=1e-6)
        _close(rv, 0.9 + 0.1 * v_ref * n / (n - 1), rtol=1e-5, atol=1e-6)
        assert int(nbt) == 1
    # ineligible geometry (ragged tiles): the query says so and the fused entry point refuses
    assert ops.conv_stat_tiles(1, 16, 16, 1, 30, 30, (3, 3)) == 0


def test_conv_batch_strided_views():
    """Producers/consumers address channel slices of a concat buffer (no torch.cat on the hot path)."""
    ops = _ops()
    N, C1, C2, H, W = 2, 16, 16, 32, 32
    cat = _rand(N, C1 + C2, 1, H, W, seed=5).cuda()
    w = _rand(32, C2, 3, 3, seed=6, scale=0.2).cuda()
    ycat = torch.zeros(N, 64, 1, H, W, device="cuda")
    ops.conv_fwd(cat[:, C1:], ops.conv_pack(w, 0), None, ycat[:, 32:], C2, 32, (3, 3))
    ref = F.conv2d(cat[:, C1:, 0].cpu(), w.cpu(), padding=1)
    _close(ycat[:, 32:, 0], ref)
    assert ycat[:, :32].abs().max().item() == 0.0


@pytest.mark.parametrize("per_sample,slope,shape", [
    (False, 0.01, (4, 16, 1, 32, 32)),
    (False, 0.01, (3, 32, 1, 16, 16)),
    (True, 0.0, (2, 16, 16, 16, 16)),
    (True, 0.0, (2, 8, 6, 6, 6)),
])
def test_norm_act_fwd_bwd(per_sample, slope, shape):
    ops = _ops()
    N, C = shape[0], shape[1]
    x = _rand(*shape, seed=7, scale=2.0) + 0.3
    x.requires_grad_(True)
    gamma = (1 + 0.1 * _rand(C, seed=8)).requires_grad_(not per_sample)
    beta = (0.1 * _rand(C, seed=9)).requires_grad_(not per_sample)
    rm, rv = torch.zeros(C), torch.ones(C)
    if per_sample:
        z = F.instance_norm(x, eps=1e-5)
        a_ref = F.relu(z)
    else:
        z = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
        a_ref = F.leaky_relu(z, slope)
    da = _rand(*shape, seed=10)
    a_ref.backward(da)

    xd = x.detach().cuda()
    G = N * C if per_sample else C
    mean = torch.empty(G, device="cuda")
    rstd = torch.empty(G, device="cuda")
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.norm_stats(xd, per_sample, 1e-5, mean, rstd, None if per_sample else rmd, None if per_sample else rvd,
                   None if per_sample else nbt)
    a = torch.empty(*shape, device="cuda")
    g = None if per_sample else gamma.detach().cuda()
    bt = None if per_sample else beta.detach().cuda()
    ops.norm_act_fwd(xd, a, per_sample, mean, rstd, g, bt, slope)
    _close(a, a_ref)
    if not per_sample:
        _close(rmd, rm, atol=1e-6)
        _close(rvd, rv, atol=1e-6)
        assert int(nbt.item()) == 1
    dx = torch.empty(*shape, device="cuda")
    dg = None if per_sample else torch.empty(C, device="cuda")
    db = None if per_sample else torch.empty(C, device="cuda")
    ops.norm_act_bwd(xd, da.cuda(), dx, per_sample, mean, rstd, g, bt, slope, dgamma=dg, dbeta=db)
    _close(dx, x.grad, rtol=5e-4, atol=1e-5)
    if not per_sample:
        _close(dg, gamma.grad, rtol=5e-4, atol=1e-4)
        _close(db, beta.grad, rtol=5e-4, atol=1e-4)


@pytest.mark.parametrize("post", [False, True])
@pytest.mark.parametrize("per_sample,shape", [(True, (2, 16, 8, 8, 8)), (True, (3, 8, 6, 6, 6)), (False, (4, 16, 1, 32, 32))])
def test_norm_residual_act_fwd_bwd(per_sample, shape, post):
    """mis_norm_res_act_{fwd,bwd}: lrelu(norm(x) + res) -- MONAI's UnetResBlock tail (UNETR / SwinUNETR) -- in one pass,
    against torch autograd; the shortcut is a channel slice of a wider buffer (batch-strided view) and its gradient is
    written or accumulated."""
    ops = _ops()
    N, C = shape[0], shape[1]
    x = (_rand(*shape, seed=17, scale=2.0) + 0.3).requires_grad_(True)
    wide = _rand(N, C + 8, *shape[2:], seed=18)
    r = wide[:, 8:].clone().requires_grad_(True)
    gamma = (1 + 0.1 * _rand(C, seed=8)).requires_grad_(not per_sample)
    beta = (0.1 * _rand(C, seed=9)).requires_grad_(not per_sample)
    z = F.instance_norm(x, eps=1e-5) if per_sample else \
        F.batch_norm(x, torch.zeros(C), torch.ones(C), gamma, beta, training=True, momentum=0.1, eps=1e-5)
    ref = F.leaky_relu(z, 0.01) + r if post else F.leaky_relu(z + r, 0.01)      # post: V-Net's x_up + skip
    dy = _rand(*shape, seed=19)
    ref.backward(dy)

    xd, wd = x.detach().cuda(), wide.cuda()
    rd = wd[:, 8:]
    G = N * C if per_sample else C
    mean, rstd = torch.empty(G, device="cuda"), torch.empty(G, device="cuda")
    ops.norm_stats(xd, per_sample, 1e-5, mean, rstd, None, None, None)
    g = None if per_sample else gamma.detach().cuda()
    bt = None if per_sample else beta.detach().cuda()
    y = torch.full(shape, float("nan"), device="cuda")
    ops.norm_res_act_fwd(xd, rd, y, per_sample, mean, rstd, g, bt, 0.01, post=post)
    _close(y, ref)
    dx = torch.full(shape, float("nan"), device="cuda")
    dwide = torch.full(tuple(wide.shape), float("nan"), device="cuda")
    dr = dwide[:, 8:]
    dg = None if per_sample else torch.empty(C, device="cuda")
    db = None if per_sample else torch.empty(C, device="cuda")
    ops.norm_res_act_bwd(xd, rd, dy.cuda(), dx, dr, False, per_sample, mean, rstd, g, bt, 0.01, dgamma=dg, dbeta=db,
                         post=post)
    _close(dx, x.grad, rtol=5e-4, atol=1e-5)
    _close(dr, r.grad, rtol=1e-6, atol=1e-7)
    assert torch.isnan(dwide[:, :8]).all()          # nothing outside the slice is touched
    if not per_sample:
        _close(dg, gamma.grad, rtol=5e-4, atol=1e-4)
        _close(db, beta.grad, rtol=5e-4, atol=1e-4)
    ops.norm_res_act_bwd(xd, rd, dy.cuda(), dx, dr, True, per_sample, mean, rstd, g, bt, 0.01, dgamma=dg, dbeta=db,
                         post=post)
    _close(dr, 2 * r.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape,cg", [((2, 32, 4, 8, 8), 2), ((2, 64, 2, 8, 8), 4), ((3, 16, 4, 4, 8), 1)])
def test_group_norm_act_fwd_bwd(shape, cg):
    """nn.GroupNorm(16, C) + ReLU (reference code/networks/vnet.py:19-20) against torch CPU: statistics per
    (sample, group of cg = C/16 channels), per-channel affine, dx / dgamma / dbeta."""
    ops = _ops()
    N, C = shape[0], shape[1]
    assert C // cg == 16
    x = _rand(*shape, seed=21, scale=2.0) + 0.3
    x.requires_grad_(True)
    gamma = (1 + 0.3 * _rand(C, seed=22)).requires_grad_(True)
    beta = (0.2 * _rand(C, seed=23)).requires_grad_(True)
    a_ref = F.relu(F.group_norm(x, 16, gamma, beta, 1e-5))
    da = _rand(*shape, seed=24)
    a_ref.backward(da)
    xd = x.detach().cuda()
    mean = torch.empty(N * 16, device="cuda")
    rstd = torch.empty(N * 16, device="cuda")
    ops.group_norm_stats(xd, cg, 1e-5, mean, rstd)
    xg = x.detach().reshape(N, 16, -1)
    _close(mean, xg.mean(2).flatten(), atol=1e-5)
    _close(rstd, (xg.var(2, unbiased=False) + 1e-5).rsqrt().flatten(), rtol=1e-4)
    a = torch.empty(*shape, device="cuda")
    g, bt = gamma.detach().cuda(), beta.detach().cuda()
    ops.norm_act_fwd(xd, a, True, mean, rstd, g, bt, 0.0, cg=cg)
    _close(a, a_ref)
    dx = torch.empty(*shape, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.norm_act_bwd(xd, da.cuda(), dx, True, mean, rstd, g, bt, 0.0, dgamma=dg, dbeta=db, cg=cg)
    _close(dx, x.grad, rtol=5e-4, atol=1e-5)
    _close(dg, gamma.grad, rtol=5e-4, atol=1e-4)
    _close(db, beta.grad, rtol=5e-4, atol=1e-4)


def test_act_only_fwd_bwd():
    """normalization='none' blocks (reference vnet.py:22,84): ReLU (+ dropout) without statistics."""
    ops = _ops()
    shape = (2, 8, 4, 8, 8)
    x = _rand(*shape, seed=31)
    da = _rand(*shape, seed=32)
    xd = x.cuda()
    mean, rstd = torch.zeros(8, device="cuda"), torch.ones(8, device="cuda")
    a = torch.empty(*shape, device="cuda")
    ops.norm_act_fwd(xd, a, False, mean, rstd, None, None, 0.0)
    assert torch.equal(a.cpu(), F.relu(x))
    dx = torch.empty(*shape, device="cuda")
    ops.norm_act_bwd(xd, da.cuda(), dx, False, mean, rstd, None, None, 0.0, no_norm=True)
    assert torch.equal(dx.cpu(), da * (x > 0).float())


def test_norm_act_dropout_mask_injected_and_philox():
    ops = _ops()
    shape = (2, 16, 1, 32, 32)
    x = _rand(*shape, seed=11).cuda()
    C = 16
    mean = torch.empty(C, device="cuda"); rstd = torch.empty(C, device="cuda")
    ops.norm_stats(x, False, 1e-5, mean, rstd)
    p = 0.3
    mask = (torch.rand(*shape, generator=torch.Generator().manual_seed(1)) >= p).float() / (1 - p)
    a0 = torch.empty(*shape, device="cuda"); a1 = torch.empty(*shape, device="cuda")
    ops.norm_act_fwd(x, a0, False, mean, rstd, None, None, 0.01)
    ops.norm_act_fwd(x, a1, False, mean, rstd, None, None, 0.01, drop_p=p, drop_mask=mask.cuda())
    _close(a1, a0.cpu() * mask)
    # Philox path: keep-rate, scaling, determinism per (seed, offset, salt), fwd/bwd mask agreement
    st = ops.new_step_state()
    ops.step_init(st, 1337, 0, 0.01, 30000, 0.99, 0.1, 200.0)
    a2 = torch.empty(*shape, device="cuda"); a3 = torch.empty(*shape, device="cuda")
    ops.norm_act_fwd(x, a2, False, mean, rstd, None, None, 0.01, drop_p=p, drop_salt=5, state=st)
    ops.norm_act_fwd(x, a3, False, mean, rstd, None, None, 0.01, drop_p=p, drop_salt=5, state=st)
    assert torch.equal(a2, a3)
    kept = (a2 != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.02
    nz = a2 != 0
    _close(a2[nz], (a0 / (1 - p))[nz])
    ops.norm_act_fwd(x, a3, False, mean, rstd, None, None, 0.01, drop_p=p, drop_salt=6, state=st)
    assert not torch.equal(a2, a3)
    # backward regenerates the same mask: dx must vanish exactly where the forward dropped
    da = torch.ones(*shape, device="cuda")
    dx_nodrop = torch.empty(*shape, device="cuda"); dx = torch.empty(*shape, device="cuda")
    ops.norm_act_bwd(x, da * nz.float() / (1 - p), dx_nodrop, False, mean, rstd, None, None, 0.01)
    ops.norm_act_bwd(x, da, dx, False, mean, rstd, None, None, 0.01, drop_p=p, drop_salt=5, state=st)
    _close(dx, dx_nodrop, rtol=1e-5, atol=1e-6)
    ops.step_advance(st, 0.01, 30000, 0.99, 0.1, 200.0)
    ops.norm_act_fwd(x, a3, False, mean, rstd, None, None, 0.01, drop_p=p, drop_salt=5, state=st)
    assert not torch.equal(a2, a3)


@pytest.mark.parametrize("shape", [(2, 8, 1, 32, 32), (2, 4, 8, 16, 16), (1, 3, 4, 8, 24), (1, 2, 6, 12, 12),
                                   (2, 2, 4, 4, 8)])
def test_maxpool(shape):
    ops = _ops()
    x = _rand(*shape, seed=12)
    x.requires_grad_(True)
    three_d = shape[2] > 1
    y_ref = F.max_pool3d(x, 2) if three_d else F.max_pool2d(x[:, :, 0], 2).unsqueeze(2)
    dy = _rand(*y_ref.shape, seed=13)
    y_ref.backward(dy)
    xd = x.detach().cuda()
    y = torch.empty(*y_ref.shape, device="cuda")
    idx = torch.empty(y.numel(), dtype=torch.uint8, device="cuda")
    ops.maxpool2_fwd(xd, y, idx)
    assert torch.equal(y.cpu(), y_ref.detach())
    dx = torch.full(shape, 1.0, device="cuda")
    ops.maxpool2_bwd(dy.cuda(), idx, dx, accumulate=False)
    assert torch.equal(dx.cpu(), x.grad)
    ops.maxpool2_bwd(dy.cuda(), idx, dx, accumulate=True)
    assert torch.equal(dx.cpu(), 2 * x.grad)


def test_maxpool3d_ties_and_strided_views():
    """First maximum of a window wins (torch semantics) also in the 4-outputs-per-thread kernels; inputs / outputs may be
    channel slices of a wider buffer; accumulate adds to what is there."""
    ops = _ops()
    x = torch.zeros(1, 2, 4, 4, 16)                     # all ties
    x[0, 1, 0, 1, 3] = 1.0
    x.requires_grad_(True)
    y_ref = F.max_pool3d(x, 2)
    dy = _rand(*y_ref.shape, seed=41)
    y_ref.backward(dy)
    wide = torch.zeros(1, 5, 4, 4, 16, device="cuda")
    wide[:, 2:4] = x.detach().cuda()
    ywide = torch.zeros(1, 3, 2, 2, 8, device="cuda")
    idx = torch.empty(y_ref.numel(), dtype=torch.uint8, device="cuda")
    ops.maxpool2_fwd(wide[:, 2:4], ywide[:, 1:3], idx)
    assert torch.equal(ywide[:, 1:3].cpu(), y_ref.detach()) and float(ywide[:, 0].abs().max()) == 0.0
    dxw = torch.ones(1, 5, 4, 4, 16, device="cuda")
    ops.maxpool2_bwd(dy.cuda(), idx, dxw[:, 2:4], accumulate=True)
    assert torch.equal(dxw[:, 2:4].cpu(), 1.0 + x.grad) and float((dxw[:, :2] - 1).abs().max()) == 0.0


@pytest.mark.parametrize("shape,align", [((2, 8, 1, 16, 16), True), ((1, 4, 1, 8, 24), True),
                                         ((2, 4, 6, 6, 6), False), ((1, 3, 8, 4, 12), False),
                                         ((1, 2, 5, 3, 7), False), ((2, 1, 1, 5, 4), False), ((1, 2, 2, 1, 1), False),
                                         ((1, 2, 12, 12, 12), False), ((1, 2, 1, 5, 7), True), ((2, 3, 1, 1, 2), True),
                                         ((2, 3, 1, 32, 64), True), ((1, 2, 1, 48, 40), True)])     # full / ragged LDS tiles
def test_upsample(shape, align):
    ops = _ops()
    x = _rand(*shape, seed=14)
    x.requires_grad_(True)
    three_d = shape[2] > 1
    if three_d:
        y_ref = F.interpolate(x, scale_factor=(2, 2, 2), mode="trilinear", align_corners=align)
    else:
        y_ref = F.interpolate(x[:, :, 0], scale_factor=2, mode="bilinear", align_corners=align).unsqueeze(2)
    dy = _rand(*y_ref.shape, seed=15)
    y_ref.backward(dy)
    y = torch.empty(*y_ref.shape, device="cuda")
    ops.upsample2_fwd(x.detach().cuda(), y, align)
    _close(y, y_ref, rtol=1e-5, atol=1e-6)
    dx = torch.empty(*shape, device="cuda")
    ops.upsample2_bwd(dy.cuda(), dx, align)
    _close(dx, x.grad, rtol=1e-5, atol=1e-5)
    ops.upsample2_bwd(dy.cuda(), dx, align, accumulate=True)     # second consumer of the same gradient buffer
    _close(dx, 2 * x.grad, rtol=1e-5, atol=2e-5)


def _tail_reference(s, t, label, L, w, gate=True):
    s = s.clone().requires_grad_(True)
    C = s.shape[1]
    ps = torch.softmax(s, dim=1)
    pt = torch.softmax(t, dim=1)
    ce = F.cross_entropy(s[:L], label[:L].long())
    onehot = torch.stack([(label[:L] == c).float() for c in range(C)], dim=1)
    dice = 0.0
    for c in range(C):
        i = (ps[:L, c] * onehot[:, c]).sum()
        y = (onehot[:, c] * onehot[:, c]).sum()
        z = (ps[:L, c] * ps[:L, c]).sum()
        dice = dice + (1 - (2 * i + 1e-5) / (z + y + 1e-5))
    dice = dice / C
    cons = ((ps[L:] - pt) ** 2).mean() if gate else torch.zeros(())
    loss = 0.5 * (dice + ce) + w * cons
    loss.backward()
    return loss.item(), ce.item(), dice.item(), float(cons), s.grad


@pytest.mark.parametrize("C,shape,ldtype", [(4, (1, 32, 32), torch.uint8), (2, (8, 8, 8), torch.int64)])
def test_loss_tail(C, shape, ldtype):
    ops = _ops()
    B, L = 6, 3
    s = _rand(B, C, *shape, seed=16, scale=3.0)
    t = _rand(B - L, C, *shape, seed=17, scale=3.0)
    label = torch.randint(0, C, (B, *shape), generator=torch.Generator().manual_seed(18)).to(ldtype)
    w = 0.0731
    ref = _tail_reference(s, t, label, L, w)
    sd, td = s.cuda(), t.cuda()
    out = torch.zeros(16, device="cuda")
    ds = torch.empty_like(sd)
    ops.loss_tail(sd, td, label[:L].contiguous().cuda(), L, out, ds, cons_weight=w)
    o = out.cpu()
    for i in range(4):
        assert abs(o[i].item() - ref[i]) <= 1e-5 + 1e-5 * abs(ref[i]), (i, o[i].item(), ref[i])
    _close(ds, ref[4], rtol=1e-4, atol=1e-9)


def test_sgd_ema_and_schedule():
    ops = _ops()
    import math
    n = 100003
    p = _rand(n, seed=19); g = _rand(n, seed=20); m = _rand(n, seed=21); e = _rand(n, seed=22)
    lr, mu, wd, alpha = 0.0123, 0.9, 1e-4, 0.97
    d = g + wd * p
    m_ref = mu * m + d
    p_ref = p - lr * m_ref
    e_ref = e * alpha + (1 - alpha) * p_ref
    pd, gd, md, ed = p.cuda(), g.cuda(), m.cuda(), e.cuda()
    ops.sgd_ema_step(pd, gd, md, ed, lr=lr, momentum=mu, weight_decay=wd, ema_alpha=alpha)
    _close(pd, p_ref, rtol=1e-6, atol=1e-7)
    _close(md, m_ref, rtol=1e-6, atol=1e-7)
    _close(ed, e_ref, rtol=1e-6, atol=1e-7)
    # device-side schedule == reference host arithmetic (train_mean_teacher_2D.py:119-128,234-236)
    st = ops.new_step_state()
    base, mx = 0.01, 30000
    for k0 in (0, 1, 999, 1000, 1001, 29999):
        ops.step_init(st, 7, k0, base, mx, 0.99, 0.1, 200.0, 150, 1000)
        for k in (k0, k0 + 1):
            s = ops.read_step_state(st)
            assert s["iter_num"] == k
            lr_ref = base if k == 0 else base * (1.0 - (k - 1) / mx) ** 0.9
            a_ref = min(1 - 1 / (k + 1), 0.99)
            cur = min(max(float(k // 150), 0.0), 200.0)
            w_ref = 0.1 * math.exp(-5.0 * (1.0 - cur / 200.0) ** 2)
            assert abs(s["lr"] - lr_ref) < 1e-9 + 1e-6 * lr_ref
            assert abs(s["ema_alpha"] - a_ref) < 1e-7
            assert abs(s["cons_weight"] - w_ref) < 1e-9 + 1e-6 * w_ref
            assert s["cons_gate"] == (1.0 if k >= 1000 else 0.0)
            ops.step_advance(st, base, mx, 0.99, 0.1, 200.0, 150, 1000)


def test_teacher_noise_and_argmax():
    ops = _ops()
    x = _rand(4, 1, 1, 64, 64, seed=23).cuda()
    st = ops.new_step_state()
    ops.step_init(st, 1337, 5, 0.01, 30000, 0.99, 0.1, 200.0)
    y = torch.empty_like(x)
    ops.teacher_noise(x, y, st)
    e = (y - x).cpu()
    assert e.abs().max().item() <= 0.2 + 1e-6
    inner = e[(e.abs() < 0.199)]
    assert abs(inner.mean().item()) < 0.01
    assert abs(e.std().item() - 0.0968) < 0.01   # std of N(0, 0.1) clipped at 2 sigma
    assert 0.03 < (e.abs() >= 0.1999).float().mean().item() < 0.06   # P(|z|>2) = 4.55 %
    lg = _rand(3, 4, 1, 16, 16, seed=24).cuda()
    out = torch.empty(3 * 256, dtype=torch.uint8, device="cuda")
    ops.argmax_channels(lg, out)
    assert torch.equal(out.cpu().view(3, 1, 16, 16).long(), lg.cpu().argmax(dim=1))


@pytest.mark.parametrize("per_sample,slope,no_norm", [(True, 0.01, False), (False, 0.0, False), (False, 0.01, True)])
def test_norm_act_on_a_volume_of_27_voxels(per_sample, slope, no_norm):
    """Volumes whose voxel count is not a multiple of 4 (3 x 3 x 3: SwinUNETR's deepest level at 96^3) run the scalar
    kernels of norm_act.hip: statistics, apply and backward against torch fp64 (nn.InstanceNorm3d / nn.BatchNorm3d +
    nn.LeakyReLU, MONAI UnetResBlock)."""
    import torch.nn.functional as F
    from mis_hip import ops
    N, C, shape = 3, 8, (3, 3, 3)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(N, C, *shape, generator=g, dtype=torch.float64) * 2 - 1).requires_grad_(True)
    da = torch.rand(N, C, *shape, generator=g, dtype=torch.float64) * 2 - 1
    if no_norm:
        z = x
    elif per_sample:
        z = F.instance_norm(x, eps=1e-5)
    else:
        z = F.batch_norm(x, None, None, None, None, training=True, eps=1e-5)
    y = F.leaky_relu(z, slope)
    y.backward(da)
    xd = x.detach().float().cuda()
    G = C if (no_norm or not per_sample) else N * C
    mean, rstd = torch.zeros(G, device="cuda"), torch.ones(G, device="cuda")
    if not no_norm:
        ops.norm_stats(xd, per_sample, 1e-5, mean, rstd)
    yd = torch.full((N, C) + shape, float("nan"), device="cuda")
    ops.norm_act_fwd(xd, yd, per_sample and not no_norm, mean, rstd, None, None, slope)
    assert (yd.cpu().double() - y.detach()).abs().max().item() <= 2e-5
    dx = torch.full((N, C) + shape, float("nan"), device="cuda")
    ops.norm_act_bwd(xd, da.float().cuda(), dx, per_sample and not no_norm, mean, rstd, None, None, slope, no_norm=no_norm)
    assert (dx.cpu().double() - x.grad).abs().max().item() <= 5e-5 * max(1.0, x.grad.abs().max().item())

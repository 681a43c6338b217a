"""Normalisation passes fused into their neighbours at the two ends of the 3-D networks: the last block with the 1x1x1
classifier (mis_norm_head_fwd / _bwd, norm_act.hip; reference code/networks/unet_3D.py: up_concat1 -> dropout2 -> final)
and the first layer's weight gradient through its norm (mis_norm_act_bwd_sums + mis_conv_wgrad_cin1_norm).  Op-level
against torch autograd in float64 and against the un-fused kernels, and Mean-Teacher steps with the fusions on and off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from mis_hip import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _close(a, b, rtol=2e-5, atol=2e-6):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * b.abs().max().item(), f"max err {err:.3e} vs scale {b.abs().max().item():.3e}"


@pytest.mark.parametrize("K", [2, 3, 4])
@pytest.mark.parametrize("per_sample,slope,shape,p", [(False, 0.0, (3, 16, 4, 8, 16), 0.3), (False, 0.01, (2, 16, 8, 8, 8), 0.0),
                                                      (True, 0.0, (2, 16, 4, 4, 32), 0.5), (False, 0.0, (1, 16, 2, 6, 10), 0.3)])
def test_norm_head_forward_backward(per_sample, slope, shape, p, K):
    ops = _ops()
    N, C, D, H, W = shape
    assert ops.norm_head_eligible(C, K, per_sample, None, None)
    x = _rand(*shape, seed=1, scale=2.0).requires_grad_(True)
    gamma = None if per_sample else (_rand(C, seed=2) * 0.5 + 1.0).requires_grad_(True)
    beta = None if per_sample else (_rand(C, seed=3) * 0.3).requires_grad_(True)
    w = _rand(K, C, 1, 1, 1, seed=4, scale=0.5).requires_grad_(True)
    b = _rand(K, seed=5).requires_grad_(True)
    mask = ((torch.rand(*shape, generator=torch.Generator().manual_seed(6)) >= p).double() / (1 - p)) if p > 0 else None
    if per_sample:
        z = F.instance_norm(x, eps=1e-5)
    else:
        z = F.batch_norm(x, None, None, gamma, beta, training=True, eps=1e-5)
    a = F.leaky_relu(z, slope)
    if mask is not None:
        a = a * mask
    ref = F.conv3d(a, w, b)
    dl = _rand(*ref.shape, seed=7)
    ref.backward(dl)

    xd = x.detach().float().cuda()
    G = N * C if per_sample else C
    mean, rstd = torch.empty(G, device="cuda"), torch.empty(G, device="cuda")
    ops.norm_stats(xd, per_sample, 1e-5, mean, rstd)
    gd = None if gamma is None else gamma.detach().float().cuda()
    bd = None if beta is None else beta.detach().float().cuda()
    wd, hb = w.detach().float().cuda().view(K, C), b.detach().float().cuda()
    md = None if mask is None else mask.float().cuda()
    logits = torch.full((N, K, D, H, W), float("nan"), device="cuda")
    ops.norm_head_fwd(xd, logits, per_sample, mean, rstd, gd, bd, slope, wd, hb, drop_p=p, drop_mask=md)
    _close(logits, ref)

    dx = torch.full(shape, float("nan"), device="cuda")
    dw, db = torch.full((K, C), float("nan"), device="cuda"), torch.full((K,), float("nan"), device="cuda")
    dg = None if gamma is None else torch.full((C,), float("nan"), device="cuda")
    dbt = None if beta is None else torch.full((C,), float("nan"), device="cuda")
    ops.norm_head_bwd(xd, dl.float().cuda(), dx, per_sample, mean, rstd, gd, bd, slope, wd, dw, db, drop_p=p,
                      drop_mask=md, dgamma=dg, dbeta=dbt)
    _close(dx, x.grad, rtol=1e-4, atol=1e-6)
    _close(dw, w.grad.view(K, C), rtol=1e-4)
    _close(db, b.grad, rtol=1e-4)
    if gamma is not None:
        _close(dg, gamma.grad, rtol=1e-4)
        _close(dbt, beta.grad, rtol=1e-4)
    # deterministic, and the accumulate flags add
    dx2, dw2, db2 = torch.empty_like(dx), dw.clone(), db.clone()
    ops.norm_head_bwd(xd, dl.float().cuda(), dx2, per_sample, mean, rstd, gd, bd, slope, wd, dw2, db2, drop_p=p,
                      drop_mask=md, dgamma=None if dg is None else dg.clone(), dbeta=None if dbt is None else dbt.clone(),
                      accumulate_w=True)
    assert torch.equal(dx, dx2)
    _close(dw2, 2 * w.grad.view(K, C), rtol=1e-4)
    _close(db2, 2 * b.grad, rtol=1e-4)


def test_norm_head_philox_matches_unfused_kernels():
    """Device-RNG dropout (element-wise and nn.Dropout3d channel mode): the fused pass draws the same masks as
    mis_norm_act_fwd / _bwd followed by the 1x1x1 conv kernels."""
    ops = _ops()
    _philox_case(2)


@pytest.mark.parametrize("K", [3, 4])
def test_norm_head_philox_more_classes(K):
    _philox_case(K)


def _philox_case(K):
    ops = _ops()
    shape, p = (2, 16, 4, 8, 32), 0.3
    N, C = shape[:2]
    x = _rand(*shape, seed=21, scale=2.0).float().cuda()
    w = _rand(K, C, 1, 1, 1, seed=22, scale=0.5).float().cuda()
    b = _rand(K, seed=23).float().cuda()
    dl = _rand(N, K, *shape[2:], seed=24).float().cuda()
    mean, rstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.norm_stats(x, False, 1e-5, mean, rstd)
    st = ops.new_step_state()
    ops.step_init(st, 1337, 3, 0.01, 30000, 0.99, 0.1, 200.0)
    for salt in (5, 5 | 0x80000000):
        a = torch.empty(shape, device="cuda")
        ops.norm_act_fwd(x, a, False, mean, rstd, None, None, 0.0, drop_p=p, drop_salt=salt, state=st)
        ref = torch.empty(N, K, *shape[2:], device="cuda")
        ops.conv_fwd(a, ops.conv_pack(w, 0), b, ref, C, K, (1, 1, 1))
        logits = torch.empty_like(ref)
        ops.norm_head_fwd(x, logits, False, mean, rstd, None, None, 0.0, w.view(K, C), b, drop_p=p, drop_salt=salt, state=st)
        _close(logits, ref, rtol=1e-5, atol=1e-6)
        da = torch.empty(shape, device="cuda")
        ops.conv_fwd(dl, ops.conv_pack(w, 1), None, da, K, C, (1, 1, 1))
        dx_ref, dx = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
        ops.norm_act_bwd(x, da, dx_ref, False, mean, rstd, None, None, 0.0, drop_p=p, drop_salt=salt, state=st)
        dw_ref = torch.empty(K, C, 1, 1, 1, device="cuda")
        ops.conv_wgrad(a, dl, dw_ref, (1, 1, 1))
        dw, db = torch.empty(K, C, device="cuda"), torch.empty(K, device="cuda")
        ops.norm_head_bwd(x, dl, dx, False, mean, rstd, None, None, 0.0, w.view(K, C), dw, db, drop_p=p, drop_salt=salt,
                          state=st)
        _close(dx, dx_ref, rtol=1e-5, atol=1e-6)
        _close(dw, dw_ref.view(K, C), rtol=1e-5)
        _close(db, dl.sum(dim=(0, 2, 3, 4)), rtol=1e-5)


def test_norm_head_refusals():
    ops = _ops()
    assert not ops.norm_head_eligible(32, 2, False, None, None)
    assert ops.norm_head_eligible(16, 4, False, None, None) and not ops.norm_head_eligible(16, 5, False, None, None)
    assert not ops.norm_head_eligible(16, 2, True, torch.ones(16), None)      # InstanceNorm with affine: GroupNorm kernels
    assert not ops.norm_head_eligible(16, 2, False, None, None, no_norm=True)
    x = torch.zeros(1, 32, 2, 4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.norm_head_fwd(x, torch.zeros(1, 2, 2, 4, 8, device="cuda"), False, torch.zeros(32, device="cuda"),
                          torch.ones(32, device="cuda"), None, None, 0.0, torch.zeros(2, 32, device="cuda"), None)


@pytest.mark.parametrize("per_sample,slope,N,D,H,W", [(False, 0.0, 2, 4, 8, 32), (True, 0.01, 3, 8, 16, 64), (False, 0.0, 1, 4, 8, 96)])
def test_first_layer_weight_gradient_through_the_norm(per_sample, slope, N, D, H, W):
    """mis_norm_act_bwd_sums + mis_conv_wgrad_cin1_norm (gradient at the conv output formed on the load path) ==
    mis_norm_act_bwd + mis_conv_wgrad, and == torch autograd in float64 (reference unet_3D.py:28 conv1 -> norm -> ReLU)."""
    ops = _ops()
    C = 16
    assert ops.conv_wgrad_cin1_norm_eligible(N, C, D, H, W)
    x = _rand(N, 1, D, H, W, seed=41).requires_grad_(False)
    w = _rand(C, 1, 3, 3, 3, seed=42, scale=0.5).requires_grad_(True)
    gamma = None if per_sample else (_rand(C, seed=43) * 0.5 + 1.0).requires_grad_(True)
    beta = None if per_sample else (_rand(C, seed=44) * 0.3).requires_grad_(True)
    y = F.conv3d(x, w, padding=1)
    z = F.instance_norm(y, eps=1e-5) if per_sample else F.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5)
    a = F.leaky_relu(z, slope)
    da = _rand(*a.shape, seed=45)
    a.backward(da)

    xd, yd, dad = x.float().cuda(), y.detach().float().cuda(), da.float().cuda()
    G = N * C if per_sample else C
    mean, rstd = torch.empty(G, device="cuda"), torch.empty(G, device="cuda")
    ops.norm_stats(yd, per_sample, 1e-5, mean, rstd)
    gd = None if gamma is None else gamma.detach().float().cuda()
    bd = None if beta is None else beta.detach().float().cuda()
    sums = torch.full((G, 2), float("nan"), device="cuda")
    dg = None if gamma is None else torch.full((C,), float("nan"), device="cuda")
    dbt = None if beta is None else torch.full((C,), float("nan"), device="cuda")
    ops.norm_act_bwd_sums(yd, dad, per_sample, mean, rstd, gd, bd, slope, sums, dg, dbt)
    dw = torch.full((C, 1, 3, 3, 3), float("nan"), device="cuda")
    ops.conv_wgrad_cin1_norm(xd, dad, yd, per_sample, mean, rstd, gd, bd, sums, slope, dw)
    _close(dw, w.grad, rtol=2e-4, atol=1e-6)
    if gamma is not None:
        _close(dg, gamma.grad, rtol=1e-4)
        _close(dbt, beta.grad, rtol=1e-4)
    # the two-pass form
    dy = torch.empty_like(yd)
    ops.norm_act_bwd(yd, dad, dy, per_sample, mean, rstd, gd, bd, slope)
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(xd, dy, dw2, (3, 3, 3))
    _close(dw, dw2, rtol=1e-5, atol=1e-6)
    ops.conv_wgrad_cin1_norm(xd, dad, yd, per_sample, mean, rstd, gd, bd, sums, slope, dw2, accumulate=True)
    _close(dw2, 2 * w.grad, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("shape,per_sample,with_skip,p", [((2, 8, 4, 8, 16), False, True, 0.0), ((2, 6, 8, 4, 8), True, True, 0.0),
                                                          ((3, 16, 1, 16, 32), False, True, 0.3), ((1, 4, 6, 6, 12), False, False, 0.0),
                                                          ((2, 32, 1, 8, 8), False, False, 0.0)])
def test_norm_backward_with_the_pool_gradient_on_its_load_path(shape, per_sample, with_skip, p):
    """mis_norm_act_bwd_pool == mis_maxpool2_bwd (accumulating into the skip gradient) followed by mis_norm_act_bwd
    (reference unet_3D.py:35-47: conv_k feeds maxpool_k and the decoder's skip connection), 3-D and 2-D."""
    ops = _ops()
    N, C, D, H, W = shape
    x = _rand(*shape, seed=51, scale=2.0).float().cuda()
    G = N * C if per_sample else C
    mean, rstd = torch.empty(G, device="cuda"), torch.empty(G, device="cuda")
    ops.norm_stats(x, per_sample, 1e-5, mean, rstd)
    gamma = None if per_sample else (_rand(C, seed=52) * 0.5 + 1.0).float().cuda()
    beta = None if per_sample else (_rand(C, seed=53) * 0.3).float().cuda()
    mask = ((torch.rand(*shape, generator=torch.Generator().manual_seed(54)) >= p).float() / (1 - p)).cuda() if p > 0 else None
    a = torch.empty(shape, device="cuda")
    ops.norm_act_fwd(x, a, per_sample, mean, rstd, gamma, beta, 0.01, drop_p=p, drop_mask=mask)
    pshape = (N, C, D // 2 if D > 1 else 1, H // 2, W // 2)
    pooled = torch.empty(pshape, device="cuda")
    idx = torch.empty(pooled.numel(), dtype=torch.uint8, device="cuda")
    ops.maxpool2_fwd(a, pooled, idx)
    if W % 8 == 0:      # forward: the apply pass writes the pooled output and the argmax codes itself
        a2, pooled2 = torch.full(shape, float("nan"), device="cuda"), torch.full(pshape, float("nan"), device="cuda")
        idx2 = torch.full_like(idx, 255)
        ops.norm_act_fwd_pool(x, a2, pooled2, idx2, per_sample, mean, rstd, gamma, beta, 0.01, drop_p=p, drop_mask=mask)
        assert torch.equal(a2, a) and torch.equal(pooled2, pooled) and torch.equal(idx2, idx)
    dpool = _rand(*pshape, seed=55).float().cuda()
    dskip = _rand(*shape, seed=56).float().cuda() if with_skip else None
    # two-pass form
    da = dskip.clone() if with_skip else torch.empty(shape, device="cuda")
    ops.maxpool2_bwd(dpool, idx, da, accumulate=with_skip)
    dx_ref = torch.empty(shape, device="cuda")
    dg_ref = None if gamma is None else torch.empty(C, device="cuda")
    db_ref = None if beta is None else torch.empty(C, device="cuda")
    ops.norm_act_bwd(x, da, dx_ref, per_sample, mean, rstd, gamma, beta, 0.01, drop_p=p, drop_mask=mask, dgamma=dg_ref,
                     dbeta=db_ref)
    dx = torch.full(shape, float("nan"), device="cuda")
    dg = None if gamma is None else torch.full((C,), float("nan"), device="cuda")
    db = None if beta is None else torch.full((C,), float("nan"), device="cuda")
    ops.norm_act_bwd_pool(x, dskip, dpool, idx, dx, per_sample, mean, rstd, gamma, beta, 0.01, drop_p=p, drop_mask=mask,
                          dgamma=dg, dbeta=db)
    assert torch.equal(dx, dx_ref)
    if gamma is not None:
        assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)


@pytest.mark.parametrize("kind", ["unet3d", "vnet", "first:unet3d", "first:vnet", "pool:unet3d", "pool:unet2d"])
def test_step_with_fused_head_equals_unfused(kind):
    """A Mean-Teacher step (dropout on, device RNG) gives the same losses, gradients and weights with the classifier
    fused into the last block's pass and without.  One step: at this 32^3 fixture the 2^3 / 4^3 levels normalise over
    32 .. 256 values, and the 1e-8 weight differences after a first update already move their gradients by per cent."""
    from mis_hip import plan as plan_mod
    from mis_hip.step import MeanTeacherTrainer
    from networks.net_factory_3d import net_factory_3d
    from oracle import filler
    attr = {"first": "FUSE_FIRST", "pool": "FUSE_POOL"}.get(kind.split(":")[0], "FUSE_HEAD") if ":" in kind else "FUSE_HEAD"
    kind = kind.split(":")[-1]
    if kind == "unet2d":
        from networks.net_factory import net_factory
        make = lambda: net_factory("unet", 1, 2)
        vshape, lshape, ldt = (4, 1, 64, 64), (4, 64, 64), torch.uint8
    else:
        key = {"unet3d": "unet_3D", "vnet": "vnet"}[kind]
        make = lambda: net_factory_3d(key, 1, 2)
        vshape, lshape, ldt = (4, 1, 32, 32, 32), (4, 32, 32, 32), torch.int64
    torch.manual_seed(5)
    sd0 = {k: v.clone() for k, v in make().state_dict().items()}
    vol = filler.image(vshape, "volume").cuda()
    lab = filler.labels(lshape, 2, ldt).cuda()
    res = []
    keep = getattr(plan_mod, attr)
    for fuse in (True, False):
        setattr(plan_mod, attr, fuse)
        try:
            m, e = make(), make()
            m.load_state_dict(sd0); e.load_state_dict(sd0)
            tr = MeanTeacherTrainer(m, e, labeled_bs=2, num_classes=2, cons_start_iter=0, seed=11, iter_num=1500)
            tr.step(vol, lab)
            torch.cuda.synchronize()
            mark = {"FUSE_HEAD": "head", "FUSE_FIRST": "norm_bwd", "FUSE_POOL": "pool"}[attr]
            fused = [type(op).__name__ for p in m._plans.values() for op in p.ops if getattr(op, mark, None) is not None]
            assert bool(fused) == fuse
            res.append((tr.losses(), m.flat_grad.clone(), m.flat_param.clone(), e.flat_param.clone()))
        finally:
            setattr(plan_mod, attr, keep)
    (l0, g0, p0, t0), (l1, g1, p1, t1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 2e-6, (k, l0[k], l1[k])
    gs = float(g1.abs().max())
    assert (g0 - g1).abs().max().item() <= 1e-4 * gs
    assert (p0 - p1).abs().max().item() <= 1e-6 and (t0 - t1).abs().max().item() <= 1e-6

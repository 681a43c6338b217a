"""V-Net specific entry points (space/depth re-layout, add, input-major 1x1 packs, channel dropout) and the
plan ops built from them, against stock torch CPU fp32 ops (reference layers: vnet.py:73 Conv3d(k=2,s=2),
vnet.py:100 ConvTranspose3d(k=2,s=2), vnet.py:177 Dropout3d, vnet.py:210-222 additive skips)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _close(a, b, rtol=2e-4, atol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref scale {ref:.3e}"


def _pref(t):
    from mis_hip.plan import _PRef
    return _PRef(t.cuda().contiguous(), torch.zeros_like(t).cuda())


def _act(t):
    from mis_hip.plan import Act
    return Act(tensor=t.cuda().contiguous())


def _ctx():
    from mis_hip.plan import Ctx
    return Ctx(True)


@pytest.mark.parametrize("N,C,D,H,W", [(2, 3, 4, 6, 8), (2, 3, 4, 6, 12), (1, 5, 6, 4, 24)])     # W % 8 == 0: float4 form
def test_space_to_depth_round_trip_and_layout(N, C, D, H, W):
    from mis_hip import ops
    x = _rand(N, C, D, H, W, seed=1)
    xs = torch.empty(N, 8 * C, D // 2, H // 2, W // 2, device="cuda")
    ops.space_to_depth2(x.cuda(), xs, (N, C, D, H, W), True)
    ref = x.view(N, C, D // 2, 2, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(xs.shape)
    assert torch.equal(xs.cpu(), ref)                       # coarse channel = c*8 + kz*4 + ky*2 + kx
    back = torch.full((N, C, D, H, W), 7.0, device="cuda")
    ops.space_to_depth2(xs, back, (N, C, D, H, W), False)
    assert torch.equal(back.cpu(), x)
    bias = _rand(C, seed=2).cuda()
    ops.space_to_depth2(xs, back, (N, C, D, H, W), False, bias=bias, accumulate=True)
    _close(back, 2 * x + bias.cpu().view(1, C, 1, 1, 1), rtol=1e-6, atol=1e-6)
    ops.space_to_depth2(x.cuda(), xs, (N, C, D, H, W), True, accumulate=True)
    assert torch.equal(xs.cpu(), 2 * ref)


def test_add_copy_and_strided():
    from mis_hip import ops
    a, b = _rand(2, 5, 4, 4, 8, seed=3).cuda(), _rand(2, 8, 4, 4, 8, seed=4).cuda()
    out = torch.empty(2, 5, 4, 4, 8, device="cuda")
    ops.add(a, b[:, 2:7], out)
    assert torch.equal(out, a + b[:, 2:7])
    ops.add(a, None, out)
    assert torch.equal(out, a)
    ops.add(out, a, out)                                    # in place accumulate
    assert torch.equal(out, a + a)


@pytest.mark.parametrize("N,Cin,Cout,dims", [(2, 16, 32, (4, 16, 16)), (1, 32, 64, (3, 18, 18)), (2, 16, 32, (2, 20, 14))])
def test_k2s2_in_place_kernels(N, Cin, Cout, dims):
    """conv_k2s2.hip through the C-ABI: Conv3d(k2s2) forward on the fine volume and its data gradient (= the transposed
    convolution) against torch, channel-slice views, ragged 16-voxel tiles (18 x 18, 20 x 14 planes), accumulate."""
    from mis_hip import ops
    Do, Ho, Wo = dims
    assert ops.conv_k2s2_eligible(Cin, Cout, dims, False) and ops.conv_k2s2_eligible(Cout, Cin, dims, True)
    wide = _rand(N, Cin + 4, 2 * Do, 2 * Ho, 2 * Wo, seed=21)
    x = wide[:, 4:].clone().requires_grad_(True)
    w = _rand(Cout, Cin, 2, 2, 2, seed=22, scale=0.2)
    b = _rand(Cout, seed=23)
    y_ref = F.conv3d(x, w, b, stride=2)
    dy = _rand(*y_ref.shape, seed=24)
    y_ref.backward(dy)
    wd_, xd = wide.cuda(), None
    xd = wd_[:, 4:]
    y = torch.full(tuple(y_ref.shape), float("nan"), device="cuda")
    ops.conv_k2s2_down(xd, w.cuda().contiguous(), b.cuda(), y)
    _close(y, y_ref)
    y2 = y.clone()
    ops.conv_k2s2_down(xd, w.cuda().contiguous(), None, y2, accumulate=True)
    _close(y2, 2 * y_ref - b.view(1, -1, 1, 1, 1))
    dwide = torch.full(tuple(wide.shape), float("nan"), device="cuda")
    dx = dwide[:, 4:]
    ops.conv_k2s2_up(dy.cuda(), w.cuda().contiguous(), None, dx)
    _close(dx, x.grad)
    assert torch.isnan(dwide[:, :4]).all()
    ops.conv_k2s2_up(dy.cuda(), w.cuda().contiguous(), None, dx, accumulate=True)
    _close(dx, 2 * x.grad)
    # the same pair as ConvTranspose3d forward (+ bias) and its data gradient
    wt = _rand(Cout, Cin, 2, 2, 2, seed=25, scale=0.2)          # ConvTranspose3d(Cout -> Cin) parameter [in][out][2][2][2]
    bt = _rand(Cin, seed=26)
    xc = _rand(N, Cout, Do, Ho, Wo, seed=27).requires_grad_(True)
    yt_ref = F.conv_transpose3d(xc, wt, bt, stride=2)
    g = _rand(*yt_ref.shape, seed=28)
    yt_ref.backward(g)
    yt = torch.full(tuple(yt_ref.shape), float("nan"), device="cuda")
    ops.conv_k2s2_up(xc.detach().cuda(), wt.cuda().contiguous(), bt.cuda(), yt)
    _close(yt, yt_ref)
    dxc = torch.full(tuple(xc.shape), float("nan"), device="cuda")
    ops.conv_k2s2_down(g.cuda(), wt.cuda().contiguous(), None, dxc)
    _close(dxc, xc.grad)


@pytest.mark.parametrize("N,Cin,Cout,d", [(8, 16, 32, 48), (8, 32, 64, 24)])
def test_k2s2_full_size_adjoint_identities(N, Cin, Cout, d):
    """Size-independent properties at the BASELINE geometry (config 3's V-Net: 4+4 volumes of 96^3, the 96^3 <-> 48^3 and
    48^3 <-> 24^3 levels), where a CPU reference would take minutes: the three in-place kernels are one bilinear form
    B(w, x, g) = <conv_k2s2(x; w), g>, so  <down(x), g> = <x, up(g)> = <wgrad(g, x), w>  (fp64 dot products of fp32 results)."""
    from mis_hip import ops
    gen = torch.Generator(device="cuda").manual_seed(41)
    x = torch.rand(N, Cin, 2 * d, 2 * d, 2 * d, generator=gen, device="cuda") - 0.5
    g = torch.rand(N, Cout, d, d, d, generator=gen, device="cuda") - 0.5
    w = (torch.rand(Cout, Cin * 8, generator=gen, device="cuda") - 0.5) * 0.2
    y = torch.empty_like(g)
    ops.conv_k2s2_down(x, w, None, y)
    dx = torch.empty_like(x)
    ops.conv_k2s2_up(g, w, None, dx)
    dw = torch.empty(Cout * Cin * 8, device="cuda")
    ops.conv_k2s2_wgrad(g, x, dw)
    a = torch.dot(y.double().view(-1), g.double().view(-1)).item()
    b = torch.dot(x.double().view(-1), dx.double().view(-1)).item()
    c = torch.dot(dw.double(), w.double().view(-1)).item()
    scale = (y.double().norm() * g.double().norm()).item()
    assert abs(a - b) <= 1e-6 * scale and abs(a - c) <= 1e-6 * scale, (a, b, c, scale)
    # linearity in x
    y2 = torch.empty_like(y)
    ops.conv_k2s2_down(2.5 * x, w, None, y2)
    assert (y2 - 2.5 * y).abs().max().item() <= 1e-5 * y.abs().max().item()


@pytest.mark.parametrize("N,CF,CC,dims", [(2, 16, 32, (4, 16, 16)), (1, 32, 64, (3, 18, 24)), (3, 16, 32, (2, 20, 16))])
def test_k2s2_weight_gradient_in_place(N, CF, CC, dims):
    """mis_conv_k2s2_wgrad against torch for both parameter layouts (Conv3d: coarse = dy; ConvTranspose3d: coarse = x),
    ragged 16-voxel tiles, channel-slice views, accumulate, run-to-run identical."""
    from mis_hip import ops
    Do, Ho, Wo = dims
    assert ops.conv_k2s2_wgrad_eligible(CF, CC, dims)
    x = _rand(N, CF, 2 * Do, 2 * Ho, 2 * Wo, seed=31)
    w = _rand(CC, CF, 2, 2, 2, seed=32, scale=0.2).requires_grad_(True)
    y = F.conv3d(x, w, None, stride=2)
    dy = _rand(*y.shape, seed=33)
    y.backward(dy)
    wide = torch.zeros(N, CF + 4, 2 * Do, 2 * Ho, 2 * Wo)
    wide[:, 4:] = x
    xd = wide.cuda()[:, 4:]
    dw = torch.full((CC, CF * 8), float("nan"), device="cuda")
    ops.conv_k2s2_wgrad(dy.cuda(), xd, dw.view(-1))
    _close(dw.view_as(w), w.grad, rtol=3e-4, atol=1e-4)
    dw2 = dw.clone()
    ops.conv_k2s2_wgrad(dy.cuda(), xd, dw2.view(-1), accumulate=True)
    _close(dw2.view_as(w), 2 * w.grad, rtol=3e-4, atol=1e-4)
    dw3 = torch.empty_like(dw)
    ops.conv_k2s2_wgrad(dy.cuda(), xd, dw3.view(-1))
    assert torch.equal(dw, dw3)
    # ConvTranspose3d(CC -> CF): coarse = its input, fine = the gradient of its output, dw in its [CC][CF][2][2][2] layout
    wt = _rand(CC, CF, 2, 2, 2, seed=34, scale=0.2).requires_grad_(True)
    xc = _rand(N, CC, Do, Ho, Wo, seed=35)
    yt = F.conv_transpose3d(xc, wt, None, stride=2)
    g = _rand(*yt.shape, seed=36)
    yt.backward(g)
    dwt = torch.full((CC, CF * 8), float("nan"), device="cuda")
    ops.conv_k2s2_wgrad(xc.cuda(), g.cuda(), dwt.view(-1))
    _close(dwt.view_as(wt), wt.grad, rtol=3e-4, atol=1e-4)


@pytest.mark.parametrize("N,Cin,Cout,d", [(2, 16, 32, 8), (1, 32, 64, 4), (2, 128, 256, 2), (1, 20, 24, 6), (2, 16, 32, 16),
                                          (1, 32, 64, 18), (1, 32, 64, 24)])
def test_down_conv_op(N, Cin, Cout, d):
    from mis_hip.plan import DownConvOp
    x = _rand(N, Cin, 2 * d, 2 * d, 2 * d, seed=5).requires_grad_(True)
    w = _rand(Cout, Cin, 2, 2, 2, seed=6, scale=0.2).requires_grad_(True)
    b = _rand(Cout, seed=7)
    y_ref = F.conv3d(x, w, b, stride=2)
    dy = _rand(*y_ref.shape, seed=8)
    y_ref.backward(dy)
    xa, ya = _act(x.detach()), _act(torch.empty_like(y_ref))
    wp, bp = _pref(w.detach()), _pref(b)
    op = DownConvOp(xa, ya, wp, bp)
    assert op.direct == op.direct_dx == (d >= 16)          # the in-place kernels serve V-Net's two largest levels
    assert op.direct_wg == (d >= 16 and d % 8 == 0)
    op.fwd(_ctx())
    _close(ya.t, y_ref)
    ya.g = dy.cuda()
    op.bwd(_ctx())
    _close(xa.grad(), x.grad)
    _close(wp.grad, w.grad, rtol=3e-4, atol=1e-4)
    # second consumer already wrote the input gradient: accumulate
    op.bwd(_ctx())
    _close(xa.grad(), 2 * x.grad)


@pytest.mark.parametrize("N,Cin,Cout,d", [(2, 32, 16, 8), (1, 64, 32, 4), (2, 256, 128, 2), (1, 24, 20, 6), (2, 32, 16, 16),
                                          (1, 64, 32, 18), (1, 64, 32, 24)])
def test_up_conv_op(N, Cin, Cout, d):
    from mis_hip.plan import UpConvOp
    x = _rand(N, Cin, d, d, d, seed=9).requires_grad_(True)
    w = _rand(Cin, Cout, 2, 2, 2, seed=10, scale=0.2).requires_grad_(True)
    b = _rand(Cout, seed=11)
    y_ref = F.conv_transpose3d(x, w, b, stride=2)
    dy = _rand(*y_ref.shape, seed=12)
    y_ref.backward(dy)
    xa, ya = _act(x.detach()), _act(torch.empty_like(y_ref))
    wp, bp = _pref(w.detach()), _pref(b)
    op = UpConvOp(xa, ya, wp, bp)
    assert op.direct == op.direct_dx == (d >= 16)
    assert op.direct_wg == (d >= 16 and d % 8 == 0)
    op.fwd(_ctx())
    _close(ya.t, y_ref)
    ya.g = dy.cuda()
    op.bwd(_ctx())
    _close(xa.grad(), x.grad)
    _close(wp.grad, w.grad, rtol=3e-4, atol=1e-4)


def test_add_op_backward_accumulates_into_shared_skip():
    from mis_hip.plan import AddOp
    a, b = _act(_rand(1, 4, 4, 4, 4, seed=13)), _act(_rand(1, 4, 4, 4, 4, seed=14))
    out = _act(torch.empty(1, 4, 4, 4, 4))
    op = AddOp(a, b, out)
    op.fwd(_ctx())
    assert torch.equal(out.t, a.t + b.t)
    out.g = _rand(1, 4, 4, 4, 4, seed=15).cuda()
    b.grad().fill_(1.0)
    b.mark_written()                                        # e.g. the down conv consumed the skip first
    op.bwd(_ctx())
    assert torch.equal(a.grad(), out.g)
    assert torch.equal(b.grad(), out.g + 1.0)


def test_channel_dropout_drops_whole_feature_maps_and_backward_regenerates():
    """nn.Dropout3d semantics on the device RNG: per (n, c) Bernoulli(0.5), kept maps scaled by 2."""
    from mis_hip import ops
    from mis_hip.plan import NormActOp, Ctx
    N, C, S = 4, 64, (4, 4, 8)
    x = _rand(N, C, *S, seed=16) + 3.0                      # positive after BN+ReLU for most voxels
    xa, ya = _act(x), _act(torch.empty_like(x))
    gamma, beta = _pref(torch.ones(C)), _pref(torch.full((C,), 2.0))
    op = NormActOp(xa, ya, False, gamma, beta, None, 0.0, 0.5, site=3)
    op.drop3d = True
    state = ops.new_step_state()
    ops.step_init(state, 1234, 0, 0.01, 30000, 0.99, 0.1, 200.0)
    ctx = Ctx(True, state=state, rng_stream=1)
    op.fwd(ctx)
    off = NormActOp(xa, _act(torch.empty_like(x)), False, gamma, beta, None, 0.0, 0.0, site=3)
    off.fwd(ctx)
    y, y0 = ya.t.cpu(), off.y.t.cpu()
    kept = 0
    for n in range(N):
        for c in range(C):
            if y[n, c].abs().max() == 0:
                continue
            kept += 1
            _close(y[n, c], 2.0 * y0[n, c], rtol=1e-6, atol=1e-6)
    assert 0.3 * N * C < kept < 0.7 * N * C
    ya.g = torch.ones_like(x).cuda()
    op.bwd(ctx)
    dx = xa.grad().cpu()
    dropped = (y.abs().amax(dim=(2, 3, 4)) == 0)
    # a dropped map passes no gradient to beta; dbeta counts 2 per kept positive voxel
    pos = (y0 > 0).double().sum(dim=(2, 3, 4))
    expect_dbeta = (2.0 * pos * (~dropped)).sum(0)
    _close(beta.grad, expect_dbeta, rtol=1e-5, atol=1e-3)
    assert torch.isfinite(dx).all()
    # same seed/offset -> same mask; different stream -> different mask
    op2 = NormActOp(xa, _act(torch.empty_like(x)), False, gamma, beta, None, 0.0, 0.5, site=3)
    op2.drop3d = True
    op2.fwd(ctx)
    assert torch.equal(op2.y.t, ya.t)
    op2.fwd(Ctx(True, state=state, rng_stream=2))
    assert not torch.equal(op2.y.t, ya.t)

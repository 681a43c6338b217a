"""nn.ConvTranspose2d(C1, C2, kernel_size=2, stride=2) of UpBlock(bilinear=False) (reference code/networks/unet.py:76-78) as
mis_hip.plan.UpConv2dOp (1x1 MFMA convolution + 2-D pixel shuffle): forward and all three gradients against torch float64,
writing into a channel slice of a wider buffer like the decoder's concat."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(3, 32, 16, 8, 12), (2, 256, 128, 4, 4), (5, 64, 32, 16, 16), (2, 16, 8, 6, 10)])
def test_upconv2d_matches_conv_transpose2d(N, Cin, Cout, H, W):
    from mis_hip.plan import Act, UpConv2dOp, _PRef
    g = torch.Generator().manual_seed(3)
    x = torch.rand((N, Cin, H, W), generator=g) * 2 - 1
    w = (torch.rand((Cin, Cout, 2, 2), generator=g) * 2 - 1) * Cin ** -0.5
    b = torch.rand((Cout,), generator=g) - 0.5
    dy = torch.rand((N, Cout, 2 * H, 2 * W), generator=g) * 2 - 1
    xa = Act(tensor=x.cuda().view(N, Cin, 1, H, W).contiguous())
    cat = Act((N, Cout + 8, 1, 2 * H, 2 * W))                  # the op writes channels 8.. of a wider buffer
    cat.t.fill_(float("nan"))
    ya = cat.slice(8, Cout)
    wref = _PRef(w.cuda(), torch.full_like(w, float("nan"), device="cuda"))
    bref = _PRef(b.cuda(), torch.full_like(b, float("nan"), device="cuda"))
    op = UpConv2dOp(xa, ya, wref, bref, bias_grad=True)
    op.fwd(None)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv_transpose2d(xd, wd, bd, stride=2)
    got = ya.t.cpu().double().view_as(ref)
    assert (got - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    assert torch.isnan(cat.t[:, :8]).all()                      # nothing outside the slice
    cat.grad().fill_(float("nan"))
    ya.grad().copy_(dy.cuda().view(N, Cout, 1, 2 * H, 2 * W))
    ref.backward(dy.double())
    op.bwd(None)
    for name, mine, want in (("dx", xa.grad().cpu().double().view_as(xd), xd.grad), ("dw", wref.grad.cpu().double(), wd.grad),
                             ("db", bref.grad.cpu().double(), bd.grad)):
        err = (mine - want).abs().max().item()
        assert err <= 5e-6 * max(1.0, want.abs().max().item()), (name, err, want.abs().max().item())

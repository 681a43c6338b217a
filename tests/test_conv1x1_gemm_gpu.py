"""mis_conv1x1_gemm (csrc/conv1x1_gemm.hip): the 1x1x1 convolution of small channel-major volumes as a batched, split-K GEMM.

What it replaces: the contraction of nn.Conv3d(C, 2C, 2, stride=2) / nn.ConvTranspose3d(2C, C, 2, stride=2) on V-Net's 12^3 and
6^3 levels (reference code/networks/vnet.py:73, :100) after space-to-depth.  Checked against torch fp64 on the CPU and against
the direct kernel (mis_conv_fwd, k = 1), tolerance 2e-4 of the largest reference value as for the other fp32 MFMA kernels."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _close(a, b, rtol=2e-4, atol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err, ref = (a - b).abs().max().item(), b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref scale {ref:.3e}"


# N, Cin, Cout, (D, H, W)
CASES = [
    (8, 512, 128, (12, 12, 12)),      # V-Net block_three_dw (64 -> 128, 24^3 -> 12^3) after space-to-depth: split over the channels
    (8, 1024, 256, (6, 6, 6)),        # block_four_dw: 32 (image, tile) entries, 8 slices
    (8, 128, 512, (12, 12, 12)),      # block_six_up (128 -> 64, 12^3 -> 24^3): many tiles, short contraction, no split
    (4, 256, 1024, (6, 6, 6)),        # block_five_up on the teacher's half batch
    (3, 64, 20, (5, 6, 4)),           # ragged tiles in both dimensions, one k-step of 64
    (1, 96, 132, (2, 2, 9)),          # 36 voxels, K not a multiple of the k-step
    (2, 200, 64, (8, 8, 8)),
]


@pytest.mark.parametrize("N,Cin,Cout,sp", CASES)
def test_conv1x1_gemm_matches_reference(N, Cin, Cout, sp):
    from mis_hip import ops
    x = _rand(N, Cin, *sp, seed=1)
    w = _rand(Cout, Cin, seed=2, scale=Cin ** -0.5)
    b = _rand(Cout, seed=3)
    ref = F.conv3d(x, w.view(Cout, Cin, 1, 1, 1), b)
    xd, wt, bd = x.float().cuda(), w.t().contiguous().float().cuda(), b.float().cuda()
    y = torch.full((N, Cout) + sp, float("nan"), device="cuda")
    assert ops.conv1x1_gemm_eligible(xd, y, Cin, Cout)
    ops.conv1x1_gemm(xd, wt, bd, y)
    _close(y, ref)
    # deterministic; `accumulate` adds; no bias
    y2 = torch.empty_like(y)
    ops.conv1x1_gemm(xd, wt, bd, y2)
    assert torch.equal(y, y2)
    ops.conv1x1_gemm(xd, wt, None, y2, accumulate=True)
    _close(y2, 2 * ref - b.view(1, -1, 1, 1, 1))
    # the direct kernel agrees
    yd = torch.empty_like(y)
    ops.conv_fwd(xd, ops.conv_pack_raw(w.float().cuda().contiguous(), Cout, Cin, 1, 0), bd, yd, Cin, Cout, (1, 1, 1))
    _close(y, yd)


def test_conv1x1_gemm_on_channel_slices_and_weight_row_stride():
    """x / y = channel slices of wider buffers (batch strides != C * S); the weights a column slice of a wider matrix."""
    from mis_hip import ops
    N, C0, Cin, Cout, sp = 4, 32, 256, 128, (6, 6, 6)
    cat = _rand(N, C0 + Cin, *sp, seed=5).float().cuda()
    wide = _rand(Cin, Cout + 64, seed=6, scale=0.05).float().cuda()
    wt = wide[:, 64:]                                   # [Cin][Cout], row stride Cout + 64
    out = torch.zeros(N, 16 + Cout, *sp, device="cuda")
    assert ops.conv1x1_gemm_eligible(cat[:, C0:], out[:, 16:], Cin, Cout)
    ops.conv1x1_gemm(cat[:, C0:], wt, None, out[:, 16:])
    ref = F.conv3d(cat[:, C0:].cpu().double(), wt.t().cpu().double().reshape(Cout, Cin, 1, 1, 1))
    _close(out[:, 16:], ref)
    assert out[:, :16].abs().max().item() == 0.0


def test_conv1x1_gemm_refusals():
    from mis_hip import ops
    x = torch.zeros(2, 64, 3, 3, 3, device="cuda")          # 27 voxels: rows are not whole float4s
    y = torch.zeros(2, 64, 3, 3, 3, device="cuda")
    assert not ops.conv1x1_gemm_eligible(x, y, 64, 64)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.conv1x1_gemm(x, torch.zeros(64, 64, device="cuda"), None, y)
    big = torch.zeros(1, 64, 32, 32, 32, device="cuda")     # many voxels: the spatially tiled kernel's job
    assert not ops.conv1x1_gemm_eligible(big, big, 64, 64)


# N, M (channels of a), Nc (channels of b), (D, H, W)
WGRAD_CASES = [
    (8, 128, 512, (12, 12, 12)),      # V-Net block_three_dw: dw[Cout][8 Cin]
    (8, 256, 1024, (6, 6, 6)),        # block_four_dw
    (8, 256, 1024, (6, 6, 6)),
    (8, 128, 512, (12, 12, 12)),
    (4, 1024, 256, (6, 6, 6)),
    (3, 72, 100, (5, 6, 4)),          # ragged tiles, one chunk of 120 voxels
    (2, 64, 200, (8, 8, 8)),
]


@pytest.mark.parametrize("N,M,Nc,sp", WGRAD_CASES)
def test_conv1x1_wgrad_matches_reference(N, M, Nc, sp):
    from mis_hip import ops
    a = _rand(N, M, *sp, seed=11)
    b = _rand(N, Nc, *sp, seed=12)
    ref = torch.einsum("nmdhw,nkdhw->mk", a, b)
    ad, bd = a.float().cuda(), b.float().cuda()
    assert ops.conv1x1_wgrad_eligible(ad, bd)
    dw = torch.full((M, Nc), float("nan"), device="cuda")
    ops.conv1x1_wgrad(ad, bd, dw)
    _close(dw, ref, rtol=1e-5, atol=1e-6)
    dw2 = torch.empty_like(dw)
    ops.conv1x1_wgrad(ad, bd, dw2)
    assert torch.equal(dw, dw2)                       # fixed slices, fixed-order reduction
    ops.conv1x1_wgrad(ad, bd, dw2, accumulate=True)
    _close(dw2, 2 * ref, rtol=1e-5, atol=1e-6)
    # the spatially tiled kernel agrees (a = dy, b = x: dw[Cout][Cin])
    dwd = torch.empty(M, Nc, 1, 1, 1, device="cuda")
    ops.conv_wgrad(bd, ad, dwd, (1, 1, 1))
    _close(dw, dwd.view(M, Nc), rtol=1e-5, atol=1e-6)


def test_conv1x1_wgrad_on_channel_slices_into_a_wider_gradient():
    from mis_hip import ops
    N, sp = 4, (6, 6, 6)
    abuf = _rand(N, 32 + 128, *sp, seed=21).float().cuda()
    bbuf = _rand(N, 256 + 16, *sp, seed=22).float().cuda()
    wide = torch.zeros(128, 256 + 64, device="cuda")
    ops.conv1x1_wgrad(abuf[:, 32:], bbuf[:, :256], wide[:, 64:])
    ref = torch.einsum("nmdhw,nkdhw->mk", abuf[:, 32:].cpu().double(), bbuf[:, :256].cpu().double())
    _close(wide[:, 64:], ref, rtol=1e-5, atol=1e-6)
    assert wide[:, :64].abs().max().item() == 0.0

"""SwinUNETR encoder kernels (csrc/swin3d.hip) through the C-ABI against the torch restatement of MONAI's published
algorithm (oracle/swinunetr.py) -- PARITY UNPINNED: MONAI is an un-vendored dependency of the reference
(code/networks/net_factory_3d.py:7,37-38), absent from the image; what is pinned is the call site.

window gather / scatter = F.pad + torch.roll + window_partition (and window_reverse + roll back + un-pad); merge3d =
the v0.9 "merging" slice order; attention = WindowAttention.forward with the [:n, :n] block of the 7^3 relative-position
index and the 0 / -100 shift mask, forward and backward against torch fp64 autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _tops():
    from mis_hip import tops
    return tops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


@pytest.mark.parametrize("B,dims,C,shifted", [(2, (8, 8, 8), 8, False), (2, (8, 8, 8), 8, True), (1, (16, 12, 9), 12, True),
                                              (2, (4, 4, 4), 16, True), (1, (32, 32, 32), 48, True)])
def test_window_gather_and_scatter(B, dims, C, shifted):
    from oracle.swinunetr import window_geometry, window_partition, window_reverse
    tops = _tops()
    win, shift = window_geometry(dims)
    if not shifted:
        shift = (0, 0, 0)
    D, H, W = dims
    x = _rand(B, D, H, W, C, seed=1)
    pd, ph, pw = ((win[i] - s % win[i]) % win[i] for i, s in enumerate(dims))
    xp = F.pad(x, (0, 0, 0, pw, 0, ph, 0, pd))
    if any(shift):
        xp = torch.roll(xp, shifts=(-shift[0], -shift[1], -shift[2]), dims=(1, 2, 3))
    ref = window_partition(xp, win)
    nWB = tops.win3d_windows(B, dims, win)
    n = win[0] * win[1] * win[2]
    assert ref.shape == (nWB, n, C)
    xd = x.float().cuda().reshape(-1, C)
    wd = torch.full((nWB * n, C), float("nan"), device="cuda")
    tops.win3d_gather(xd, wd, B, dims, C, win, shift)
    assert torch.equal(wd.cpu().double().view(nWB, n, C), ref.float().double())
    # inverse: window_reverse + roll back + un-pad
    y = _rand(nWB, n, C, seed=2)
    r = window_reverse(y, win, (B, D + pd, H + ph, W + pw))
    if any(shift):
        r = torch.roll(r, shifts=shift, dims=(1, 2, 3))
    r = r[:, :D, :H, :W, :]
    out = torch.full((B * D * H * W, C), float("nan"), device="cuda")
    tops.win3d_gather(y.float().cuda().reshape(-1, C), out, B, dims, C, win, shift, inverse=True)
    assert torch.equal(out.cpu().double().view(B, D, H, W, C), r.float().double())


@pytest.mark.parametrize("B,dims,C", [(2, (4, 6, 8), 8), (1, (16, 16, 16), 96)])
def test_patch_merging_gather_and_its_gradient(B, dims, C):
    from oracle.swinunetr import MERGE_OFFSETS
    tops = _tops()
    D, H, W = dims
    x = _rand(B, D, H, W, C, seed=3).requires_grad_(True)
    ref = torch.cat([x[:, o[0]::2, o[1]::2, o[2]::2, :] for o in MERGE_OFFSETS], -1)
    g = _rand(*ref.shape, seed=4)
    ref.backward(g)
    xd = x.detach().float().cuda().reshape(-1, C)
    m = torch.full((B * (D // 2) * (H // 2) * (W // 2), 8 * C), float("nan"), device="cuda")
    tops.merge3d(xd, m, B, dims, C)
    assert torch.equal(m.cpu().double(), ref.detach().float().double().reshape(-1, 8 * C))
    dx = torch.full((B * D * H * W, C), float("nan"), device="cuda")
    tops.merge3d(g.float().cuda().reshape(-1, 8 * C), dx, B, dims, C, inverse=True)
    err = (dx.cpu().double().view(B, D, H, W, C) - x.grad).abs().max().item()
    assert err <= 1e-6, err       # two fp32 adds at most


def _attn_ref(qkv, table, nH, n, mask):
    """WindowAttention.forward after the qkv Linear, before proj (oracle/swinunetr.py::_attention), in the dtype given."""
    from oracle.swinunetr import relative_position_index
    b = qkv.shape[0]
    c = qkv.shape[2] // 3
    q3 = qkv.reshape(b, n, 3, nH, c // nH).permute(2, 0, 3, 1, 4)
    q, k, v = q3[0] * (c // nH) ** -0.5, q3[1], q3[2]
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index()[:n, :n].reshape(-1)
    attn = attn + table[idx].reshape(n, n, -1).permute(2, 0, 1).unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(b // nw, nw, nH, n, n) + mask.to(attn.dtype).unsqueeze(1).unsqueeze(0)).view(-1, nH, n, n)
    attn = torch.softmax(attn, dim=-1)
    return (attn @ v).transpose(1, 2).reshape(b, n, c)


@pytest.mark.parametrize("B,dims,nH,shifted", [(1, (8, 8, 8), 3, False), (2, (8, 8, 8), 3, True), (2, (4, 4, 4), 6, False),
                                               (1, (16, 16, 16), 6, True), (1, (9, 8, 10), 3, True)])
def test_window_attention_forward_backward(B, dims, nH, shifted):
    from oracle.swinunetr import region_ids, window_geometry
    tops = _tops()
    win, shift = window_geometry(dims)
    if not shifted or not any(shift):
        shift, shifted = (0, 0, 0), False
    n = win[0] * win[1] * win[2]
    pdims = tuple(-(-s // win[k]) * win[k] for k, s in enumerate(dims))
    nW = (pdims[0] // win[0]) * (pdims[1] // win[1]) * (pdims[2] // win[2])
    BW, C = B * nW, nH * 16
    qkv = _rand(BW, n, 3 * C, seed=5, scale=1.5).requires_grad_(True)
    table = _rand(13 ** 3, nH, seed=6, scale=0.5).requires_grad_(True)
    mask, region = None, None
    if shifted:
        r = region_ids(pdims, win, shift)
        mask = r.unsqueeze(1) - r.unsqueeze(2)
        mask = mask.masked_fill(mask != 0, -100.0).masked_fill(mask == 0, 0.0).double()
        region = r.to(torch.int32).contiguous().cuda()
    ref = _attn_ref(qkv, table, nH, n, mask)
    g = _rand(BW, n, C, seed=7)
    ref.backward(g)

    qd = qkv.detach().float().cuda().reshape(BW * n, 3 * C)
    td = table.detach().float().cuda()
    out = torch.full((BW * n, C), float("nan"), device="cuda")
    stats = torch.empty(BW * nH * n * 2, device="cuda")
    tops.win3d_attn_fwd(qd, out, stats, td, region, BW, nW, n, nH)
    err = (out.cpu().double().view(BW, n, C) - ref.detach()).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.detach().abs().max().item()), err
    dq = torch.full((BW * n, 3 * C), float("nan"), device="cuda")
    dt = torch.full((13 ** 3, nH), float("nan"), device="cuda")
    tops.win3d_attn_bwd(qd, out, g.float().cuda().reshape(BW * n, C), dq, stats, td, region, dt, BW, nW, n, nH)
    e1 = (dq.cpu().double().view(BW, n, 3 * C) - qkv.grad).abs().max().item()
    assert e1 <= 5e-5 * max(1.0, qkv.grad.abs().max().item()), e1
    e2 = (dt.cpu().double() - table.grad).abs().max().item()
    assert e2 <= 5e-5 * max(1.0, table.grad.abs().max().item()), e2
    # deterministic, and accumulate adds
    dq2 = torch.empty_like(dq)
    dt2 = dt.clone()
    tops.win3d_attn_bwd(qd, out, g.float().cuda().reshape(BW * n, C), dq2, stats, td, region, dt2, BW, nW, n, nH,
                        accumulate_table=True)
    assert torch.equal(dq, dq2)
    assert torch.allclose(dt2, 2 * dt, rtol=1e-6, atol=1e-7)

"""Winograd F(2x2x2, 3x3x3) kernels (csrc/conv_wino.hip, csrc/conv_wino_wgrad.hip) through the C-ABI, against torch
fp64 on the CPU and against the direct MFMA kernels.

What they replace: nn.Conv3d(k=3, pad=1) forward, its autograd data gradient and weight gradient in UnetConv3 /
UnetUp3_CT / ConvBlock (reference code/networks/utils.py:99-123, unet_3D.py:28-57, vnet.py:15-22).  The arithmetic is
fp32 end to end; the tolerance is the one the direct kernels are held to in test_kernels_gpu.py (2e-4 relative to the
largest reference value, fp32 with another summation order)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    from mis_hip import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _close(a, b, rtol=2e-4, atol=2e-5):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref scale {ref:.3e}"


# N, Cin, Cout, D, H, W, expected forward variant
FWD_CASES = [
    (1, 16, 16, 4, 4, 32, 0),
    (2, 16, 32, 8, 12, 64, 0),
    (1, 48, 16, 4, 8, 96, 0),       # the decoder's concat conv (3 chunks of 16 input channels)
    (2, 24, 48, 4, 8, 32, 0),       # 6 chunks, 3 output blocks
    (2, 16, 32, 4, 8, 16, 1),
    (1, 32, 32, 8, 8, 48, 1),
    (3, 96, 32, 4, 8, 16, 1),
    (1, 32, 32, 8, 8, 8, 2),        # the 24^3 level's boxes (8 x 8 x 8 voxels, dword halo rows)
    (2, 64, 48, 8, 16, 24, 2),
    (4, 64, 128, 12, 12, 12, 3),    # the 12^3 level: boxes of 3 x 3 x 6 tiles, 54 of the 64 tile lanes busy
    (4, 32, 128, 18, 6, 12, 3),
    (2, 128, 384, 12, 12, 12, 3),
    (8, 32, 64, 12, 8, 20, 2),      # partly filled 8 x 8 x 8 boxes (tiles outside the volume are not stored)
    (8, 128, 256, 6, 6, 6, 2),      # 128 (box, channel group) entries: the contraction is cut into 2 slices (conv_wino.hip SPLIT)
    (8, 256, 256, 6, 6, 6, 2),      # unet_3D center.conv2 / V-Net block five: 2 slices of 32 chunks
    (4, 256, 256, 6, 6, 6, 2),      # the teacher's half batch: 4 slices
    (4, 128, 128, 12, 12, 12, 3),   # 12^3, half batch: 2 slices of the 3 x 3 x 6-tile boxes
    (2, 64, 128, 6, 6, 6, -1),      # too few entries even when split: direct kernel
]
# (N, Cin, Cout, D, H, W) -> contraction slices mis_conv3d_wino_fwd_ws uses
SPLITS = {(8, 128, 256, 6, 6, 6): 2, (8, 256, 256, 6, 6, 6): 2, (4, 256, 256, 6, 6, 6): 4, (4, 128, 128, 12, 12, 12): 2,
          (4, 64, 128, 12, 12, 12): 2, (2, 128, 384, 12, 12, 12): 1, (1, 32, 32, 8, 8, 8): 1,
          (2, 64, 48, 8, 16, 24): 2,      # whole 8 x 8 x 8 boxes split like partly filled ones (entry count alone decides)
          (8, 32, 64, 12, 8, 20): 1}


@pytest.mark.parametrize("case", FWD_CASES)
def test_wino_forward_and_data_gradient(case):
    ops = _ops()
    N, Cin, Cout, D, H, W, variant = case
    assert ops.conv_wino_select(N, Cin, Cout, D, H, W, (3, 3, 3)) == variant
    if variant < 0:
        return
    if case[:6] in SPLITS:
        from mis_hip import lib
        assert lib.load().mis_conv3d_wino_fwd_splits(N, Cin, Cout, D, H, W, variant) == SPLITS[case[:6]]
    x = _rand(N, Cin, D, H, W, seed=1).requires_grad_(True)
    w = _rand(Cout, Cin, 3, 3, 3, seed=2, scale=0.2).requires_grad_(True)
    b = _rand(Cout, seed=3)
    y_ref = F.conv3d(x, w, b, padding=1)
    dy = _rand(*y_ref.shape, seed=4)
    y_ref.backward(dy)

    xd, wd, bd, dyd = x.detach().float().cuda(), w.detach().float().cuda(), b.float().cuda(), dy.float().cuda()
    y = torch.full((N, Cout, D, H, W), float("nan"), device="cuda")
    ops.conv_fwd(xd, ops.conv_pack(wd, 4), bd, y, Cin, Cout, (3, 3, 3), wino=variant)
    _close(y, y_ref)

    # the same launch on dy with the flipped / transposed transformed filter is the data gradient
    vb = ops.conv_wino_select(N, Cout, Cin, D, H, W, (3, 3, 3))
    if vb >= 0:
        dx = torch.full((N, Cin, D, H, W), float("nan"), device="cuda")
        ops.conv_fwd(dyd, ops.conv_pack(wd, 5), None, dx, Cout, Cin, (3, 3, 3), wino=vb)
        _close(dx, x.grad)

    # ... and both agree with the direct kernels
    yd = torch.empty_like(y)
    ops.conv_fwd(xd, ops.conv_pack(wd, 0), bd, yd, Cin, Cout, (3, 3, 3))
    _close(y, yd)


@pytest.mark.parametrize("N,Cin,Cout,D,H,W,per_sample", [(2, 16, 16, 4, 8, 32, True), (3, 32, 32, 4, 8, 16, False),
                                                         (2, 48, 16, 8, 4, 64, True), (4, 64, 128, 12, 12, 12, True),
                                                         (4, 32, 64, 12, 12, 12, False),
                                                         # split contraction: the statistics come from the reduction launch
                                                         (8, 128, 256, 6, 6, 6, True), (8, 256, 256, 6, 6, 6, False),
                                                         (4, 128, 128, 12, 12, 12, False)])
def test_wino_fused_statistics(N, Cin, Cout, D, H, W, per_sample):
    """The per-box (sum, sumsq) partials of the Winograd epilogue + mis_norm_stats_finalize == the statistics of the
    InstanceNorm / BatchNorm that follows the conv (reference utils.py:104-107: Conv3d -> norm -> ReLU)."""
    ops = _ops()
    v = ops.conv_wino_select(N, Cin, Cout, D, H, W, (3, 3, 3))
    assert v >= 0
    T = ops.conv_stat_tiles(N, Cin, Cout, D, H, W, (3, 3, 3), wino=v)
    assert T > 0
    x = _rand(N, Cin, D, H, W, seed=21)
    w = _rand(Cout, Cin, 3, 3, 3, seed=22, scale=0.3)
    b = _rand(Cout, seed=23)
    y_ref = F.conv3d(x, w, b, padding=1)
    part = torch.full((Cout * N * T, 2), float("nan"), device="cuda")
    y = torch.empty(N, Cout, D, H, W, device="cuda")
    strides = (T, Cout * T) if per_sample else (N * T, T)
    ops.conv_fwd(x.float().cuda(), ops.conv_pack(w.float().cuda(), 4), b.float().cuda(), y, Cin, Cout, (3, 3, 3),
                 stat=(part, *strides), wino=v)
    _close(y, y_ref)
    assert torch.isfinite(part).all()
    G = N * Cout if per_sample else Cout
    mean = torch.empty(G, device="cuda"); rstd = torch.empty(G, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    ops.norm_stats_finalize(part, N, Cout, D * H * W, T, per_sample, 1e-5, mean, rstd,
                            None if per_sample else rm, None if per_sample else rv, None if per_sample else nbt)
    dims = (2, 3, 4) if per_sample else (0, 2, 3, 4)
    m_ref = y_ref.mean(dim=dims).flatten()
    v_ref = y_ref.var(dim=dims, unbiased=False).flatten()
    _close(mean, m_ref, rtol=1e-5, atol=1e-6)
    _close(rstd, 1.0 / torch.sqrt(v_ref + 1e-5), rtol=1e-5, atol=1e-6)


def test_wino_channel_slices_of_concat_buffers():
    """Input = a channel slice of a concat buffer, output = a slice of another one (batch strides != C * S)."""
    ops = _ops()
    N, C0, Cin, Cout, D, H, W = 2, 16, 16, 32, 4, 8, 32
    cat = _rand(N, C0 + Cin, D, H, W, seed=5).float().cuda()
    w = _rand(Cout, Cin, 3, 3, 3, seed=6, scale=0.2).float().cuda()
    ycat = torch.zeros(N, 16 + Cout, D, H, W, device="cuda")
    v = ops.conv_wino_select(N, Cin, Cout, D, H, W, (3, 3, 3))
    assert v >= 0
    ops.conv_fwd(cat[:, C0:], ops.conv_pack(w, 4), None, ycat[:, 16:], Cin, Cout, (3, 3, 3), wino=v)
    ref = F.conv3d(cat[:, C0:].cpu().double(), w.cpu().double(), padding=1)
    _close(ycat[:, 16:], ref)
    assert ycat[:, :16].abs().max().item() == 0.0
    # weight gradient from slices
    dy = _rand(N, 16 + Cout, D, H, W, seed=7).float().cuda()
    dw = torch.empty_like(w)
    ops.conv_wgrad(cat[:, C0:], dy[:, 16:], dw, (3, 3, 3))
    wr = w.cpu().double().requires_grad_(True)
    F.conv3d(cat[:, C0:].cpu().double(), wr, padding=1).backward(dy[:, 16:].cpu().double())
    _close(dw, wr.grad, rtol=3e-4, atol=1e-4)


def test_split_forward_on_channel_slices_of_concat_buffers():
    """The split-contraction launch (6^3 level) with input / output views of wider buffers: the partial outputs live in the
    workspace, the reduction writes through the output's batch stride."""
    ops = _ops()
    N, C0, Cin, Cout, S = 8, 32, 128, 256, 6
    cat = _rand(N, C0 + Cin, S, S, S, seed=25).float().cuda()
    w = _rand(Cout, Cin, 3, 3, 3, seed=26, scale=0.2).float().cuda()
    b = _rand(Cout, seed=27).float().cuda()
    ycat = torch.zeros(N, 16 + Cout, S, S, S, device="cuda")
    v = ops.conv_wino_select(N, Cin, Cout, S, S, S, (3, 3, 3))
    assert v == 2
    keep, ops.DISPATCH = ops.DISPATCH, set()
    try:
        ops.conv_fwd(cat[:, C0:], ops.conv_pack(w, 4), b, ycat[:, 16:], Cin, Cout, (3, 3, 3), wino=v)
        assert "wino_fwd_split:v2@6" in ops.DISPATCH
    finally:
        ops.DISPATCH = keep
    ref = F.conv3d(cat[:, C0:].cpu().double(), w.cpu().double(), b.cpu().double(), padding=1)
    _close(ycat[:, 16:], ref)
    assert ycat[:, :16].abs().max().item() == 0.0
    # deterministic
    y2 = torch.zeros_like(ycat)
    ops.conv_fwd(cat[:, C0:], ops.conv_pack(w, 4), b, y2[:, 16:], Cin, Cout, (3, 3, 3), wino=v)
    assert torch.equal(ycat, y2)


def test_flat_weight_gradient_from_channel_slices():
    """12^3 level: x = the skip half of a decoder's concat buffer, dy = a slice too (batch strides != C * S)."""
    ops = _ops()
    N, C0, Cin, Cout, S = 2, 32, 32, 16, 12
    cat = _rand(N, C0 + Cin, S, S, S, seed=15).float().cuda()
    dy = _rand(N, 16 + Cout, S, S, S, seed=17).float().cuda()
    dw = torch.full((Cout, Cin, 3, 3, 3), float("nan"), device="cuda")
    ops.conv_wgrad(cat[:, C0:], dy[:, 16:], dw, (3, 3, 3))
    wr = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(cat[:, C0:].cpu().double(), wr, padding=1).backward(dy[:, 16:].cpu().double())
    _close(dw, wr.grad, rtol=1e-5, atol=1e-6)


# N, Cin, Cout, D, H, W
WGRAD_CASES = [
    (1, 16, 16, 2, 4, 32),
    (2, 16, 16, 4, 8, 64),
    (1, 24, 40, 6, 4, 32),        # ragged channel blocks on both sides
    (2, 48, 16, 4, 8, 96),
    (2, 16, 32, 4, 8, 16),
    (1, 48, 16, 8, 4, 48),
    (3, 8, 8, 4, 4, 16),
    (2, 32, 16, 8, 4, 8),         # stages of 8 x 4 x 8 voxels (24^3 level)
    (1, 64, 64, 8, 12, 24),       # (W = 24, H = 12: box kernel -- the three-run ring needs H % 8)
    # z-ring of three runs per stage (round 4: W % 24 / H % 8 -> 2 x 8 x 24-voxel stages, the 24^3 level): planes in a ring of 6,
    # dy straight into registers
    (1, 16, 16, 2, 8, 24),        # one stage
    (1, 16, 16, 4, 8, 24),        # two stages: prologue + one fill
    (2, 24, 40, 6, 16, 24),       # ragged channel blocks, two columns along y, three stages (the ring wraps once)
    (1, 16, 32, 24, 24, 24),      # a whole 24^3 volume: 12 stages per column, the ring wraps four times, all y faces
    (3, 32, 16, 10, 8, 48 + 24),  # W = 72: three columns along x (every x face class), 5 stages
    (8, 64, 64, 24, 24, 24),      # unet_3D conv3.conv2 at the full batch
    # z-ring kernels (round 4: W % 32 / H % 4 -> 2 x 4 x 32-voxel stages, W % 16 / H % 8 -> 2 x 8 x 16); a workgroup owns a
    # contiguous range of the (image, y, x, z) stage sequence
    (1, 16, 16, 4, 4, 32),        # two stages, eight workgroups: single-stage and empty ranges
    (8, 16, 16, 4, 4, 32),        # one whole two-stage column per workgroup: the prologue alone feeds it
    (1, 16, 16, 64, 4, 32),       # ranges of four stages inside one column: segments that start mid-column (real z halo)
    (2, 16, 16, 12, 8, 32),       # ranges that cross column ends (drain + prologue inside the kernel)
    (3, 16, 16, 16, 16, 64),      # many columns, > 256 stages: every y / x face class
    (1, 24, 40, 8, 8, 32),        # ragged channel blocks, 6 pairs
    (2, 32, 32, 8, 16, 48),       # 2 x 8 x 16-voxel stages (48^3 level geometry), 3 columns along x
    (1, 16, 48, 20, 8, 16),       # 10 stages per column, 2 x 8 x 16
    (2, 16, 16, 24, 24, 96),      # a 96^3-level slab: long ranges, the steady-state loop
    # flat form (round 4): 12 x 12 planes, lanes fetch their own patches from a packed zero-padded copy, no LDS stage
    (1, 16, 16, 12, 12, 12),      # 216 tiles, 54 chunks
    (3, 24, 40, 2, 12, 12),       # 108 tiles, ragged channel blocks on both sides
    (1, 16, 16, 6, 12, 12),       # 108 tiles over 8 workgroup waves: a ragged last chunk is impossible (36 | tiles), odd runs are
    (2, 64, 128, 12, 12, 12),     # 32 pairs, several waves per pair
    (8, 128, 128, 12, 12, 12),    # unet_3D conv4 at the full batch
    (2, 16, 32, 4, 12, 12),       # a slab of 12 x 12 planes
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wino_weight_gradient(case):
    ops = _ops()
    from mis_hip import lib
    N, Cin, Cout, D, H, W = case
    assert lib.load().mis_conv3d_wino_wgrad_select(N, Cin, Cout, D, H, W) >= 0
    x = _rand(N, Cin, D, H, W, seed=11)
    dy = _rand(N, Cout, D, H, W, seed=12)
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w, padding=1).backward(dy)
    xd, dyd = x.float().cuda(), dy.float().cuda()
    dw = torch.full((Cout, Cin, 3, 3, 3), float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, dw, (3, 3, 3))
    _close(dw, w.grad, rtol=1e-5, atol=1e-6)
    # deterministic (fixed summation order), and `accumulate` adds
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw2, (3, 3, 3))
    assert torch.equal(dw, dw2)
    ops.conv_wgrad(xd, dyd, dw2, (3, 3, 3), accumulate=True)
    _close(dw2, 2 * w.grad, rtol=1e-5, atol=1e-6)
    # the direct kernel agrees
    keep, ops.WINO = ops.WINO, 0
    try:
        dwd = torch.empty_like(dw)
        ops.conv_wgrad(xd, dyd, dwd, (3, 3, 3))
    finally:
        ops.WINO = keep
    _close(dw, dwd, rtol=1e-5, atol=1e-6)


def test_ring_weight_gradient_in_units_mode():
    """The z-ring kernel deals column SEGMENTS to the workgroups of an XCD interleaved at the 96^3 level and contiguous stage
    ranges elsewhere (by tensor size): MIS_WGRAD_RING_UNITS=1 forces the units mode on the small ring shapes of WGRAD_CASES
    (segment starts inside a column, single-stage segments, every face class) in a fresh process."""
    import subprocess
    import sys
    code = """
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r)
from mis_hip import ops, lib
cases = [(1, 16, 16, 4, 4, 32), (8, 16, 16, 4, 4, 32), (1, 16, 16, 64, 4, 32), (2, 16, 16, 12, 8, 32), (3, 16, 16, 16, 16, 64),
         (1, 24, 40, 8, 8, 32), (2, 32, 32, 8, 16, 48), (1, 16, 48, 20, 8, 16), (1, 16, 16, 2, 4, 32),
         (1, 16, 16, 2, 8, 24), (2, 24, 40, 6, 16, 24), (1, 16, 32, 24, 24, 24), (3, 32, 16, 10, 8, 72)]
for N, Cin, Cout, D, H, W in cases:
    assert lib.load().mis_conv3d_wino_wgrad_select(N, Cin, Cout, D, H, W) in (3, 4, 6)
    g = torch.Generator().manual_seed(N * 1000 + D)
    x = torch.randn(N, Cin, D, H, W, generator=g, dtype=torch.float64)
    dy = torch.randn(N, Cout, D, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w, padding=1).backward(dy)
    dw = torch.full((Cout, Cin, 3, 3, 3), float('nan'), device='cuda')
    ops.conv_wgrad(x.float().cuda(), dy.float().cuda(), dw, (3, 3, 3))
    err = (dw.cpu().double() - w.grad).abs().max().item()
    assert err <= 1e-6 + 1e-5 * w.grad.abs().max().item(), (N, Cin, Cout, D, H, W, err)
print('units mode ok')
""" % (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT)
    env = dict(os.environ, MIS_WGRAD_RING_UNITS="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "units mode ok" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]


def test_wino_select_and_refusal():
    """Geometries the Winograd kernels do not cover are reported by the select functions and refused by the entry points
    (no silent fallback inside the library: the caller picks the direct kernel)."""
    ops = _ops()
    from mis_hip import lib
    L = lib.load()
    sel = lambda *a: ops.conv_wino_select(*a, (3, 3, 3))
    assert sel(2, 16, 16, 96, 96, 96) == 0 and sel(2, 32, 32, 48, 48, 48) == 1
    assert sel(2, 1, 16, 96, 96, 96) == -1          # first layer: 1 input channel
    assert sel(2, 16, 2, 96, 96, 96) == -1          # Cout not a multiple of 16
    assert sel(2, 64, 64, 24, 24, 24) == 2
    assert sel(4, 128, 128, 12, 12, 12) == 3        # 12^3: 6 x 6 x 12 boxes, when there are enough of them
    assert sel(1, 32, 16, 12, 12, 12) == -1 and sel(8, 128, 256, 6, 6, 6) == 2
    assert sel(4, 128, 256, 6, 6, 6) == 2 and sel(1, 128, 256, 6, 6, 6) == -1      # 64 entries x 2 contraction slices; 16 x 4
    assert sel(2, 16, 16, 6, 6, 30) == -1
    assert ops.conv_wino_select(2, 16, 16, 1, 64, 64, (3, 3)) == ops.WINO2D + 2      # 2-D: conv_wino2d.hip, 8 x 32-pixel boxes
    assert ops.conv_wino_select(2, 16, 32, 1, 64, 64, (3, 3)) == ops.WINO2D + 3 and ops.conv_wino_select(2, 16, 16, 1, 48, 48, (3, 3)) == ops.WINO2D
    assert ops.conv_wino_select(2, 16, 32, 1, 16, 16, (3, 3)) == ops.WINO2D + 1 and ops.conv_wino_select(2, 16, 16, 1, 24, 32, (3, 3)) == ops.WINO2D + 2
    x = torch.zeros(1, 16, 6, 6, 30, device="cuda")
    y = torch.zeros(1, 16, 6, 6, 30, device="cuda")
    wt = ops.conv_pack(torch.zeros(16, 16, 3, 3, 3, device="cuda"), 4)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.conv_fwd(x, wt, None, y, 16, 16, (3, 3, 3), wino=0)
    assert L.mis_conv3d_wino_wgrad_select(1, 16, 16, 6, 6, 30) == -1
    wsel = L.mis_conv3d_wino_wgrad_select
    assert wsel(8, 16, 16, 96, 96, 96) == 3 and wsel(8, 32, 32, 48, 48, 48) == 4      # z-ring kernels
    assert wsel(8, 64, 64, 24, 24, 24) == 6 and wsel(1, 16, 16, 2, 4, 32) == 3          # three-run ring at 24^3
    assert wsel(1, 64, 64, 8, 12, 24) == 2 and wsel(2, 32, 16, 8, 4, 8) == 2            # box kernel: H % 8 != 0 / W = 8
    assert wsel(1, 16, 16, 6, 4, 16) == -1 and wsel(1, 16, 16, 4, 4, 16) == 1           # 2 x 8 x 16 stages need H % 8
    assert wsel(1, 16, 16, 3, 4, 32) == -1                                              # odd depth: no stage of two planes
    assert wsel(8, 128, 128, 12, 12, 12) == 5 and wsel(2, 16, 32, 4, 12, 12) == 5       # flat form: 12 x 12 planes
    assert wsel(8, 128, 256, 6, 6, 6) == -1 and wsel(8, 128, 128, 10, 10, 10) == -1     # 6^3 stays on the direct kernel
    with pytest.raises(RuntimeError):
        ops.conv_pack(torch.zeros(16, 16, 3, 3, device="cuda"), 4)      # the transform is defined for 3x3x3 only


def test_wino_full_size_layer_matches_direct():
    """One config-3 layer at its real size (8 x 16 -> 16 at 96^3, BASELINE configs[2]): Winograd == direct within fp32
    rounding, forward and weight gradient."""
    ops = _ops()
    N, C, S = 8, 16, 96
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, C, S, S, S, device="cuda", generator=g)
    dy = torch.randn(N, C, S, S, S, device="cuda", generator=g)
    w = torch.randn(C, C, 3, 3, 3, device="cuda", generator=g) * 0.05
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    ops.conv_fwd(x, ops.conv_pack(w, 0), None, y0, C, C, (3, 3, 3))
    ops.conv_fwd(x, ops.conv_pack(w, 4), None, y1, C, C, (3, 3, 3), wino=ops.conv_wino_select(N, C, C, S, S, S, (3, 3, 3)))
    assert (y0 - y1).abs().max().item() <= 2e-5 * y0.abs().max().item()
    d0, d1 = torch.empty_like(w), torch.empty_like(w)
    ops.conv_wgrad(x, dy, d1, (3, 3, 3))
    keep, ops.WINO = ops.WINO, 0
    try:
        ops.conv_wgrad(x, dy, d0, (3, 3, 3))
    finally:
        ops.WINO = keep
    assert (d0 - d1).abs().max().item() <= 2e-5 * d0.abs().max().item()


@pytest.mark.parametrize("N,D,H,W", [(1, 4, 8, 32), (2, 8, 16, 64), (3, 4, 8, 96), (2, 1, 32, 32), (3, 1, 64, 96)])
def test_first_layer_weight_gradient(N, D, H, W):
    """Conv3d(1 -> 16) (unet_3D.py:28 conv1, vnet.py:123 block_one) and, D == 1, Conv2d(1 -> 16) (unet.py:37): the
    taps-as-columns MFMA kernel (conv_wgrad_cin1.hip) behind mis_conv_wgrad, against torch fp64; deterministic;
    `accumulate` adds."""
    ops = _ops()
    k = (3, 3, 3) if D > 1 else (3, 3)
    x = _rand(N, 1, D, H, W, seed=31)
    dy = _rand(N, 16, D, H, W, seed=32)
    w = torch.zeros(16, 1, *((3, 3, 3) if D > 1 else (1, 3, 3)), dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w, padding=(1, 1, 1) if D > 1 else (0, 1, 1)).backward(dy)
    xd, dyd = x.float().cuda(), dy.float().cuda()
    dw = torch.full(tuple(w.shape), float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, dw, k)
    _close(dw, w.grad, rtol=1e-5, atol=1e-6)
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw2, k)
    assert torch.equal(dw, dw2)
    ops.conv_wgrad(xd, dyd, dw2, k, accumulate=True)
    _close(dw2, 2 * w.grad, rtol=1e-5, atol=1e-6)


# N, Cin, Cout, H, W
W2D_CASES = [(1, 16, 16, 16, 16), (2, 16, 32, 32, 48), (3, 32, 16, 16, 32), (2, 64, 64, 32, 32), (1, 8, 16, 48, 16),
             (2, 12, 48, 16, 64),
             # 8 x 32-pixel boxes (W % 32 == 0, H % 8 == 0): heights that are not multiples of 16, several boxes in both directions
             (2, 16, 16, 24, 96), (3, 32, 64, 8, 32), (2, 16, 16, 64, 256), (1, 24, 32, 40, 64)]


@pytest.mark.parametrize("case", W2D_CASES)
def test_wino2d_forward_and_data_gradient(case):
    """F(2x2, 3x3) (conv_wino2d.hip) for nn.Conv2d(k=3, padding=1) of the 2-D UNet's ConvBlock (unet.py:30-45):
    forward, data gradient (same launch on dy with pack mode 7), fused statistics, against torch fp64 / the direct kernel."""
    ops = _ops()
    N, Cin, Cout, H, W = case
    v = ops.conv_wino_select(N, Cin, Cout, 1, H, W, (3, 3))
    assert v >= ops.WINO2D
    x = _rand(N, Cin, H, W, seed=41).requires_grad_(True)
    w = _rand(Cout, Cin, 3, 3, seed=42, scale=0.2).requires_grad_(True)
    b = _rand(Cout, seed=43)
    y_ref = F.conv2d(x, w, b, padding=1)
    dy = _rand(*y_ref.shape, seed=44)
    y_ref.backward(dy)
    xd = x.detach().float().cuda().unsqueeze(2).contiguous()
    wd, bd = w.detach().float().cuda(), b.float().cuda()
    dyd = dy.float().cuda().unsqueeze(2).contiguous()
    T = ops.conv_stat_tiles(N, Cin, Cout, 1, H, W, (3, 3), wino=v)
    assert T == ((H // 8) * (W // 32) if v >= ops.WINO2D + 2 else (H // 16) * (W // 16))
    part = torch.full((Cout * N * T, 2), float("nan"), device="cuda")
    y = torch.full((N, Cout, 1, H, W), float("nan"), device="cuda")
    ops.conv_fwd(xd, ops.conv_pack(wd, 6), bd, y, Cin, Cout, (3, 3), stat=(part, N * T, T), wino=v)
    _close(y[:, :, 0], y_ref)
    mean = torch.empty(Cout, device="cuda"); rstd = torch.empty(Cout, device="cuda")
    ops.norm_stats_finalize(part, N, Cout, H * W, T, False, 1e-5, mean, rstd)
    _close(mean, y_ref.mean(dim=(0, 2, 3)), rtol=1e-5, atol=1e-6)
    _close(rstd, 1.0 / torch.sqrt(y_ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5), rtol=1e-5, atol=1e-6)
    vb = ops.conv_wino_select(N, Cout, Cin, 1, H, W, (3, 3))
    if vb >= 0:
        dx = torch.full((N, Cin, 1, H, W), float("nan"), device="cuda")
        ops.conv_fwd(dyd, ops.conv_pack(wd, 7), None, dx, Cout, Cin, (3, 3), wino=vb)
        _close(dx[:, :, 0], x.grad)
    yd = torch.empty_like(y)
    ops.conv_fwd(xd, ops.conv_pack(wd, 0), bd, yd, Cin, Cout, (3, 3))
    _close(y, yd)
    # geometries the 2-D kernel does not cover
    assert ops.conv_wino_select(2, 16, 16, 1, 56, 56, (3, 3)) == -1 and ops.conv_wino_select(2, 1, 16, 1, 64, 64, (3, 3)) == -1
    assert ops.conv_wino_select(2, 16, 4, 1, 64, 64, (3, 3)) == -1


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(1, 16, 16, 8, 16), (2, 16, 32, 32, 48), (3, 24, 40, 16, 32), (2, 64, 64, 32, 32),
                                            (1, 8, 16, 48, 16), (2, 32, 48, 24, 64),
                                            # stages of 4 x 32 pixels (W % 32 == 0): H not a multiple of 8, many stages, ragged blocks
                                            (2, 16, 16, 12, 96), (5, 16, 16, 64, 256), (1, 40, 24, 20, 32)])
def test_wino2d_weight_gradient(N, Cin, Cout, H, W):
    """F(2x2, 3x3) weight gradient (conv_wino2d_wgrad.hip) behind ops.conv_wgrad for the 2-D UNet's 3x3 convs, against
    torch fp64 and the direct kernel; ragged channel blocks; deterministic; `accumulate` adds."""
    ops = _ops()
    from mis_hip import lib
    assert lib.load().mis_conv2d_wino_wgrad_select(N, Cin, Cout, H, W) >= 0
    x = _rand(N, Cin, H, W, seed=51)
    dy = _rand(N, Cout, H, W, seed=52)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(dy)
    xd, dyd = x.float().cuda().unsqueeze(2).contiguous(), dy.float().cuda().unsqueeze(2).contiguous()
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, dw, (3, 3))
    _close(dw, w.grad, rtol=1e-5, atol=1e-6)
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(xd, dyd, dw2, (3, 3))
    assert torch.equal(dw, dw2)
    ops.conv_wgrad(xd, dyd, dw2, (3, 3), accumulate=True)
    _close(dw2, 2 * w.grad, rtol=1e-5, atol=1e-6)
    keep, ops.WINO = ops.WINO, 0
    try:
        dwd = torch.empty_like(dw)
        ops.conv_wgrad(xd, dyd, dwd, (3, 3))
    finally:
        ops.WINO = keep
    _close(dw, dwd, rtol=1e-5, atol=1e-6)
    assert lib.load().mis_conv2d_wino_wgrad_select(2, 16, 16, 28, 28) == -1      # 28 x 28: direct kernel


@pytest.mark.parametrize("N,Cdy,Cda,D,H,W,slope", [(2, 16, 16, 8, 8, 32, 0.0), (3, 32, 32, 8, 8, 16, 0.0), (2, 48, 16, 4, 8, 32, 0.01),
                                                   (8, 16, 16, 4, 4, 96, 0.0), (2, 32, 32, 4, 8, 48, 0.0)])
def test_wino_data_gradient_with_norm_backward_partials(N, Cdy, Cda, D, H, W, slope):
    """mis_conv3d_wino_dgrad_norm: the data gradient is the plain Winograd launch's, and its (sum dz, sum dz * x) partials
    + mis_norm_act_bwd_tiles give the InstanceNorm + (Leaky)ReLU backward of mis_norm_act_bwd (reference: autograd of
    nn.InstanceNorm3d -> nn.ReLU between the two convs of UnetConv3, utils.py:105-109)."""
    ops = _ops()
    v = ops.conv_wino_select(N, Cdy, Cda, D, H, W, (3, 3, 3))
    assert v in (0, 1)
    xn = (_rand(N, Cda, D, H, W, seed=31) * 2 + _rand(N, Cda, 1, 1, 1, seed=32)).float().cuda()     # per-(n, c) offsets
    dy = _rand(N, Cdy, D, H, W, seed=33).float().cuda()
    w = _rand(Cdy, Cda, 3, 3, 3, seed=34, scale=0.2).float().cuda()      # forward conv: Cda -> Cdy channels
    wpd = ops.conv_pack(w, 5)
    mean, rstd = torch.empty(N * Cda, device="cuda"), torch.empty(N * Cda, device="cuda")
    ops.norm_stats(xn, True, 1e-5, mean, rstd)
    # reference: plain data gradient, then the three-pass normalisation backward
    da_ref = torch.empty(N, Cda, D, H, W, device="cuda")
    ops.conv_fwd(dy, wpd, None, da_ref, Cdy, Cda, (3, 3, 3), wino=v)
    dx_ref = torch.empty_like(da_ref)
    ops.norm_act_bwd(xn, da_ref, dx_ref, True, mean, rstd, None, None, slope)
    # fused
    T = ops.conv_stat_tiles(N, Cdy, Cda, D, H, W, (3, 3, 3), wino=v)
    part = torch.full((N * Cda * T, 2), float("nan"), device="cuda")
    da = torch.full((N, Cda, D, H, W), float("nan"), device="cuda")
    assert ops.conv_dgrad_norm(dy, wpd, da, Cdy, Cda, xn, mean, slope, part, v) == T
    assert torch.equal(da, da_ref)
    assert torch.isfinite(part).all()
    sums = torch.empty(N * Cda, 2, device="cuda")
    dx = torch.full_like(da, float("nan"))
    ops.norm_act_bwd_tiles(xn, da, dx, mean, rstd, slope, part, T, sums)
    # the sums against torch fp64
    xh = (xn.double() - mean.double().view(N, Cda, 1, 1, 1)) * rstd.double().view(N, Cda, 1, 1, 1)
    dz = torch.where(xh > 0, da.double(), da.double() * slope)
    S = D * H * W
    s1 = dz.sum(dim=(2, 3, 4)).flatten() / S
    s2 = (dz * xh).sum(dim=(2, 3, 4)).flatten() / S
    _close(sums[:, 0], s1, rtol=2e-5, atol=1e-6)
    _close(sums[:, 1], s2, rtol=2e-5, atol=1e-6)
    _close(dx, dx_ref, rtol=2e-5, atol=2e-6)
    # only the sums (first layer: the consumer applies the backward itself)
    sums2 = torch.empty_like(sums)
    ops.norm_act_bwd_tiles(xn, da, None, mean, rstd, slope, part, T, sums2)
    assert torch.equal(sums, sums2)

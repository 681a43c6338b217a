"""The data-parallel overlap contract of ``Plan.backward(on_progress=...)`` / ``SwinPlan.backward`` on one GPU.

``mis_hip.dist.GradBucketer`` all-reduces a bucket of the flat gradient buffer as soon as ``on_progress(lo)`` reports
that everything at offsets >= lo is final (``dist.param_progress``: the highest gradient range of the ops still to run).
A gradient written AFTER its range was reported final would be exchanged stale -- silently wrong, and only on
multi-GPU runs.  Here every network of the path runs a real backward with a callback that snapshots the reported suffix;
each snapshot must equal the same range of the finished buffer (the previous backward left different values there, so a
late write shows).  Every parameter must also be owned by some op, or its gradient must be exactly zero (conv biases in
front of a normalisation, UNETR's unused cls_token).

Reference: the reference is single-GPU; this is the standard DDP bucket contract."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    ("unet2d", (2, 1, 64, 64), 4),
    ("unet2d_deconv", (2, 1, 64, 64), 4),       # UpBlock(bilinear=False): UpConv2dOp owns the ConvTranspose2d parameters
    ("unet3d", (2, 1, 32, 32, 32), 2),
    ("unet3d", (2, 1, 96, 96, 96), 2),        # the BASELINE geometry: Winograd boxes + fused head + first-layer fusion
    ("vnet", (2, 1, 32, 32, 32), 2),
    ("vnet_groupnorm", (2, 1, 32, 32, 32), 2),
    ("vnet_instancenorm", (2, 1, 32, 32, 32), 2),
    ("vnet_none", (2, 1, 32, 32, 32), 2),
    ("swin", (2, 1, 224, 224), 4),
    ("swin_w8", (2, 1, 256, 256), 4),
    ("unetr", (1, 1, 96, 96, 96), 2),
    ("swinunetr", (1, 1, 64, 64, 64), 2),
]


def _make(kind, C):
    if kind in ("unetr", "swinunetr"):
        from networks.net_factory_3d import net_factory_3d
        return net_factory_3d(kind, 1, C)
    from test_parity_gpu import _build
    return _build(kind, C)[1]()


def _owned_ranges(plan, flat_grad):
    """[lo, hi) of every parameter gradient some op of the plan holds (the scan of dist.param_progress)."""
    base, esz, total = flat_grad.data_ptr(), flat_grad.element_size(), flat_grad.numel()
    out = []
    for op in plan.ops:
        for v in vars(op).values():
            g = getattr(v, "grad", None)
            if isinstance(g, torch.Tensor) and hasattr(v, "data") and g.numel() and \
                    base <= g.data_ptr() < base + total * esz:
                lo = (g.data_ptr() - base) // esz
                out.append((lo, lo + g.numel()))
    return out


@pytest.mark.parametrize("kind,shape,C", CASES, ids=[f"{k}-{'x'.join(map(str, s[2:]))}" for k, s, _ in CASES])
def test_reported_suffix_is_final(kind, shape, C):
    torch.manual_seed(7)
    model = _make(kind, C)
    model.train()
    model.dropout_enabled = False
    x = torch.rand(shape, device="cuda")
    logits = model.forward_raw(x)
    dl = model.logits_grad_buffer()
    # backward #1 leaves other values in the gradient buffer
    dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1)
    model.backward_raw()
    torch.cuda.synchronize()
    before = model.flat_grad.clone()
    dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1 + 0.05)
    snaps = []

    def cb(lo):
        snaps.append((int(lo), model.flat_grad[int(lo):].clone()))      # on the compute stream, as the bucketer's all-reduce

    model.backward_raw(on_progress=cb)
    torch.cuda.synchronize()
    final = model.flat_grad.clone()
    assert not torch.equal(before, final), "the two backward passes must differ for a stale value to show"
    assert snaps and snaps[-1][0] == 0, "the last report must cover the whole buffer"
    los = [lo for lo, _ in snaps]
    assert los == sorted(los, reverse=True), los
    for lo, snap in snaps:
        same = snap.view(torch.int32) == final[lo:].view(torch.int32)
        if not bool(same.all()):
            bad = int((~same).nonzero()[0]) + lo
            name = next((n for n, (off, cnt, _) in model._offsets.items() if off <= bad < off + cnt), "?")
            raise AssertionError(f"{kind}: gradient of {name} (offset {bad}) changed after on_progress({lo})")
    # ownership: a parameter nobody owns must have an exactly-zero gradient
    plan = model._last[0]
    owned = torch.zeros(final.numel(), dtype=torch.bool)
    for lo, hi in _owned_ranges(plan, model.flat_grad):
        owned[lo:hi] = True
    fin = final.cpu()
    for name, (off, cnt, _) in model._offsets.items():
        if not bool(owned[off:off + cnt].all()):
            assert float(fin[off:off + cnt].abs().max()) == 0.0, f"{kind}: {name} has a gradient but no owning op"


def test_weight_gradient_side_stream_is_bit_identical(monkeypatch):
    """MIS_WGRAD_STREAM: weight gradients issued on a side stream beside the data-gradient chain give the same flat
    gradient buffer as the single-stream backward (same kernels, deterministic reductions)."""
    from mis_hip import plan
    res = []
    for on in (False, True):
        monkeypatch.setattr(plan, "WGRAD_STREAM", on)
        torch.manual_seed(11)
        model = _make("unet3d", 2)
        model.train()
        model.dropout_enabled = False
        x = torch.rand((2, 1, 32, 32, 32), device="cuda")
        model.forward_raw(x)
        dl = model.logits_grad_buffer()
        dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1)
        model.backward_raw()
        torch.cuda.synchronize()
        res.append(model.flat_grad.clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("kind,shape,C", [("swin", (2, 1, 224, 224), 4), ("swin_w8", (2, 1, 256, 256), 4),
                                          ("unetr", (1, 1, 96, 96, 96), 2), ("swinunetr", (1, 1, 64, 64, 64), 2)])
def test_finishing_launches_on_the_side_stream_are_bit_identical(monkeypatch, kind, shape, C):
    """MIS_DEFER_FINALS: LayerNorm's affine gradients and the relative-position-bias-table gradient finished on the
    weight-gradient side stream (mis_layernorm_bwd_{parts,final}, mis_window_attention_{bwd_parts,dtable}_ws) give the same
    flat gradient buffer as the one-call forms on the data-gradient chain -- twice in a row (the second backward reuses the
    ops' own workspaces)."""
    from mis_hip import plan
    res = []
    for on in (False, True):
        monkeypatch.setattr(plan, "DEFER", on)
        torch.manual_seed(11)
        model = _make(kind, C)
        model.train()
        model.dropout_enabled = False
        x = torch.rand(shape, device="cuda")
        grads = []
        for it in range(2):
            model.forward_raw(x + 0.1 * it)
            dl = model.logits_grad_buffer()
            dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1)
            model.backward_raw()
            torch.cuda.synchronize()
            grads.append(model.flat_grad.clone())
        res.append(grads)
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind,shape,C", [("swin", (2, 1, 224, 224), 4), ("unetr", (1, 1, 96, 96, 96), 2),
                                          ("swinunetr", (1, 1, 64, 64, 64), 2)])
def test_residual_backward_inside_the_layernorm_backward_is_bit_identical(monkeypatch, kind, shape, C):
    """MIS_SWIN_LNRES: `x = shortcut + drop_path(branch); norm(x)` -- the LayerNorm's backward pass also writes the shortcut's
    and the branch's gradients (mis_layernorm_bwd_residual_parts); same flat gradient buffer as the two passes (DropPath off
    here; tests/test_token_kernels_gpu.py runs a Mean-Teacher step with it on)."""
    from mis_hip import swin_plan
    res = []
    for on in (False, True):
        monkeypatch.setattr(swin_plan, "LNRES", on)
        torch.manual_seed(11)
        model = _make(kind, C)
        model.train()
        model.dropout_enabled = False
        x = torch.rand(shape, device="cuda")
        grads = []
        for it in range(2):
            model.forward_raw(x + 0.1 * it)
            dl = model.logits_grad_buffer()
            dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1)
            model.backward_raw()
            torch.cuda.synchronize()
            grads.append(model.flat_grad.clone())
        assert sum(1 for op in model._last[0].ops if getattr(op, "res_bwd", None) is not None) > 0
        res.append(grads)
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind,shape,C", [("swin", (2, 1, 224, 224), 4), ("unet3d", (2, 1, 32, 32, 32), 2), ("unetr", (1, 1, 96, 96, 96), 2)])
def test_forward_nobody_differentiates_gives_the_same_logits_and_refuses_backward(kind, shape, C):
    """forward_raw(no_backward=True) -- the EMA teacher, validation -- drops what only a backward reads (SwinUnet: the MLPs'
    pre-activations, the 16x expanded tokens of the tail); the logits are the same bits, and backward_raw() after it raises
    instead of differentiating through buffers that were not written."""
    torch.manual_seed(3)
    model = _make(kind, C)
    model.train()
    model.dropout_enabled = False
    x = torch.rand(shape, device="cuda")
    a = model.forward_raw(x).clone()
    b = model.forward_raw(x, no_backward=True).clone()
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="no_backward"):
        model.backward_raw()
    model.forward_raw(x)
    dl = model.logits_grad_buffer()
    dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1)
    model.backward_raw()
    torch.cuda.synchronize()
    assert torch.isfinite(model.flat_grad).all()

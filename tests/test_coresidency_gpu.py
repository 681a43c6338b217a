"""A kernel's results must not depend on which foreign waves share its SIMDs (round 5: a gfx950 erratum made every Winograd
convolution return wrong rows beside bf16-MFMA waves -- packed fp32 ops with op_sel:[0,1]).  Forward + backward
of the CNNs run beside mis_debug_spin waves (register-light, LDS-free, one pipe kept busy) on a second stream; logits and the
flat gradient must equal the quiet run bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(kind):
    if kind == "unet2d":
        from networks.net_factory import net_factory
        return net_factory("unet", 1, 4), (16, 1, 256, 256)
    if kind == "swin":
        from networks.net_factory import net_factory
        return net_factory("ViT_Seg", 1, 4), (8, 1, 224, 224)
    from networks.net_factory_3d import net_factory_3d
    return net_factory_3d("unet_3D" if kind == "unet3d" else "vnet", 1, 2), (2, 1, 96, 96, 96)


@pytest.mark.parametrize("kind", ["unet2d", "unet3d", "vnet", "swin"])
def test_results_do_not_depend_on_foreign_waves(kind):
    from mis_hip import lib as _l
    torch.manual_seed(0)
    net, shape = _net(kind)
    net.train()
    net.dropout_enabled = False
    x = torch.rand(shape, device="cuda")
    L = _l.load()
    sink = torch.zeros(1024, device="cuda")
    side = torch.cuda.Stream()

    def run(spin):
        if spin:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _l.check(L.mis_debug_spin(spin, 4096, 40000, _l.ptr(sink), _l.stream_ptr()), "mis_debug_spin")
        y = net.forward_raw(x)
        net.flat_grad.zero_()
        net.backward_raw(torch.full_like(y, 1e-3))
        torch.cuda.synchronize()
        return y.clone(), net.flat_grad.clone()

    ref = run(0)
    assert all(torch.equal(a, b) for a, b in zip(ref, run(0)))                       # deterministic when quiet
    for spin, name in ((1, "bf16 MFMA"), (2, "fp32 MFMA"), (3, "unpacked VALU"), (4, "packed fp32 VALU")):
        for rep in range(2):
            cur = run(spin)
            for what, a, b in zip(("logits", "flat gradient"), cur, ref):
                assert torch.equal(a, b), f"{kind}: {what} changes beside {name} waves (max diff {(a - b).abs().max().item():.2e})"
    with pytest.raises(RuntimeError):
        _l.check(L.mis_debug_spin(9, 1, 1, _l.ptr(sink), _l.stream_ptr()), "mis_debug_spin")

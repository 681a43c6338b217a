"""UNETR (SURVEY s.8 row n4) on the HIP path against oracle/unetr.py -- a torch restatement of the published MONAI
blocks the reference's code/networks/unetr.py assembles.  PARITY UNPINNED: MONAI is not vendored in the reference and
not installed here, so these tests pin the HIP kernels to that restatement, not to the reference's own arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _close(a, b, rtol=2e-4, atol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * b.abs().max().item(), f"max err {err:.3e} vs scale {b.abs().max().item():.3e}"


@pytest.mark.parametrize("B,N,nH", [(2, 216, 12), (3, 27, 2), (1, 256, 1), (2, 70, 3)])
def test_full_attention_fwd_bwd(B, N, nH):
    from mis_hip import tops
    C = nH * 64
    qkv = _rand(B * N, 3 * C, seed=1, scale=0.7).double().requires_grad_(True)
    q, k, v = qkv.view(B, N, 3, nH, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ k.transpose(-2, -1)) * 64 ** -0.5, dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B * N, C)
    dout = _rand(B * N, C, seed=2).double()
    ref.backward(dout)
    qd = qkv.detach().float().cuda()
    out = torch.empty(B * N, C, device="cuda")
    stats = torch.empty(B * nH * N * 2, device="cuda")
    tops.full_attention_fwd(qd, out, stats, B, N, nH, 64 ** -0.5)
    _close(out, ref, rtol=1e-4, atol=1e-5)
    dqkv = torch.full((B * N, 3 * C), float("nan"), device="cuda")
    tops.full_attention_bwd(qd, dout.float().cuda(), dqkv, stats, B, N, nH, 64 ** -0.5)
    _close(dqkv, qkv.grad, rtol=2e-4, atol=1e-5)


def test_patch_embedding_helpers():
    from mis_hip import tops
    B, S, P = 2, 32, 16
    x = _rand(B, 1, S, S, S, seed=3)
    h = S // P
    ref = x.view(B, 1, h, P, h, P, h, P).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B * h ** 3, P ** 3)
    cols = torch.empty(B * h ** 3, P ** 3, device="cuda")
    tops.patch3d_im2col(x.cuda(), cols, P)
    assert torch.equal(cols.cpu(), ref)
    L, C = h ** 3, 64
    t, pos = _rand(B * L, C, seed=4), _rand(L, C, seed=5)
    out = torch.empty(B * L, C, device="cuda")
    tops.add_rowcycle(t.cuda(), pos.cuda(), out, L)
    assert torch.equal(out.cpu(), (t.view(B, L, C) + pos).view(B * L, C))
    dpos = torch.empty(L, C, device="cuda")
    tops.sum_rowcycle(t.cuda(), dpos, L)
    _close(dpos, t.view(B, L, C).sum(0), rtol=1e-6, atol=1e-6)


def _filled(onet, tag=""):
    from oracle import filler
    sd = filler.fill_state_dict({tag + k: v for k, v in onet.new_state().items()})
    return {k[len(tag):]: v for k, v in sd.items()}


def test_unetr_state_dict_and_factory_surface():
    from networks.net_factory_3d import net_factory_3d
    from oracle.unetr import OracleUNETR
    net = net_factory_3d("unetr", 1, 2)
    keys = [s[0] for s in OracleUNETR(2).spec()]
    assert list(net.state_dict().keys()) == keys
    assert sum(p.numel() for p in net.parameters()) == sum(int(np.prod(s[1])) for s in OracleUNETR(2).spec())
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 64, 64, 64, device="cuda"))


@pytest.mark.timeout(1200)
def test_unetr_mean_teacher_step_matches_oracle():
    """One Mean-Teacher step of UNETR (32^3 patches: 8 tokens, same 12-layer encoder and 4-level decoder) against
    oracle.step on the oracle network: logits, losses, gradients, updated weights."""
    from mis_hip.step import MeanTeacherTrainer
    from networks.unetr import UNETR
    from oracle import filler
    from oracle.step import mean_teacher_step
    from oracle.unetr import OracleUNETR
    C, L, it, img = 2, 1, 1200, (32, 32, 32)
    onet = OracleUNETR(C, img_size=img)
    sd0, tsd0 = _filled(onet), _filled(onet, "t.")
    make = lambda: UNETR(1, C, img, feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, conv_block=True)
    model, ema = make(), make()
    model.load_state_dict(sd0)
    ema.load_state_dict(tsd0)
    model.train(); ema.train()
    volume = filler.image((2, 1) + img, "volume")
    label = filler.labels((2,) + img, C, torch.int64)
    noise = filler.noise((1, 1) + img, "noise")
    tr = MeanTeacherTrainer(model, ema, labeled_bs=L, num_classes=C, cons_start_iter=0, iter_num=it)
    tr.step(volume.cuda(), label.cuda(), noise=noise.cuda())
    got = tr.losses()
    student = {k: v.clone() for k, v in sd0.items()}
    teacher = {k: v.clone() for k, v in tsd0.items()}
    orc = mean_teacher_step(onet, student, teacher, {}, volume, label, noise, it, labeled_bs=L, num_classes=C,
                            cons_start_iter=0, drop_student="off", drop_teacher="off")
    sl = model._last[0].out.t.cpu().reshape(orc["logits"].shape)
    tl = ema._last[0].out.t.cpu().reshape(orc["teacher_logits"].shape)
    assert (sl - orc["logits"]).abs().max().item() <= 1e-3
    assert (tl - orc["teacher_logits"]).abs().max().item() <= 1e-3
    for k in ("loss", "loss_ce", "loss_dice", "consistency_loss"):
        assert abs(got[k] - orc[k]) <= 2e-4, (k, got[k], orc[k])
    gscale = max(float(g.abs().max()) for g in orc["grads"].values())
    for n, g in model.named_flat(model.flat_grad):
        ref = orc["grads"][n]
        assert (g.cpu() - ref).abs().max().item() <= 0.05 * float(ref.abs().max()) + 2e-3 * gscale, n
    for n, v in model.named_flat(model.flat_param):
        assert (v.cpu() - student[n]).abs().max().item() <= 1e-6 + orc["lr"] * 0.05 * gscale, n


@pytest.mark.timeout(1200)
def test_unetr_forward_at_96_matches_oracle():
    """The factory's geometry (96^3, 216 tokens): eval-mode forward of one volume against the oracle."""
    from networks.net_factory_3d import net_factory_3d
    from oracle import filler
    from oracle.unetr import OracleUNETR
    onet = OracleUNETR(2)
    sd0 = _filled(onet)
    net = net_factory_3d("unetr", 1, 2)
    net.load_state_dict(sd0)
    net.eval()
    x = filler.image((1, 1, 96, 96, 96), "volume")
    with torch.no_grad():
        y = net(x.cuda())
    ref = onet.forward(sd0, x, training=False)
    assert (y.cpu() - ref).abs().max().item() <= 1e-3

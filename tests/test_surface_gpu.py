"""Drop-in operator surface of the reference on the GPU: utils.losses, utils.ramps,
update_ema_variables, the train_*.py command lines, hipGraph replay of the step."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT =os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cv-ssl-mis_amd")


def _rand(*shape, seed=0):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1


@pytest.mark.parametrize("C,shape,ldt", [(4, (32, 32), torch.uint8), (2, (8, 16, 16), torch.int64)])
def test_dice_loss_module_matches_oracle(C, shape, ldt):
    from oracle.losses import dice_loss
    from utils.losses import DiceLoss
    logits = (_rand(3, C, *shape, seed=1) * 3).requires_grad_(True)
    label = torch.randint(0, C, (3, 1) + shape, generator=torch.Generator().manual_seed(2)).to(ldt)
    ref = dice_loss(torch.softmax(logits, 1), label, C)
    ref.backward()
    lg = logits.detach().cuda().requires_grad_(True)
    got = DiceLoss(C)(torch.softmax(lg, dim=1), label.cuda())
    (got * 1.0).backward()
    assert abs(got.item() - ref.item()) < 1e-5
    assert (lg.grad.cpu() - logits.grad).abs().max().item() <= 1e-4 * logits.grad.abs().max().item() + 1e-9
    # softmax=True path and the reference's shape assertion
    got2 = DiceLoss(C)(lg.detach(), label.cuda(), softmax=True)
    assert abs(got2.item() - ref.item()) < 1e-5
    with pytest.raises(AssertionError, match="predict & target shape do not match"):
        DiceLoss(C + 1)(lg.detach(), label.cuda(), softmax=True)


def test_softmax_mse_loss_matches_oracle():
    from oracle.losses import softmax_mse
    from utils.losses import softmax_mse_loss
    a = (_rand(2, 4, 16, 16, seed=3) * 2).requires_grad_(True)
    b = _rand(2, 4, 16, 16, seed=4) * 2
    w = _rand(2, 4, 16, 16, seed=5)
    ref = softmax_mse(a, b)
    (ref * w).sum().backward()
    ad = a.detach().cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    got = softmax_mse_loss(ad, bd)
    (got * w.cuda()).sum().backward()
    assert (got.detach().cpu() - ref.detach()).abs().max().item() < 1e-6
    assert (ad.grad.cpu() - a.grad).abs().max().item() < 1e-6
    assert bd.grad is None            # "Sends gradients to inputs but not the targets"


def test_ramps_and_ema_surface():
    from networks.net_factory import net_factory
    from utils import ramps
    from utils.losses import update_ema_variables
    assert ramps.sigmoid_rampup(0, 200.0) == pytest.approx(math.exp(-5.0))
    assert ramps.sigmoid_rampup(300, 200.0) == 1.0 and ramps.sigmoid_rampup(3, 0) == 1.0
    m, e = net_factory("unet", 1, 4), net_factory("unet", 1, 4)
    p0, e0 = m.flat_param.clone(), e.flat_param.clone()
    update_ema_variables(m, e, 0.99, 0)          # step 0: alpha = 0 -> teacher := student
    assert torch.equal(e.flat_param, p0)
    e.flat_param.copy_(e0)
    update_ema_variables(m, e, 0.99, 10 ** 6)
    assert torch.allclose(e.flat_param, e0 * 0.99 + (1 - 0.99) * p0, rtol=0, atol=1e-7)
    # parameters() iterate in the reference order and alias the flat buffer
    for (n1, a), (n2, b) in zip(m.named_parameters(), e.named_parameters()):
        assert n1 == n2 and a.shape == b.shape
    assert next(iter(m.parameters())).data_ptr() == m.flat_param.data_ptr()


def test_hip_graph_replay_matches_eager():
    """The step captured once in a hipGraph and replayed gives bit-identical training to eager launches."""
    from mis_hip.step import MeanTeacherTrainer
    from networks.net_factory import net_factory
    from oracle import filler
    from oracle.nets import OracleUNet2D
    sd0 = filler.fill_state_dict(OracleUNet2D(1, 4).new_state())
    vol = filler.image((4, 1, 64, 64), "volume").cuda()
    lab = filler.labels((4, 64, 64), 4, torch.uint8).cuda()
    outs = []
    for use_graph in (False, True):
        m, e = net_factory("unet", 1, 4), net_factory("unet", 1, 4)
        m.load_state_dict(sd0); e.load_state_dict(sd0)
        tr = MeanTeacherTrainer(m, e, labeled_bs=2, num_classes=4, cons_start_iter=0, seed=7, iter_num=998,
                                use_graph=use_graph)
        for _ in range(4):
            tr.step(vol, lab)
        torch.cuda.synchronize()
        outs.append((tr.losses(), m.flat_param.clone(), e.flat_param.clone(), tr.iter_num))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert outs[0][3] == outs[1][3] == 1002


def test_teacher_on_side_stream_is_bit_identical(monkeypatch):
    """MIS_TWO_STREAM: the teacher forward on a side stream (eager and captured) trains bit-identically."""
    from mis_hip import step
    from networks.net_factory import net_factory
    from oracle import filler
    from oracle.nets import OracleUNet2D
    sd0 = filler.fill_state_dict(OracleUNet2D(1, 4).new_state())
    vol = filler.image((4, 1, 64, 64), "volume").cuda()
    lab = filler.labels((4, 64, 64), 4, torch.uint8).cuda()
    outs = []
    for two, use_graph in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(step, "TWO_STREAM", two)
        m, e = net_factory("unet", 1, 4), net_factory("unet", 1, 4)
        m.load_state_dict(sd0); e.load_state_dict(sd0)
        tr = step.MeanTeacherTrainer(m, e, labeled_bs=2, num_classes=4, cons_start_iter=0, seed=7, iter_num=998,
                                     use_graph=use_graph)
        for _ in range(4):
            tr.step(vol, lab)
        torch.cuda.synchronize()
        outs.append((tr.losses(), m.flat_param.clone(), e.flat_param.clone(),
                     [b.clone() for _, b in e.named_buffers()]))
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        assert torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
        assert all(torch.equal(a, b) for a, b in zip(o[3], outs[0][3]))


@pytest.mark.parametrize("script,extra", [
    ("train_mean_teacher_2D.py", ["--patch_size", "64", "64", "--batch_size", "4", "--labeled_bs", "2"]),
    ("train_mean_teacher_3D.py", ["--patch_size", "32", "32", "32", "--batch_size", "2", "--labeled_bs", "1"]),
    ("train_mean_teacher_3D.py", ["--model", "vnet", "--patch_size", "32", "32", "32", "--batch_size", "4",
                                  "--labeled_bs", "2"]),
    ("train_uncertainty_aware_mean_teacher_2D.py", ["--patch_size", "64", "64", "--batch_size", "4",
                                                    "--labeled_bs", "2"]),
    ("train_uncertainty_aware_mean_teacher_3D.py", ["--patch_size", "32", "32", "32", "--batch_size", "2",
                                                    "--labeled_bs", "1"]),
    ("train_uncertainty_aware_mean_teacher_ViT_2D.py", ["--patch_size", "224", "224", "--batch_size", "2",
                                                        "--labeled_bs", "1"]),
    ("train_cross_pseudo_supervision_2D.py", ["--patch_size", "64", "64", "--batch_size", "4", "--labeled_bs", "2"]),
    ("train_cross_pseudo_supervision_3D.py", ["--patch_size", "32", "32", "32", "--batch_size", "2",
                                              "--labeled_bs", "1"]),
    ("train_cross_pseudo_supervision_2D_ViT.py", ["--patch_size", "224", "224", "--batch_size", "2",
                                                  "--labeled_bs", "1"]),
    ("train_cross_teaching_between_cnn_transformer_2D.py", ["--patch_size", "224", "224", "--batch_size", "2",
                                                            "--labeled_bs", "1"]),
    ("train_cnn_meet_vit_2D.py", ["--patch_size", "224", "224", "--batch_size", "2", "--labeled_bs", "1"]),
])
def test_train_cli_runs(script, extra, tmp_path):
    env = dict(os.environ, PYTHONPATH=PKG)
    work = tmp_path / "code"
    work.mkdir()
    r = subprocess.run([sys.executable, os.path.join(PKG, script), "--max_iterations", "3", "--exp", "t/MT"] + extra,
                       cwd=str(work), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Training Finished!" in r.stdout
    two_students = "cross_" in script or "cnn_meet_vit" in script
    assert ("iteration 3 : model1 loss :" if two_students else "iteration 3 : loss :") in r.stdout
    logs = list((tmp_path / "model").rglob("log.txt"))
    assert logs and ("model2 loss" if two_students else "loss_dice") in logs[0].read_text()


def test_train_cli_on_resident_dataset(tmp_path):
    """--root_path with a dataset: HBM-resident pool + two-stream sampler + one augmentation launch per batch."""
    rng = np.random.default_rng(0)
    acdc = tmp_path / "data" / "ACDC"
    (acdc / "data" / "slices").mkdir(parents=True)
    names = [f"patient{i // 8:03d}_frame01_slice_{i % 8}" for i in range(80)]
    for n in names:
        h, w = (int(v) for v in rng.integers(40, 70, 2))
        np.savez(acdc / "data" / "slices" / (n + ".npz"), image=rng.random((h, w)).astype(np.float32),
                 label=rng.integers(0, 4, (h, w)).astype(np.uint8))
    (acdc / "train_slices.list").write_text("\n".join(names) + "\n")
    brats = tmp_path / "data" / "BraTS2019"
    (brats / "data").mkdir(parents=True)
    cases = [f"BraTS19_{i}" for i in range(6)]
    for c in cases:
        s = tuple(int(v) for v in rng.integers(30, 44, 3))
        np.savez(brats / "data" / (c + ".npz"), image=rng.random(s).astype(np.float32),
                 label=rng.integers(0, 2, s).astype(np.uint8))
    (brats / "train.txt").write_text("\n".join(cases) + "\n")
    env = dict(os.environ, PYTHONPATH=PKG)
    work = tmp_path / "code"
    work.mkdir()
    for script, extra, note in (
            ("train_mean_teacher_2D.py", ["--root_path", str(acdc), "--patch_size", "64", "64", "--batch_size", "4",
                                          "--labeled_bs", "2", "--labeled_num", "3"], "80 cases of"),
            ("train_mean_teacher_3D.py", ["--root_path", str(brats), "--patch_size", "32", "32", "32", "--batch_size",
                                          "2", "--labeled_bs", "1", "--labeled_num", "2"], "6 cases of")):
        r = subprocess.run([sys.executable, os.path.join(PKG, script), "--max_iterations", "5", "--exp", "t/DS"] + extra,
                           cwd=str(work), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Training Finished!" in r.stdout and "iteration 5 : loss :" in r.stdout
        assert note in r.stdout and "resident in HBM" in r.stdout


def test_backward_after_same_shape_forward_fails_loudly():
    """One static plan per input shape: a second same-shape forward overwrites the activations the first
    loss.backward() needs; torch autograd would keep both graphs, so this must raise, never silently use the wrong
    activations.  A different-shape forward in between is harmless."""
    from networks.net_factory import net_factory
    model = net_factory("unet", 1, 4)
    model.train()
    model.dropout_enabled = False
    x = torch.rand(2, 1, 32, 32, device="cuda")
    y = model(x)
    with torch.no_grad():
        model(torch.rand(1, 1, 32, 32, device="cuda"))     # other shape: other plan
    y.mean().backward()                                    # still valid
    g0 = model.flat_grad.clone()
    y = model(x)
    with torch.no_grad():
        model(torch.rand(2, 1, 32, 32, device="cuda"))     # same shape: pseudo-label / validation style pass
    with pytest.raises(RuntimeError, match="activations were overwritten"):
        y.mean().backward()
    y = model(x)
    y.mean().backward()
    assert torch.equal(model.flat_grad, g0)


@pytest.mark.parametrize("kind", ["uamt2d", "uamt3d", "cnnvit"])
def test_side_stream_trainers_are_bit_identical(kind, monkeypatch):
    """UA-MT (five teacher forwards beside the student's) and CNN-meets-ViT (CNN student + teacher beside the Transformer
    student, the CNN's backward beside the Transformer's) with MIS_TWO_STREAM on and off: same kernels, same order per
    stream, same reduction trees -- bit-identical weights, BatchNorm statistics and losses after two steps (device-side
    Philox dropout and noise included)."""
    from mis_hip import step
    from mis_hip.step import CnnMeetVitTrainer, UAMTTrainer
    from test_grad_progress_gpu import _make
    C = 2 if kind == "uamt3d" else 4
    shape = {"uamt2d": (4, 1, 64, 64), "uamt3d": (2, 1, 32, 32, 32), "cnnvit": (2, 1, 224, 224)}[kind]
    g = torch.Generator().manual_seed(3)
    vol = torch.rand(shape, generator=g).cuda()
    lab = torch.randint(0, C, (shape[0],) + shape[2:], generator=g).to(torch.int64 if C == 2 else torch.uint8).cuda()
    res = []
    for two in (False, True):
        monkeypatch.setattr(step, "TWO_STREAM", two)
        torch.manual_seed(11)
        if kind == "cnnvit":
            nets = [_make("unet2d", C), _make("swin", C), _make("swin", C)]
            nets[2].load_state_dict(nets[1].state_dict())
            tr = CnnMeetVitTrainer(nets[0], nets[1], nets[2], labeled_bs=1, num_classes=C, seed=5, iter_num=1500)
        else:
            base = "unet2d" if kind == "uamt2d" else "unet3d"
            nets = [_make(base, C), _make(base, C)]
            nets[1].load_state_dict(nets[0].state_dict())
            tr = UAMTTrainer(nets[0], nets[1], labeled_bs=shape[0] // 2, num_classes=C, seed=5, max_iterations=2000,
                             iter_num=1500)
        for n in nets:
            n.train()
        for _ in range(2):
            tr.step(vol, lab)
        torch.cuda.synchronize()
        res.append(([n.flat_param.clone() for n in nets],
                    [b.clone() for n in nets for b in n.buffers() if b.is_floating_point()], tr.losses()))
    (p0, b0, l0), (p1, b1, l1) = res
    assert all(torch.equal(x, y) for x, y in zip(p0, p1))
    assert all(torch.equal(x, y) for x, y in zip(b0, b1))
    assert l0 == l1 and all(np.isfinite(v) for v in l0.values())


@pytest.mark.parametrize("which", ["split", "encoder"])
def test_swinunet_load_from_matches_the_reference_class(which, tmp_path):
    """SwinUnet.load_from (reference vision_transformer.py:54-89; called unconditionally by train_mean_teacher_ViT.py:
    147-156) on both checkpoint formats: a wrapped whole-network checkpoint (17-character prefix stripped, head dropped)
    and an ImageNet encoder checkpoint {"model": ...} (encoder stages mirrored into the decoder, shape mismatches and
    foreign keys dropped).  Golden: which entry ends with which tensor, from the REAL class (oracle/gen_golden.py)."""
    import json
    from types import SimpleNamespace as NS
    from config import lite_config
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.gen_golden import load_from_checkpoints
    z = np.load(os.path.join(ROOT, "tests", "golden", "swin_load_from.npz"), allow_pickle=False)
    ks = [(k, tuple(sh), dt) for k, sh, dt in json.loads(str(z["meta"]))["key_shapes"]]
    model = SwinUnet(lite_config(), img_size=224, num_classes=4)
    names = [str(n) for n in z[which + "_names"]]
    assert list(model.state_dict().keys()) == names
    model.load_state_dict(filler.fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}))
    before = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    path = str(tmp_path / "ck.pth")
    torch.save(load_from_checkpoints(ks, which), path)
    model.load_from(NS(MODEL=NS(PRETRAIN_CKPT=path)))
    after = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    n_changed = 0
    for i, k in enumerate(names):
        changed = bool((after[k] != before[k]).any())
        assert changed == bool(z[which + "_changed"][i]), (k, changed)
        n_changed += changed
        t = after[k].double()
        assert abs(float(t.sum()) - float(z[which + "_sum"][i])) <= 1e-9 * max(1.0, float(z[which + "_abssum"][i])), k
        assert abs(float(t.abs().sum()) - float(z[which + "_abssum"][i])) <= 1e-9 * max(1.0, float(z[which + "_abssum"][i])), k
        assert float(t.flatten()[0]) == float(z[which + "_first"][i]) and float(t.flatten()[-1]) == float(z[which + "_last"][i]), k
    assert n_changed == (237 if which == "split" else 215)
    # the flat parameter buffer (what the kernels read) saw the load: the views still alias it
    p = dict(model.named_parameters())["swin_unet.layers_up.1.blocks.0.attn.qkv.weight"]
    assert p.data_ptr() >= model.flat_param.data_ptr() and \
        p.data_ptr() < model.flat_param.data_ptr() + model.flat_param.numel() * 4
    # no checkpoint: the reference prints "none pretrain" and leaves the weights alone
    model.load_from(NS(MODEL=NS(PRETRAIN_CKPT=None)))

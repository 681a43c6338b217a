"""UA-MT (SURVEY s.8 row n1): the HIP path through UAMTTrainer against the golden vectors of the real reference
(oracle/gen_golden.py::run_uamt_case, code/train_uncertainty_aware_mean_teacher_{2D,3D,ViT_2D}.py) and the CPU oracle.
Tolerances as in test_parity_gpu.py: 1e-3 on logits / losses, the measured fp32 envelope on gradients."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL_LOGIT = 1e-3
TOL_LOSS = 1e-3


def _sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, 64).astype(np.int64))


@pytest.mark.parametrize("name", ["uamt_unet2d_64", "uamt_unet3d_64", "uamt_swin_224"])
def test_uamt_step_matches_reference_golden_and_oracle(name):
    from oracle import filler
    from oracle.nets import OracleUNet2D, OracleUNet3D
    from oracle.step import uamt_step
    from mis_hip.step import UAMTTrainer

    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    kind, cfg, it = meta["kind"], meta["cfg"], meta["iters"][0]
    C, L, B = cfg["num_classes"], cfg["labeled_bs"], cfg["batch_size"]
    U, sp = B - L, tuple(cfg["spatial"])
    if kind == "swin":       # train_uncertainty_aware_mean_teacher_ViT_2D.py: the 2-D loop on two SwinUnets
        from networks.net_factory import net_factory
        from oracle.swin import OracleSwinUnet
        onet, make, head = OracleSwinUnet(C), (lambda: net_factory("ViT_Seg", 1, C)), "swin_unet.output.weight"
        ldt = torch.uint8
    elif kind == "unet2d":
        from networks.net_factory import net_factory
        onet, make, head = OracleUNet2D(1, C), (lambda: net_factory("unet", 1, C)), "decoder.out_conv.weight"
        ldt = torch.uint8
    else:
        from networks.net_factory_3d import net_factory_3d
        onet, make, head = OracleUNet3D(C, 1), (lambda: net_factory_3d("unet_3D", 1, C)), "final.weight"
        ldt = torch.int64
    sd0 = filler.fill_state_dict(onet.new_state())
    tsd0 = filler.fill_state_dict({"t." + k: v.clone() for k, v in onet.new_state().items()})
    tsd0 = {k[2:]: v for k, v in tsd0.items()}
    tsd0[head] = tsd0[head] * cfg["teacher_head_scale"]
    volume = filler.image((B, 1) + sp, "volume")
    label = filler.labels((B,) + sp, C, ldt)
    noise = filler.noise((U, 1) + sp, "noise")
    mc_noise = [filler.noise((2 * U, 1) + sp, f"mc_noise{i}") for i in range(4)]

    model, ema = make(), make()
    for p in ema.parameters():
        p.detach_()
    model.train(); ema.train()
    model.dropout_enabled = ema.dropout_enabled = False
    model.load_state_dict(sd0)
    ema.load_state_dict(tsd0)
    tr = UAMTTrainer(model, ema, labeled_bs=L, num_classes=C, base_lr=cfg["base_lr"],
                     max_iterations=cfg["max_iterations"], ema_decay=cfg["ema_decay"], consistency=cfg["consistency"],
                     consistency_rampup=cfg["rampup"], iter_num=it)
    mom = {}
    for n, v in model.named_flat(tr.momentum_buf):
        m = filler.uniform(v.shape, "mom." + n, -0.01, 0.01)
        v.copy_(m)
        mom[n] = m.clone()
    tr.step(volume.cuda(), label.cuda(), noise=noise.cuda(), mc_noise=[m.cuda() for m in mc_noise])
    got = tr.losses()
    pre = f"it{it}_"
    nvox = U * int(np.prod(sp))

    # ---- (a) golden vectors from the real reference ----
    for k in ("loss", "loss_ce", "loss_dice", "consistency_loss"):
        assert abs(got[k] - float(z[pre + k])) <= TOL_LOSS, (k, got[k], float(z[pre + k]))
    assert abs(got["consistency_weight"] - float(z[pre + "consistency_weight"])) <= 1e-6
    assert abs(got["threshold"] - float(z[pre + "threshold"])) <= 1e-6
    # voxels whose entropy sits within fp32 rounding of the threshold may fall on either side
    assert abs(got["unmasked_voxels"] - float(z[pre + "unmasked"])) <= max(4.0, 2e-4 * nvox)
    s_logits, t_logits = model._last[0].out.t, ema.plan_for(
        (U, 1, 1) + sp if len(sp) == 2 else (U, 1) + sp).out.t
    for t, key in ((s_logits, "logits_"), (t_logits, "teacher_logits_")):
        flat = t.detach().double().cpu().flatten()
        np.testing.assert_allclose(flat[_sample_idx(flat.numel())].numpy(), z[pre + key + "samples"], rtol=0,
                                   atol=TOL_LOGIT * max(1.0, float(np.abs(z[pre + key + "samples"]).max())))
    env = 6.0 * z[pre + "grad_relerr32"] + 2e-3
    gn = np.array([float(g.double().norm()) for _, g in model.named_flat(model.flat_grad)])
    ref_gn, gn64 = z[pre + "grad_norms"], z[pre + "grad_norms64"]
    assert np.all(np.abs(gn - ref_gn) <= env * np.maximum(ref_gn, gn64) + 1e-5 * ref_gn.max()), \
        list(zip(gn, ref_gn, gn64))
    if pre + "teacher_buf_sum" in z.files:      # BatchNorm running statistics after 1 (student) / 5 (teacher) forwards
        msd, esd = model.state_dict(), ema.state_dict()
        bufs = [n for n in msd if n.endswith("running_mean") or n.endswith("running_var")]
        np.testing.assert_allclose(np.array([float(msd[n].double().sum()) for n in bufs]), z[pre + "student_buf_sum"],
                                   rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(np.array([float(esd[n].double().sum()) for n in bufs]), z[pre + "teacher_buf_sum"],
                                   rtol=1e-4, atol=1e-3)
        nbt = [n for n in esd if n.endswith("num_batches_tracked")]
        assert all(int(esd[n]) == 5 for n in nbt) and all(int(msd[n]) == 1 for n in nbt)

    # ---- (b) the CPU oracle run here ----
    student = {k: v.clone() for k, v in sd0.items()}
    teacher = {k: v.clone() for k, v in tsd0.items()}
    orc = uamt_step(onet, student, teacher, mom, volume, label, noise, mc_noise, it, labeled_bs=L, num_classes=C,
                    base_lr=cfg["base_lr"], max_iterations=cfg["max_iterations"], ema_decay=cfg["ema_decay"],
                    consistency=cfg["consistency"], rampup=cfg["rampup"], drop_student="off", drop_teacher="off")
    for k in ("loss", "loss_ce", "loss_dice", "consistency_loss"):
        assert abs(got[k] - orc[k]) <= TOL_LOSS
    assert (s_logits.cpu().reshape(orc["logits"].shape) - orc["logits"]).abs().max().item() <= TOL_LOGIT
    tscale = max(1.0, float(orc["teacher_logits"].abs().max()))
    assert (t_logits.cpu().reshape(orc["teacher_logits"].shape) - orc["teacher_logits"]).abs().max().item() \
        <= TOL_LOGIT * tscale
    # the MC-dropout mean prediction / entropy map itself
    mp = tr._mean_probs.cpu().reshape((U, C) + sp)
    unc = -1.0 * torch.sum(mp * torch.log(mp + 1e-6), dim=1, keepdim=True)
    assert (unc - orc["uncertainty"]).abs().max().item() <= 1e-3      # teacher logits are O(40) in this fixture
    gscale = max(float(g.abs().max()) for g in orc["grads"].values())
    gmax = z[pre + "grad_max64"]
    for i, (n, g) in enumerate(model.named_flat(model.flat_grad)):
        tol_g = env[i] * max(float(orc["grads"][n].abs().max()), gmax[i]) + 5e-4 * gscale
        err = (g.cpu() - orc["grads"][n]).abs().max().item()
        assert err <= tol_g, (n, err, tol_g)
    lr = float(z[pre + "lr"])
    for i, (n, v) in enumerate(model.named_flat(model.flat_param)):
        tol_g = env[i] * gmax[i] + 1e-5 * gscale
        assert (v.cpu() - student[n]).abs().max().item() <= 1e-6 + lr * tol_g, n


def test_uamt_philox_mode_runs_and_is_reproducible():
    """Reference-faithful mode: Philox dropout in all 5 teacher passes and device noise; finite and seed-reproducible."""
    from networks.net_factory import net_factory
    from mis_hip.step import UAMTTrainer
    from oracle import filler
    from oracle.nets import OracleUNet2D
    sd0 = filler.fill_state_dict(OracleUNet2D(1, 4).new_state())
    vol = filler.image((4, 1, 32, 32), "volume").cuda()
    lab = filler.labels((4, 32, 32), 4, torch.uint8).cuda()
    res = []
    for _ in range(2):
        m, e = net_factory("unet", 1, 4), net_factory("unet", 1, 4)
        m.load_state_dict(sd0); e.load_state_dict(sd0)
        tr = UAMTTrainer(m, e, labeled_bs=2, num_classes=4, seed=7, max_iterations=100, iter_num=90)
        for _ in range(2):
            tr.step(vol, lab)
        res.append((tr.losses(), m.flat_param.clone(), tr._mean_probs.clone()))
    assert all(np.isfinite(v) for v in res[0][0].values())
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    # mean of 8 softmaxes: a probability vector per voxel
    assert (res[0][2].sum(dim=1) - 1.0).abs().max().item() <= 1e-5

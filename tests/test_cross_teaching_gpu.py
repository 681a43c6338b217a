"""Cross teaching UNet <-> SwinUnet (BASELINE config 5 geometry, 224x224) on HIP vs the golden vector of the
real reference and the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, 64).astype(np.int64))


@pytest.mark.parametrize("gold", ["cross_224", "cps_vit_224"])
def test_cross_teaching_step_matches_reference_and_oracle(gold):
    """cross_224: UNet <-> SwinUnet (train_cross_teaching_between_cnn_transformer_2D.py); cps_vit_224: two SwinUnet
    students (train_cross_pseudo_supervision_2D_ViT.py:213-241, the same loop body)."""
    from config import lite_config
    from mis_hip.step import CrossTeachingTrainer
    from networks.net_factory import net_factory
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.nets import OracleUNet2D
    from oracle.step import cross_teaching_step
    from oracle.swin import OracleSwinUnet

    z = np.load(os.path.join(GOLD, gold + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg, it = meta["cfg"], meta["iters"][0]
    kinds = meta.get("kinds", ["unet2d", "swin"])
    C, L = cfg["num_classes"], cfg["labeled_bs"]
    nets = [OracleUNet2D(1, C) if k == "unet2d" else OracleSwinUnet(C) for k in kinds]
    sds = []
    for m, onet in enumerate(nets):
        sd = filler.fill_state_dict({f"m{m}." + k: v.clone() for k, v in onet.new_state().items()})
        sds.append({k.split(".", 1)[1]: v for k, v in sd.items()})
    B, sp = cfg["batch_size"], tuple(cfg["spatial"])
    volume = filler.image((B, 1) + sp, "volume")
    label = filler.labels((B,) + sp, C, torch.uint8)
    models = [net_factory("unet", 1, C) if k == "unet2d" else SwinUnet(lite_config(), img_size=224, num_classes=C)
              for k in kinds]
    for m in range(2):
        models[m].load_state_dict(sds[m])
        models[m].train()
        models[m].dropout_enabled = False
    tr = CrossTeachingTrainer(models[0], models[1], labeled_bs=L, num_classes=C, base_lr=cfg["base_lr"],
                              max_iterations=cfg["max_iterations"], consistency=cfg["consistency"],
                              consistency_rampup=cfg["rampup"], iter_num=it)
    moms = []
    for m, buf in enumerate((tr.mom1, tr.mom2)):
        mm = {}
        for n, v in models[m].named_flat(buf):
            t = filler.uniform(v.shape, f"mom{m}." + n, -0.01, 0.01)
            v.copy_(t)
            mm[n] = t.clone()
        moms.append(mm)
    tr.step(volume.cuda(), label.cuda())
    got = tr.losses()
    pre = f"it{it}_"
    # ---- golden (real reference) ----
    assert abs(got["model1_loss"] - float(z[pre + "model1_loss"])) <= 2e-4
    assert abs(got["model2_loss"] - float(z[pre + "model2_loss"])) <= 2e-4
    assert abs(got["pseudo_supervision1"] - float(z[pre + "pseudo1"])) <= 2e-4
    assert abs(got["pseudo_supervision2"] - float(z[pre + "pseudo2"])) <= 2e-4
    assert abs(got["consistency_weight"] - float(z[pre + "consistency_weight"])) <= 1e-6
    from mis_hip import ops
    st = ops.read_step_state(tr.state)
    assert st["iter_num"] == it + 1
    for m in range(2):
        lg = models[m]._last[0].out.t.detach().double().cpu().flatten()
        np.testing.assert_allclose(lg[_sample_idx(lg.numel())].numpy(), z[pre + f"logits{m + 1}_samples"], rtol=0,
                                   atol=1e-3)
        gn = np.array([float(g.double().norm()) for _, g in models[m].named_flat(models[m].flat_grad)])
        ref_gn, gn64 = z[pre + f"grad_norms{m + 1}"], z[pre + f"grad_norms64_{m + 1}"]
        env = 6.0 * z[pre + f"grad_relerr32_{m + 1}"] + 2e-3
        assert np.all(np.abs(gn - ref_gn) <= env * np.maximum(ref_gn, gn64) + 1e-5 * ref_gn.max())
    # ---- oracle, full tensors ----
    osd = [{k: v.clone() for k, v in sd.items()} for sd in sds]
    r = cross_teaching_step(nets[0], nets[1], osd[0], osd[1], moms[0], moms[1], volume, label, it, labeled_bs=L,
                            num_classes=C, base_lr=cfg["base_lr"], max_iterations=cfg["max_iterations"],
                            consistency=cfg["consistency"], rampup=cfg["rampup"], drop1="off", drop2="off")
    lr = r["lr"]
    assert abs(st["lr"] - lr_next(it + 1, cfg)) <= 1e-9 + 1e-6 * lr
    for m in range(2):
        lg = models[m]._last[0].out.t.cpu().reshape(r[f"logits{m + 1}"].shape)
        assert (lg - r[f"logits{m + 1}"]).abs().max().item() <= 1e-3
        env = 6.0 * z[pre + f"grad_relerr32_{m + 1}"] + 2e-3
        gmax = z[pre + f"grad_max64_{m + 1}"]
        gscale = max(float(g.abs().max()) for g in r["grads"][m].values())
        for i, (n, g) in enumerate(models[m].named_flat(models[m].flat_grad)):
            ref = r["grads"][m][n]
            tol = env[i] * max(float(ref.abs().max()), gmax[i]) + 5e-4 * gscale
            assert (g.cpu() - ref).abs().max().item() <= tol, (m, n)
        for i, (n, v) in enumerate(models[m].named_flat(models[m].flat_param)):
            tol = env[i] * gmax[i] + 1e-5 * gscale
            assert (v.cpu() - osd[m][n]).abs().max().item() <= 1e-6 + lr * tol, (m, n)


def lr_next(k, cfg):
    """learning rate in effect for step k under the post-increment rule"""
    return cfg["base_lr"] * (1.0 - k / cfg["max_iterations"]) ** 0.9

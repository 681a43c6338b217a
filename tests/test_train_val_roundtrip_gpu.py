"""train -> (in-training validation, best-model checkpoint) -> inference CLI round trip on small .npz datasets.

Reference behaviour under test (code/train_mean_teacher_2D.py:262-294, code/train_mean_teacher_3D.py:201-222): every
200 iterations the student is scored on the validation split in eval mode; a new best mean Dice writes
``iter_{n}_dice_{d}.pth`` and ``{model}_best_model.pth``, which ``test_2D_fully.py`` / ``test_3D.py`` then load."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cv-ssl-mis_amd")


def _blob_case(rng, shape, classes):
    """An image whose label is a simple function of the intensity (learnable within 200 iterations)."""
    img = rng.random(shape).astype(np.float32) * 0.2
    lab = np.zeros(shape, np.uint8)
    for c in range(1, classes):
        lo = [int(rng.integers(0, max(1, s - 10))) for s in shape]
        sl = tuple(slice(l, l + 10) for l in lo) if len(shape) == 2 else (slice(None),) + tuple(
            slice(l, l + 10) for l in lo[1:])
        img[sl] = 0.3 + 0.2 * c + rng.random(img[sl].shape).astype(np.float32) * 0.05
        lab[sl] = c
    return img, lab


def _run(script, args, work):
    r = subprocess.run([sys.executable, os.path.join(PKG, script)] + args, cwd=str(work),
                       env=dict(os.environ, PYTHONPATH=PKG), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_2d_train_validate_best_model_then_test_cli(tmp_path):
    rng = np.random.default_rng(1)
    acdc = tmp_path / "data" / "ACDC"
    (acdc / "data" / "slices").mkdir(parents=True)
    names = [f"patient{i // 8:03d}_frame01_slice_{i % 8}" for i in range(80)]
    for n in names:
        img, lab = _blob_case(rng, (64, 64), 4)
        np.savez(acdc / "data" / "slices" / (n + ".npz"), image=img, label=lab)
    (acdc / "train_slices.list").write_text("\n".join(names) + "\n")
    vols = {"val.list": ["patient100_frame01", "patient101_frame01"], "test.list": ["patient102_frame01"]}
    for lst, cases in vols.items():
        for c in cases:
            sl = [_blob_case(rng, (64, 64), 4) for _ in range(3)]
            np.savez(acdc / "data" / (c + ".npz"), image=np.stack([s[0] for s in sl]),
                     label=np.stack([s[1] for s in sl]))
        (acdc / lst).write_text("\n".join(cases) + "\n")
    work = tmp_path / "code"
    work.mkdir()
    out = _run("train_mean_teacher_2D.py", ["--root_path", str(acdc), "--exp", "t/RT", "--max_iterations", "200",
                                            "--patch_size", "64", "64", "--batch_size", "4", "--labeled_bs", "2",
                                            "--labeled_num", "3", "--base_lr", "0.05"], work)
    assert "Training Finished!" in out and "iteration 200 : mean_dice :" in out
    snap = tmp_path / "model" / "t" / "RT_3_labeled" / "unet"
    best = snap / "unet_best_model.pth"
    assert best.exists(), sorted(p.name for p in snap.iterdir())
    assert any(p.name.startswith("iter_200_dice_") for p in snap.iterdir())   # a real validation score, not the fallback
    sc = (snap / "scalars.csv").read_text()
    for tag in ("info/lr", "info/total_loss", "info/loss_ce", "info/loss_dice", "info/consistency_loss",
                "info/consistency_weight", "info/val_mean_dice", "info/val_mean_hd95", "info/val_1_dice"):
        assert "," + tag + "," in sc, tag
    out = _run("test_2D_fully.py", ["--root_path", str(acdc), "--exp", "t/RT", "--labeled_num", "3"], work)
    assert "init weight from" in out and "unet_best_model.pth" in out


def test_3d_train_validate_best_model_then_test_cli(tmp_path):
    rng = np.random.default_rng(2)
    brats = tmp_path / "data" / "BraTS2019"
    (brats / "data").mkdir(parents=True)
    lists = {"train.txt": [f"BraTS19_{i}" for i in range(6)], "val.txt": ["BraTS19_v0", "BraTS19_v1"],
             "test.txt": ["BraTS19_t0"]}
    for lst, cases in lists.items():
        for c in cases:
            img = rng.random((40, 40, 40)).astype(np.float32) * 0.2
            lab = np.zeros((40, 40, 40), np.uint8)
            lo = [int(v) for v in rng.integers(4, 20, 3)]
            sl = tuple(slice(l, l + 14) for l in lo)
            img[sl] = 0.8 + rng.random(img[sl].shape).astype(np.float32) * 0.1
            lab[sl] = 1
            np.savez(brats / "data" / (c + ".npz"), image=img, label=lab)
        (brats / lst).write_text("\n".join(cases) + "\n")
    work = tmp_path / "code"
    work.mkdir()
    out = _run("train_mean_teacher_3D.py", ["--root_path", str(brats), "--exp", "t/RT3", "--max_iterations", "200",
                                            "--patch_size", "32", "32", "32", "--batch_size", "2", "--labeled_bs", "1",
                                            "--labeled_num", "2", "--base_lr", "0.05"], work)
    assert "Training Finished!" in out and "iteration 200 : dice_score :" in out
    snap = tmp_path / "model" / "t" / "RT3_2" / "unet_3D"          # train_mean_teacher_3D.py:252: no "_labeled"
    assert (snap / "unet_3D_best_model.pth").exists(), sorted(p.name for p in snap.iterdir())
    # test_3D.py:21 reads ../model/<exp>/<model>: the reference's convention is --exp <exp>_<labeled_num>
    out = _run("test_3D.py", ["--root_path", str(brats), "--exp", "t/RT3_2", "--model", "unet_3D"], work)
    assert "init weight from" in out and "Testing end" in out

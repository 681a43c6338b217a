"""Host logic of the input pipeline (SURVEY s.8 rows a13 / n4) on CPU: sampler contract, random-draw order against
the oracle, parameter-record layout, case reading."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))


def test_two_stream_sampler_contract():
    from dataloaders.dataset import TwoStreamBatchSampler
    lab, unl = list(range(10)), list(range(10, 33))
    s = TwoStreamBatchSampler(lab, unl, 7, 4)
    assert len(s) == 3 and s.primary_batch_size == 3 and s.secondary_batch_size == 4
    np.random.seed(0)
    batches = list(s)
    assert len(batches) == 3                                    # ragged labeled tail (1 index) dropped
    seen_lab, seen_unl = [], []
    for b in batches:
        assert len(b) == 7
        assert all(i in lab for i in b[:3]) and all(i in unl for i in b[3:])   # labeled FIRST
        seen_lab += b[:3]
        seen_unl += b[3:]
    assert len(set(seen_lab)) == 9                              # one pass over the labeled indices, no repeats
    assert len(set(seen_unl)) == 12                             # 12 < 23: still inside the first unlabeled shuffle
    # random-number consumption: one labeled permutation at the start, unlabeled permutations only on demand
    np.random.seed(5)
    expect_lab = np.random.permutation(lab)
    expect_unl = np.random.permutation(unl)
    np.random.seed(5)
    first = next(iter(s))
    assert list(first[:3]) == list(expect_lab[:3]) and list(first[3:]) == list(expect_unl[:4])
    # the unlabeled stream outlives the epoch boundary inside one iterator only (a new epoch reshuffles both)
    s2 = TwoStreamBatchSampler(list(range(8)), [100, 101, 102], 4, 2)
    np.random.seed(1)
    b2 = list(s2)
    assert len(b2) == 4 and all(set(b[2:]) <= {100, 101, 102} for b in b2)
    with pytest.raises(AssertionError):
        TwoStreamBatchSampler([0], [1, 2], 4, 2)


def test_random_draw_order_matches_oracle():
    from dataloaders.brats2019 import RandomRotFlipCrop
    from dataloaders.dataset import RandomGenerator
    from oracle.augment import random_generator, rot_flip_crop
    img = np.arange(30 * 22, dtype=np.float32).reshape(30, 22)
    lab = (np.arange(30 * 22) % 4).astype(np.uint8).reshape(30, 22)
    gen = RandomGenerator((16, 16))
    random.seed(3), np.random.seed(4)
    mine = [gen.draw() for _ in range(40)]
    random.seed(3), np.random.seed(4)
    theirs = [random_generator(img, lab, (16, 16))[2] for _ in range(40)]
    assert mine == theirs and {m[0] for m in mine} == {0, 1, 2}
    vol = np.zeros((20, 14, 9), np.float32)
    g3 = RandomRotFlipCrop((12, 12, 12))
    np.random.seed(9)
    d = [g3.draw(vol.shape) for _ in range(10)]
    after = np.random.randint(1 << 30)
    np.random.seed(9)
    for _ in range(10):
        i, l = rot_flip_crop(vol, vol.astype(np.uint8), (12, 12, 12))
        assert i.shape == (1, 12, 12, 12) and l.dtype == np.int64
    assert after == np.random.randint(1 << 30)
    assert all(len(o) == 3 for _, _, o in d)


def test_record_layouts_and_rotate_params():
    from dataloaders.brats2019 import CROP3D_DTYPE
    from dataloaders.dataset import AUG2D_DTYPE, rotate_params
    from mis_hip import lib
    assert AUG2D_DTYPE.itemsize == lib.AUG2D_BYTES == 88 and CROP3D_DTYPE.itemsize == lib.CROP3D_BYTES == 48
    assert AUG2D_DTYPE.fields["m00"][1] == 40 and AUG2D_DTYPE.fields["off1"][1] == 80
    assert CROP3D_DTYPE.fields["d0"][1] == 16 and CROP3D_DTYPE.fields["o2"][1] == 44
    m, off = rotate_params(0, (9, 5))
    assert np.array_equal(m, np.eye(2)) and np.array_equal(off, np.zeros(2))
    m, off = rotate_params(-20, (64, 48))
    assert m[0, 0] == m[1, 1] and m[0, 1] == -m[1, 0] and abs(m[0, 0] ** 2 + m[0, 1] ** 2 - 1) < 1e-15


def test_base_datasets_read_npz_cases(tmp_path):
    from dataloaders.brats2019 import BraTS2019
    from dataloaders.dataset import BaseDataSets
    root = tmp_path / "ACDC"
    (root / "data" / "slices").mkdir(parents=True)
    names = [f"patient001_frame01_slice_{i}" for i in range(3)]
    for i, n in enumerate(names):
        np.savez(root / "data" / "slices" / (n + ".npz"), image=np.full((6, 5), i, np.float32),
                 label=np.full((6, 5), i % 4, np.uint8))
    (root / "train_slices.list").write_text("\n".join(names) + "\n")
    np.savez(root / "data" / "patient002_frame01.npz", image=np.zeros((3, 6, 5), np.float32),
             label=np.zeros((3, 6, 5), np.uint8))
    (root / "val.list").write_text("patient002_frame01\n")
    tr = BaseDataSets(base_dir=str(root), split="train", num=2)
    assert len(tr) == 2 and tr[1]["image"][0, 0] == 1 and tr[1]["idx"] == 1
    va = BaseDataSets(base_dir=str(root), split="val")
    assert len(va) == 1 and va[0]["image"].shape == (3, 6, 5)
    b = tmp_path / "BraTS"
    (b / "data").mkdir(parents=True)
    np.savez(b / "data" / "case_a.npz", image=np.ones((4, 5, 6), np.float32), label=np.ones((4, 5, 6), np.int16))
    (b / "train.txt").write_text("case_a,extra\n")
    ds = BraTS2019(base_dir=str(b), split="train")
    assert len(ds) == 1 and ds[0]["label"].dtype == np.uint8 and ds[0]["image"].shape == (4, 5, 6)


GOLD = os.path.join(ROOT, "tests", "golden")


def test_two_stream_sampler_matches_the_reference_class():
    """sampler.npz: index streams of the REAL TwoStreamBatchSampler (code/dataloaders/dataset.py:247-294 and its
    brats2019.py twin) over several epochs, and where they leave np.random (oracle/gen_golden_io.py)."""
    from dataloaders import brats2019, dataset
    g = np.load(os.path.join(GOLD, "sampler.npz"))
    for c in range(int(g["cases"])):
        primary, secondary, batch, sbs, seed, epochs = (int(v) for v in g[f"cfg{c}"])
        prim, sec = list(range(primary)), list(range(primary, primary + secondary))
        for mod in (dataset, brats2019):
            s = mod.TwoStreamBatchSampler(prim, sec, batch, sbs)
            np.random.seed(seed)
            got = np.stack([np.array(b) for _ in range(epochs) for b in s])
            assert len(s) == int(g[f"len{c}"])
            assert np.array_equal(got, g[f"batches{c}"])
            assert int(np.random.randint(1 << 30)) == int(g[f"tail_np{c}"])


def test_oracle_augmentations_match_the_reference_goldens():
    """oracle/augment.py against the pixels of the REAL RandomGenerator / RandomRotFlip + RandomCrop + ToTensor
    (aug2d.npz, aug3d.npz): the restatement the GPU tests sweep with is pinned to the reference itself."""
    sys.path.insert(0, ROOT)
    from oracle.augment import random_generator, rot_flip_crop
    from oracle.gen_golden_io import AUG2D, aug2d_slices, aug3d_volumes
    g = np.load(os.path.join(GOLD, "aug2d.npz"))
    slices = aug2d_slices()
    for c, size in enumerate(AUG2D["out"]):
        random.seed(500 + c), np.random.seed(600 + c)
        for n, (img, lab) in enumerate(slices):
            oi, ol, draws = random_generator(img, lab, size)
            assert np.array_equal(oi, g[f"image{c}"][n]) and np.array_equal(ol, g[f"label{c}"][n])
            assert draws[0] == int(g[f"modes{c}"][n])
        assert (random.random(), int(np.random.randint(1 << 30))) == (float(g[f"tail_random{c}"]), int(g[f"tail_np{c}"]))
    g3 = np.load(os.path.join(GOLD, "aug3d.npz"))
    vols, idx = aug3d_volumes()
    np.random.seed(700)
    for n, i in enumerate(idx):
        oi, ol = rot_flip_crop(vols[i][0], vols[i][1], tuple(int(v) for v in g3["patch"]))
        assert np.array_equal(oi, g3["image"][n]) and np.array_equal(ol.astype(np.uint8), g3["label"][n])
    assert int(np.random.randint(1 << 30)) == int(g3["tail_np"])

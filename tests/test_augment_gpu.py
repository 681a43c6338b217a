"""Device input pipeline (SURVEY s.8 row n4) vs the numpy/scipy oracle: bit-exact pixels and identical consumption of
the host random generators."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _slices(rng, n, lo=5, hi=90):
    out = []
    for _ in range(n):
        H, W = (int(v) for v in rng.integers(lo, hi, 2))
        out.append((rng.random((H, W)).astype(np.float32), rng.integers(0, 4, (H, W)).astype(np.uint8)))
    return out


@pytest.mark.parametrize("out_size,seed", [((64, 64), 0), ((37, 53), 1), ((224, 224), 2), ((256, 256), 3)])
def test_random_generator_batch_is_bit_exact(out_size, seed):
    from dataloaders.dataset import DeviceSlicePool, RandomGenerator, augment_batch
    from oracle.augment import random_generator
    rng = np.random.default_rng(seed)
    slices = _slices(rng, 48) + [(rng.random((256, 216)).astype(np.float32), rng.integers(0, 4, (256, 216)).astype(np.uint8)),
                                 (rng.random((208, 256)).astype(np.float32), rng.integers(0, 4, (208, 256)).astype(np.uint8))]
    pool = DeviceSlicePool(slices)
    idx = list(rng.permutation(len(slices)))
    gen = RandomGenerator(out_size)
    random.seed(100 + seed)
    np.random.seed(200 + seed)
    image, label = augment_batch(pool, idx, gen)
    tail = (random.random(), int(np.random.randint(1 << 30)))
    random.seed(100 + seed)
    np.random.seed(200 + seed)
    modes = set()
    for b, i in enumerate(idx):
        ri, rl, draws = random_generator(slices[i][0], slices[i][1], out_size)
        modes.add(draws[0])
        assert np.array_equal(image[b].cpu().numpy(), ri), (b, i, draws, slices[i][0].shape)
        assert np.array_equal(label[b].cpu().numpy(), rl), (b, i, draws)
    assert tail == (random.random(), int(np.random.randint(1 << 30)))     # same number of draws consumed
    assert modes == {0, 1, 2}
    assert image.dtype == torch.float32 and label.dtype == torch.uint8 and image.shape == (len(idx), 1) + out_size


def test_every_rotation_angle_and_rot_flip_case():
    """All 40 angles and all 8 (k, axis) pairs, on odd/even and non-square slices."""
    from dataloaders.dataset import AUG2D_DTYPE, DeviceSlicePool, rotate_params
    from mis_hip import lib as _l
    from scipy import ndimage
    from scipy.ndimage import zoom
    rng = np.random.default_rng(7)
    L = _l.load()
    for (H, W), (oh, ow) in (((31, 44), (40, 40)), ((64, 64), (64, 64)), ((57, 33), (24, 71))):
        img = rng.random((H, W)).astype(np.float32)
        lab = rng.integers(0, 4, (H, W)).astype(np.uint8)
        pool = DeviceSlicePool([(img, lab)])
        cases = [(2, 0, 0, a) for a in range(-20, 20)] + [(1, k, ax, 0) for k in range(4) for ax in range(2)]
        recs = np.zeros(len(cases), AUG2D_DTYPE)
        for r, (mode, k, ax, a) in zip(recs, cases):
            r["H"], r["W"], r["mode"], r["k"], r["axis"] = H, W, mode, k, ax
            if mode == 2:
                m, off = rotate_params(a, (H, W))
                r["m00"], r["m01"], r["m10"], r["m11"], r["off0"], r["off1"] = m[0, 0], m[0, 1], m[1, 0], m[1, 1], off[0], off[1]
        dev = torch.from_numpy(recs.view(np.uint8)).cuda()
        out_i = torch.empty((len(cases), 1, oh, ow), device="cuda")
        out_l = torch.empty((len(cases), oh, ow), dtype=torch.uint8, device="cuda")
        _l.check(L.mis_augment2d(_l.ptr(pool.img), _l.ptr(pool.lab), _l.ptr(dev), len(cases), oh, ow, _l.ptr(out_i),
                                 _l.ptr(out_l), _l.stream_ptr()), "mis_augment2d")
        for b, (mode, k, ax, a) in enumerate(cases):
            if mode == 2:
                ti, tl = ndimage.rotate(img, a, order=0, reshape=False), ndimage.rotate(lab, a, order=0, reshape=False)
            else:
                ti, tl = np.flip(np.rot90(img, k), axis=ax), np.flip(np.rot90(lab, k), axis=ax)
            x, y = ti.shape
            assert np.array_equal(out_i[b, 0].cpu().numpy(), zoom(ti, (oh / x, ow / y), order=0)), (H, W, mode, k, ax, a)
            assert np.array_equal(out_l[b].cpu().numpy(), zoom(tl, (oh / x, ow / y), order=0)), (H, W, mode, k, ax, a)


@pytest.mark.parametrize("patch,seed", [((16, 16, 16), 0), ((24, 20, 12), 1), ((32, 32, 32), 2)])
def test_rot_flip_crop_3d_is_bit_exact(patch, seed):
    """Includes volumes smaller than the patch on one or more axes (the reference zero-pads, brats2019.py:99-108)."""
    from dataloaders.brats2019 import DeviceVolumePool, RandomRotFlipCrop, crop_batch
    from oracle.augment import rot_flip_crop
    rng = np.random.default_rng(seed)
    shapes = [(40, 36, 30), (33, 47, 25), (20, 50, 40), (patch[0], patch[1] + 9, patch[2] + 4), (18, 17, 10),
              (patch[1] + 5, patch[0] - 3, patch[2] + 1)]
    vols = [(rng.random(s).astype(np.float32), rng.integers(0, 2, s).astype(np.uint8)) for s in shapes]
    pool = DeviceVolumePool(vols)
    idx = list(rng.integers(0, len(vols), 24))
    gen = RandomRotFlipCrop(patch)
    np.random.seed(300 + seed)
    image, label = crop_batch(pool, idx, gen)
    tail = int(np.random.randint(1 << 30))
    np.random.seed(300 + seed)
    for b, i in enumerate(idx):
        ri, rl = rot_flip_crop(vols[i][0], vols[i][1], patch)
        assert np.array_equal(image[b].cpu().numpy(), ri), (b, i, shapes[i])
        assert np.array_equal(label[b].cpu().numpy(), rl), (b, i, shapes[i])
    assert tail == int(np.random.randint(1 << 30))
    assert label.dtype == torch.int64 and image.shape == (len(idx), 1) + patch
    # uint8 labels on request
    np.random.seed(300 + seed)
    _, lab8 = crop_batch(pool, idx, gen, label_dtype=torch.uint8)
    assert lab8.dtype == torch.uint8 and torch.equal(lab8.long(), label)


def test_two_stream_loader_feeds_a_training_step():
    """pool -> sampler -> one gather launch per batch -> Mean-Teacher step (labeled samples first)."""
    from dataloaders.dataset import DeviceSlicePool, DeviceTwoStreamLoader, RandomGenerator, TwoStreamBatchSampler
    from mis_hip.step import MeanTeacherTrainer
    from networks.net_factory import net_factory
    rng = np.random.default_rng(5)
    pool = DeviceSlicePool(_slices(rng, 20, 40, 90))
    sampler = TwoStreamBatchSampler(list(range(6)), list(range(6, 20)), 4, 2)
    loader = DeviceTwoStreamLoader(pool, sampler, RandomGenerator((64, 64)))
    assert len(loader) == 3
    model, ema = net_factory("unet", 1, 4), net_factory("unet", 1, 4)
    model.train(), ema.train()
    tr = MeanTeacherTrainer(model, ema, labeled_bs=2, num_classes=4, cons_start_iter=0)
    n = 0
    random.seed(1), np.random.seed(1)
    for _ in range(3):                       # more epochs than ring slots x batches: buffers are reused safely
        for batch in loader:
            assert batch["image"].shape == (4, 1, 64, 64) and batch["label"].shape == (4, 64, 64)
            tr.step(batch["image"], batch["label"])
            n += 1
    assert n == 9
    losses = tr.losses()
    assert all(np.isfinite(v) for v in losses.values())


import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_random_generator_matches_the_reference_class():
    """aug2d.npz: pixels produced by the REAL dataloaders.dataset.RandomGenerator (code/dataloaders/dataset.py:406-425) on
    seeded slices (oracle/gen_golden_io.py); the device pipeline must give the same bytes and leave both host generators
    where the reference leaves them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle"))
    from dataloaders.dataset import DeviceSlicePool, RandomGenerator, augment_batch
    from oracle.gen_golden_io import AUG2D, aug2d_slices
    g = np.load(os.path.join(GOLD, "aug2d.npz"))
    slices = aug2d_slices()
    assert abs(sum(float(s[0].astype(np.float64).sum()) for s in slices) - float(g["input_sum"])) < 1e-9
    pool = DeviceSlicePool(slices)
    for c, size in enumerate(AUG2D["out"]):
        random.seed(500 + c), np.random.seed(600 + c)
        image, label = augment_batch(pool, list(range(len(slices))), RandomGenerator(size))
        tail = (random.random(), int(np.random.randint(1 << 30)))
        assert np.array_equal(image.cpu().numpy(), g[f"image{c}"])
        assert np.array_equal(label.cpu().numpy(), g[f"label{c}"])
        assert tail == (float(g[f"tail_random{c}"]), int(g[f"tail_np{c}"]))


def test_rot_flip_crop_matches_the_reference_transforms():
    """aug3d.npz: the REAL RandomRotFlip -> RandomCrop -> ToTensor chain of code/dataloaders/brats2019.py (:134-147, :84-131,
    :196-208; order of train_mean_teacher_3D.py:102-106), including volumes smaller than the patch."""
    from dataloaders.brats2019 import DeviceVolumePool, RandomRotFlipCrop, crop_batch
    from oracle.gen_golden_io import aug3d_volumes
    g = np.load(os.path.join(GOLD, "aug3d.npz"))
    vols, idx = aug3d_volumes()
    assert idx == [int(i) for i in g["idx"]]
    pool = DeviceVolumePool(vols)
    np.random.seed(700)
    image, label = crop_batch(pool, idx, RandomRotFlipCrop(tuple(int(v) for v in g["patch"])))
    assert int(np.random.randint(1 << 30)) == int(g["tail_np"])
    assert np.array_equal(image.cpu().numpy(), g["image"])
    assert np.array_equal(label.cpu().numpy().astype(np.uint8), g["label"]) and label.dtype == torch.int64

"""Generate the validation / input-pipeline goldens from the REAL reference (SURVEY s.8 rows n3, n4a, a13).

Run in the build container only (it needs /root/reference, which never travels):

    python oracle/gen_golden_io.py

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference modules below import packages this image lacks at MODULE
level (``medpy``, ``h5py``, ``nibabel``, ``SimpleITK``, ``cv2``, ``torchvision``); none of them is touched by the
functions that are run here, so empty stand-in modules are registered in ``sys.modules`` before the import -- exactly
what ``gen_golden.py`` does for ``timm``.  ``medpy.metric.binary.dc`` / ``hd95`` ARE called by ``test_single_volume``:
the stand-ins record their (prediction == i, label == i) arguments, which is how the label map the reference built
leaves the function (it returns only the metrics).  ``Tensor.cuda`` is the identity in this GPU-less container.

What is run, on closed-form / seeded inputs, and stored as data (arrays, index streams, RNG tails -- no source text):

  val2d.npz    val_2D.test_single_volume        (code/val_2D.py:18-39)   around the real networks.unet.UNet
  val3d.npz    val_3D.test_single_case          (code/val_3D.py:14-79)   around the real networks.unet_3D.unet_3D
  cnnvit_infer.npz  test_CNNVIT.test_single_volume (code/test_CNNVIT.py:43-79) around the real SwinUnet and the real UNet
  aug2d.npz    dataloaders.dataset.RandomGenerator.__call__   (code/dataloaders/dataset.py:406-425)
  aug3d.npz    dataloaders.brats2019.RandomRotFlip -> RandomCrop -> ToTensor (code/dataloaders/brats2019.py:80-147,196-208)
  sampler.npz  dataloaders.dataset.TwoStreamBatchSampler (+ the brats2019 twin)  (dataset.py:247-294)

Each case also asserts that the oracle restatement (oracle/augment.py, the loop in tests/test_val_gpu.py) reproduces
the reference bit for bit before the file is written.
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import filler  # noqa: E402
from oracle.augment import random_generator, rot_flip_crop  # noqa: E402

RECORDED = []          # (pred mask, gt mask) pairs handed to medpy.metric.binary.dc by the reference


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def dc(pred, gt):
        RECORDED.append((np.array(pred, copy=True), np.array(gt, copy=True)))
        return 0.0

    binary = mod("medpy.metric.binary", dc=dc, hd95=lambda pred, gt: 0.0, asd=lambda pred, gt: 0.0, hd=lambda pred, gt: 0.0)
    metric = mod("medpy.metric", binary=binary)
    mod("medpy", metric=metric)
    for name in ("h5py", "nibabel", "SimpleITK", "cv2"):
        if name not in sys.modules:
            mod(name)
    if "torchvision" not in sys.modules:
        tr = mod("torchvision.transforms")
        mod("torchvision", transforms=tr)
    sys.path.insert(0, REF)


# --------------------------------------------------------------------------------------------------- validation
class Recording(torch.nn.Module):
    """Passes calls through to the reference network and keeps the logits of every call."""

    def __init__(self, net):
        super().__init__()
        self.net, self.logits = net, []

    def forward(self, x):
        y = self.net(x)
        self.logits.append(y.detach().clone())
        return y


VAL2D = dict(shape=(1, 3, 40, 36), classes=4, patch=(64, 64), image="valimg")
VAL3D = dict(shapes=((40, 36, 48), (28, 40, 32)), patch=(32, 32, 32), stride=16, image="valvol")


def gen_val2d():
    import val_2D
    from networks.unet import UNet
    from scipy.ndimage import zoom
    C, patch = VAL2D["classes"], list(VAL2D["patch"])
    net = UNet(in_chns=1, class_num=C)
    sd = filler.fill_state_dict(net.state_dict())
    sd["decoder.out_conv.weight"] = sd["decoder.out_conv.weight"] * 400.0    # confident predictions
    sd["decoder.out_conv.bias"] = torch.zeros(C)
    net.load_state_dict(sd)
    # classifier bias := minus the mean logit per class (3 decimals, stored in the fixture): all classes get predicted
    image = filler.image(VAL2D["shape"], VAL2D["image"])
    net.eval()
    with torch.no_grad():
        mean = torch.stack([net(torch.from_numpy(zoom(s.numpy(), (patch[0] / s.shape[0], patch[1] / s.shape[1]), order=0)
                                                 )[None, None].float())[0].mean((1, 2)) for s in image[0]]).mean(0)
    out_bias = torch.round(-mean * 1000) / 1000
    sd["decoder.out_conv.bias"] = out_bias.clone()
    net.load_state_dict(sd)
    rec = Recording(net)
    rec.train()
    label = filler.labels(VAL2D["shape"], C, torch.uint8)
    del RECORDED[:]
    metrics = val_2D.test_single_volume(image, label, rec, C, patch_size=patch)
    assert len(metrics) == C - 1 and len(RECORDED) <= C - 1
    # the label map: class i's mask is argument 0 of the i-th metric call (a class that was never predicted makes no call)
    Z, X, Y = VAL2D["shape"][1:]
    pred = np.zeros((Z, X, Y), np.uint8)
    lab = label[0].numpy()
    for i in range(1, C):
        # calculate_metric_percase hands (prediction == i, label == i) to dc() when class i was predicted at all
        for p, g in RECORDED:
            if np.array_equal(g.astype(bool), lab == i):
                assert not pred[p.astype(bool)].any()
                pred[p.astype(bool)] = i
    # margins of the per-pixel decision (top-1 minus top-2 probability), resized back like the prediction
    ties = np.zeros((Z, X, Y), bool)
    assert len(rec.logits) == Z
    for z, lg in enumerate(rec.logits):
        p = torch.softmax(lg, dim=1)[0]
        top2 = torch.topk(p, 2, dim=0).values
        ties[z] = zoom(((top2[0] - top2[1]) < 1e-3).numpy().astype(np.uint8), (X / patch[0], Y / patch[1]), order=0) > 0
        # the prediction rebuilt from the recorded masks must be the arg-max map the reference resized back
        want = zoom(p.argmax(0).numpy(), (X / patch[0], Y / patch[1]), order=0)
        assert np.array_equal(want.astype(np.uint8), pred[z]), "recorded masks do not rebuild the reference prediction"
    out = dict(prediction=pred, ties=ties, out_bias=out_bias.numpy(), weight_scale=400.0, shape=np.array(VAL2D["shape"]), patch=np.array(patch), classes=C,
               logit_samples=torch.stack(rec.logits)[:, 0].flatten()[::997].numpy(),
               label_sum=int(lab.astype(np.int64).sum()), image_sum=float(image.double().sum()))
    np.savez_compressed(os.path.join(GOLD, "val2d.npz"), **out)
    print("val2d: prediction classes", np.bincount(pred.ravel(), minlength=C), "ties", int(ties.sum()))


def gen_val3d():
    import math
    import val_3D
    from networks.unet_3D import unet_3D
    net = unet_3D(n_classes=2, in_channels=1)
    sd = filler.fill_state_dict(net.state_dict())
    sd["final.weight"] = sd["final.weight"] * 40.0
    net.load_state_dict(sd)
    net.eval()            # the callers (train_mean_teacher_3D.py:203, test_3D_util.py) switch to eval before the call
    out = dict(patch=np.array(VAL3D["patch"]), stride=VAL3D["stride"], weight_scale=40.0)
    ps, st = VAL3D["patch"], VAL3D["stride"]
    for n, shape in enumerate(VAL3D["shapes"]):
        image = filler.image((1, 1) + shape, VAL3D["image"])[0, 0].numpy()
        rec = Recording(net)
        rec.eval()
        label_map = val_3D.test_single_case(rec, image, st, st, ps, num_classes=2)
        assert label_map.shape == shape
        # decision margin of the averaged score map, rebuilt from the recorded logits in the reference's window order
        pads = [max(p - s, 0) for p, s in zip(ps, shape)]
        lp = [p // 2 for p in pads]
        padded = tuple(s + p for s, p in zip(shape, pads))
        sx, sy, sz = (math.ceil((padded[i] - ps[i]) / st) + 1 for i in range(3))
        score = np.zeros((2,) + padded, np.float32)
        cnt = np.zeros(padded, np.float32)
        it = iter(rec.logits)
        for x in range(sx):
            xs = min(st * x, padded[0] - ps[0])
            for y in range(sy):
                ys = min(st * y, padded[1] - ps[1])
                for z in range(sz):
                    zs = min(st * z, padded[2] - ps[2])
                    yy = torch.softmax(next(it), dim=1)[0].numpy()
                    score[:, xs:xs + ps[0], ys:ys + ps[1], zs:zs + ps[2]] += yy
                    cnt[xs:xs + ps[0], ys:ys + ps[1], zs:zs + ps[2]] += 1
        assert next(it, None) is None
        score = score / cnt[None]
        crop = tuple(slice(lp[i], lp[i] + shape[i]) for i in range(3))
        assert np.array_equal(np.argmax(score, axis=0)[crop], label_map)
        margin = np.abs(score[1] - score[0])[crop]
        out[f"shape{n}"] = np.array(shape)
        out[f"label_map{n}"] = label_map.astype(np.uint8)
        out[f"ties{n}"] = margin < 1e-3
        out[f"windows{n}"] = len(rec.logits)
        out[f"image_sum{n}"] = float(image.astype(np.float64).sum())
        print(f"val3d {shape}: windows {len(rec.logits)}  foreground {int(label_map.sum())}/{label_map.size}  "
              f"ties {int((margin < 1e-3).sum())}")
    np.savez_compressed(os.path.join(GOLD, "val3d.npz"), **out)


# --------------------------------------------------------------------------------------------------- test_CNNVIT.py
CNNVIT = dict(shape=(3, 50, 44), classes=4, image="cnnvitimg", case="patient101_frame01")


def gen_cnnvit_infer():
    """The label maps the REAL ``test_CNNVIT.test_single_volume`` (code/test_CNNVIT.py:43-79: slice-wise nearest zoom to 224 x
    224, forward, arg-max of the softmax, nearest zoom back) builds around the REAL SwinUnet and the REAL UNet.  Importing the
    script runs its top-level imports: ``h5py`` / ``SimpleITK`` / ``nibabel`` / ``medpy`` are stand-ins (h5py.File serves the
    filler volume, SimpleITK swallows the three NIfTI writes, medpy records the masks), ``config`` (yacs, absent) and
    ``networks.net_factory`` (module-level argparse + ten other backbones) are empty stand-ins that ``test_single_volume``
    never touches."""
    from types import SimpleNamespace as NS
    from scipy.ndimage import zoom
    from oracle.gen_golden import build_reference
    Z, X, Y = CNNVIT["shape"]
    C = CNNVIT["classes"]
    image = filler.image((1,) + CNNVIT["shape"], CNNVIT["image"])[0].numpy()
    label = filler.labels(CNNVIT["shape"], C, torch.uint8).numpy()

    class H5:
        def __init__(self, path, mode):
            assert path.endswith("/data/{}.h5".format(CNNVIT["case"])) and mode == "r", path
            self.d = {"image": image, "label": label}

        def __getitem__(self, k):
            return self.d[k]

    sys.modules["h5py"].File = H5
    itk = sys.modules["SimpleITK"]
    itk.GetImageFromArray = lambda a: NS(SetSpacing=lambda s: None)
    itk.WriteImage = lambda img, path: None
    for name, attrs in (("config", dict(get_config=lambda a: None)),
                        ("networks.net_factory", dict(net_factory=None, config=None, args=None))):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    swin = build_reference("swin", 1, C)          # installs the timm shim, puts REF on sys.path
    import test_CNNVIT as ref
    unet = build_reference("unet2d", 1, C)
    out = dict(shape=np.array(CNNVIT["shape"]), classes=C, image_sum=float(image.astype(np.float64).sum()),
               label_sum=int(label.astype(np.int64).sum()))
    for tag, net, head, scale in (("swin", swin, "swin_unet.output.weight", 60.0), ("unet", unet, "decoder.out_conv.weight", 400.0)):
        sd = filler.fill_state_dict(net.state_dict())
        sd[head] = sd[head] * scale                   # confident predictions
        net.load_state_dict(sd)
        if tag == "unet":
            # classifier bias (3 decimals, stored in the fixture) balanced so that every class wins a share of the pixels
            sd["decoder.out_conv.bias"] = torch.zeros(C)
            net.load_state_dict(sd)
            net.eval()
            with torch.no_grad():
                lg = torch.cat([net(torch.from_numpy(zoom(sl, (224 / X, 224 / Y), order=0))[None, None].float()) for sl in image])
            bias = -lg.mean((0, 2, 3))
            spread = float(lg.std())
            for _ in range(200):
                share = torch.bincount((lg + bias.view(1, C, 1, 1)).argmax(1).flatten(), minlength=C).float() / lg[:, 0].numel()
                bias = bias + 0.5 * spread * (1.0 / C - share)
            out["out_bias_unet"] = (torch.round(bias * 1000) / 1000).numpy()
            sd["decoder.out_conv.bias"] = torch.from_numpy(out["out_bias_unet"]).clone()
            net.load_state_dict(sd)
        rec = Recording(net)
        rec.train()                                   # test_single_volume switches to eval itself (:54)
        del RECORDED[:]
        metrics = ref.test_single_volume(CNNVIT["case"], rec, "/nonexistent/", NS(root_path="/nonexistent", model="unet"))
        assert len(metrics) == 3 and len(RECORDED) == 3 and len(rec.logits) == Z and not rec.training
        pred = np.zeros((Z, X, Y), np.uint8)
        for i, (p, g) in enumerate(RECORDED, start=1):
            # calculate_metric_percase binarises its arguments in place before dc() sees them: still the class-i masks
            assert np.array_equal(g.astype(bool), label == i)
            pred[p.astype(bool)] = i
        ties = np.zeros((Z, X, Y), bool)
        for z, lg in enumerate(rec.logits):
            assert tuple(lg.shape) == (1, C, 224, 224)
            pr = torch.softmax(lg, dim=1)[0]
            top2 = torch.topk(pr, 2, dim=0).values
            ties[z] = zoom(((top2[0] - top2[1]) < 1e-3).numpy().astype(np.uint8), (X / 224, Y / 224), order=0) > 0
            want = zoom(pr.argmax(0).numpy(), (X / 224, Y / 224), order=0)
            assert np.array_equal(want.astype(np.uint8), pred[z]), "recorded masks do not rebuild the reference prediction"
        out[f"prediction_{tag}"], out[f"ties_{tag}"], out[f"weight_scale_{tag}"] = pred, ties, scale
        out[f"logit_samples_{tag}"] = torch.stack(rec.logits)[:, 0].flatten()[::997].numpy()
        print(f"cnnvit_infer {tag}: prediction classes", np.bincount(pred.ravel(), minlength=C), "ties", int(ties.sum()))
    np.savez_compressed(os.path.join(GOLD, "cnnvit_infer.npz"), **out)


# --------------------------------------------------------------------------------------------------- augmentations
AUG2D = dict(seed=11, n=20, out=((48, 40), (64, 64)), lo=20, hi=70)
AUG3D = dict(seed=12, patch=(16, 20, 12), shapes=((30, 26, 22), (23, 37, 15), (14, 40, 30), (16, 29, 16), (12, 11, 8)), n=12)


def aug2d_slices():
    rng = np.random.default_rng(AUG2D["seed"])
    out = []
    for _ in range(AUG2D["n"]):
        H, W = (int(v) for v in rng.integers(AUG2D["lo"], AUG2D["hi"], 2))
        out.append((rng.random((H, W)).astype(np.float32), rng.integers(0, 4, (H, W)).astype(np.uint8)))
    return out


def aug3d_volumes():
    rng = np.random.default_rng(AUG3D["seed"])
    vols = [(rng.random(s).astype(np.float32), rng.integers(0, 2, s).astype(np.uint8)) for s in AUG3D["shapes"]]
    idx = [int(i) for i in rng.integers(0, len(vols), AUG3D["n"])]
    return vols, idx


def gen_aug2d():
    from dataloaders.dataset import RandomGenerator
    slices = aug2d_slices()
    out = dict(input_sum=float(sum(float(s[0].astype(np.float64).sum()) for s in slices)))
    for c, size in enumerate(AUG2D["out"]):
        gen = RandomGenerator(list(size))
        random.seed(500 + c), np.random.seed(600 + c)
        imgs, labs = [], []
        for img, lab in slices:
            s = gen({"image": img, "label": lab})
            assert s["image"].dtype == torch.float32 and s["label"].dtype == torch.uint8
            imgs.append(s["image"].numpy())
            labs.append(s["label"].numpy())
        tail = (random.random(), int(np.random.randint(1 << 30)))
        # the oracle restatement, same seeds: identical bytes, identical draws consumed
        random.seed(500 + c), np.random.seed(600 + c)
        modes = []
        for (img, lab), ri, rl in zip(slices, imgs, labs):
            oi, ol, draws = random_generator(img, lab, size)
            modes.append(draws[0])
            assert np.array_equal(oi, ri) and np.array_equal(ol, rl)
        assert tail == (random.random(), int(np.random.randint(1 << 30)))
        assert set(modes) == {0, 1, 2}, modes
        out[f"image{c}"], out[f"label{c}"] = np.stack(imgs), np.stack(labs)
        out[f"size{c}"], out[f"tail_random{c}"], out[f"tail_np{c}"] = np.array(size), tail[0], tail[1]
        out[f"modes{c}"] = np.array(modes)
        print(f"aug2d {size}: modes {np.bincount(modes)}  tail {tail}")
    out["oracle_vs_reference_worst_rel"] = 0.0       # oracle/augment.py asserted bit-equal above
    np.savez_compressed(os.path.join(GOLD, "aug2d.npz"), **out)


def gen_aug3d():
    from dataloaders.brats2019 import RandomCrop, RandomRotFlip, ToTensor
    vols, idx = aug3d_volumes()
    patch = AUG3D["patch"]
    chain = [RandomRotFlip(), RandomCrop(patch), ToTensor()]      # train_mean_teacher_3D.py:102-106
    np.random.seed(700)
    imgs, labs = [], []
    for i in idx:
        s = {"image": vols[i][0], "label": vols[i][1]}
        for t in chain:
            s = t(s)
        assert s["image"].dtype == torch.float32 and s["label"].dtype == torch.int64
        imgs.append(s["image"].numpy())
        labs.append(s["label"].numpy().astype(np.uint8))
    tail = int(np.random.randint(1 << 30))
    np.random.seed(700)
    for i, ri, rl in zip(idx, imgs, labs):
        oi, ol = rot_flip_crop(vols[i][0], vols[i][1], patch)
        assert np.array_equal(oi, ri) and np.array_equal(ol, rl)
    assert tail == int(np.random.randint(1 << 30))
    np.savez_compressed(os.path.join(GOLD, "aug3d.npz"), image=np.stack(imgs), label=np.stack(labs), idx=np.array(idx),
                        patch=np.array(patch), tail_np=tail, oracle_vs_reference_worst_rel=0.0,
                        input_sum=float(sum(float(v[0].astype(np.float64).sum()) for v in vols)))
    print(f"aug3d: {len(idx)} crops of {patch}, tail {tail}")


SAMPLER = (dict(primary=10, secondary=23, batch=7, secondary_bs=4, seed=0, epochs=2),
           dict(primary=8, secondary=3, batch=4, secondary_bs=2, seed=1, epochs=1),
           dict(primary=24, secondary=100, batch=8, secondary_bs=4, seed=2, epochs=3))


def gen_sampler():
    from dataloaders import brats2019 as ref3d
    from dataloaders import dataset as ref2d
    out = {}
    for c, cfg in enumerate(SAMPLER):
        prim = list(range(cfg["primary"]))
        sec = list(range(cfg["primary"], cfg["primary"] + cfg["secondary"]))
        streams = []
        for mod in (ref2d, ref3d):
            s = mod.TwoStreamBatchSampler(prim, sec, cfg["batch"], cfg["secondary_bs"])
            np.random.seed(cfg["seed"])
            batches = [np.array(b) for _ in range(cfg["epochs"]) for b in s]
            streams.append((np.stack(batches), len(s), int(np.random.randint(1 << 30))))
        assert np.array_equal(streams[0][0], streams[1][0]) and streams[0][1:] == streams[1][1:]
        out[f"batches{c}"], out[f"len{c}"], out[f"tail_np{c}"] = streams[0]
        out[f"cfg{c}"] = np.array([cfg[k] for k in ("primary", "secondary", "batch", "secondary_bs", "seed", "epochs")])
        print(f"sampler {cfg}: {streams[0][0].shape[0]} batches, len {streams[0][1]}")
    out["cases"] = len(SAMPLER)
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"), **out)


def main():
    install_stubs()
    torch.Tensor.cuda = lambda self, *a, **k: self           # no GPU in the build container
    torch.manual_seed(0)
    gen_sampler()
    gen_aug2d()
    gen_aug3d()
    gen_val2d()
    gen_val3d()
    gen_cnnvit_infer()


if __name__ == "__main__":
    main()

"""Sliding-window validation of a 3-D network -- same call surface as the reference's code/val_3D.py.

``test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes)`` (reference :14-79) centre-pads a
volume that is smaller than the patch, tiles it with overlapping patches (the last one of an axis pulled back to the
border), averages the class probabilities where patches overlap and takes the arg-max.

Here the whole evaluation of a volume stays on the device and is batched: the volume is uploaded once, several
windows go through the network per forward (``windows_per_launch``; eval-mode normalisation makes windows
independent, so batching does not change any number), ``mis_softmax_mean_accumulate`` turns a batch of logits into
probabilities in one launch, the score / hit-count volumes are resident, and the arg-max is ``mis_argmax_channels`` --
one device->host copy per volume instead of one per patch.  ``cal_metric`` (reference :82-88) uses the medpy-free
metrics of utils/metrics.py; ``test_all_case`` (:91-118) reads the cases through dataloaders.dataset.read_case
(.h5 via h5py when installed, else .npz).
"""
import itertools
import math

import numpy as np
import torch

from mis_hip import ops
from utils import metrics as metric


def _origins(extent, patch, stride):
    """Window start coordinates along one axis: a regular grid whose last window is clamped to the border."""
    count = math.ceil((extent - patch) / stride) + 1
    return [min(stride * i, extent - patch) for i in range(count)]


def _centre_padding(shape, patch_size):
    """(before, after) zero padding per axis that brings ``shape`` up to the patch size, split like the reference
    (the odd voxel goes after)."""
    pads = []
    for extent, patch in zip(shape, patch_size):
        missing = max(patch - extent, 0)
        pads.append((missing // 2, missing - missing // 2))
    return pads


def test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=1, windows_per_launch=4):
    shape = tuple(image.shape)
    pads = _centre_padding(shape, patch_size)
    vol = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).cuda()
    if any(lo + hi for lo, hi in pads):
        # F.pad lists the last axis first
        vol = torch.nn.functional.pad(vol, [p for lo_hi in reversed(pads) for p in lo_hi], mode="constant", value=0.0)
    padded = tuple(vol.shape)
    grid = [_origins(padded[0], patch_size[0], stride_xy), _origins(padded[1], patch_size[1], stride_xy),
            _origins(padded[2], patch_size[2], stride_z)]
    windows = [tuple(slice(o, o + p) for o, p in zip(corner, patch_size)) for corner in itertools.product(*grid)]

    scores = torch.zeros((num_classes,) + padded, dtype=torch.float32, device="cuda")
    hits = torch.zeros(padded, dtype=torch.float32, device="cuda")
    was_training = net.training
    net.eval()
    try:
        with torch.no_grad():
            for first in range(0, len(windows), windows_per_launch):
                group = windows[first:first + windows_per_launch]
                batch = torch.stack([vol[win] for win in group]).unsqueeze(1)            # [n, 1, *patch]
                logits = net.forward_raw(batch.contiguous(), no_backward=True)
                probs = torch.empty_like(logits)
                ops.softmax_mean_accumulate(logits, probs, 1, 1.0, first=True)           # softmax over the classes
                probs = probs.reshape((len(group), num_classes) + tuple(patch_size))
                for n, win in enumerate(group):
                    scores[(slice(None),) + win] += probs[n]
                    hits[win] += 1
            scores /= hits.unsqueeze(0)
            labels = torch.empty(hits.numel(), dtype=torch.uint8, device="cuda")
            ops.argmax_channels(scores.unsqueeze(0), labels)
    finally:
        net.train(was_training)
    label_map = labels.view(padded).cpu().numpy().astype(np.int64)
    crop = tuple(slice(lo, lo + extent) for (lo, _), extent in zip(pads, shape))
    return label_map[crop]


def cal_metric(gt, pred):
    """[dice, hd95] of one class, zeros when either mask is empty."""
    if pred.sum() == 0 or gt.sum() == 0:
        return np.zeros(2)
    return np.array([metric.dc(pred, gt), metric.hd95(pred, gt)])


def test_all_case(net, base_dir, test_list="full_test.list", num_classes=4, patch_size=(48, 160, 160), stride_xy=32,
                  stride_z=24, shard=None):
    """``shard=(rank, world)``: score only the cases ``rank::world`` and return ``(metric sums, cases scored)`` instead
    of the mean -- the data-parallel training loop shards its in-training validation over the ranks and sums the
    shards (mis_hip.train_common.Validator); the default is the reference's whole-list mean."""
    from dataloaders.dataset import read_case
    with open("{}/{}".format(base_dir, test_list)) as f:
        cases = [ln.replace('\n', '').split(",")[0] for ln in f.readlines()]
    mine = cases if shard is None else cases[shard[0]::shard[1]]
    total = np.zeros((num_classes - 1, 2))
    for case in mine:
        image, label = read_case("{}/data/{}".format(base_dir, case))
        prediction = test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=num_classes)
        for c in range(1, num_classes):
            total[c - 1] += cal_metric(label == c, prediction == c)
    if shard is not None:
        return total, len(mine)
    return total / len(cases)

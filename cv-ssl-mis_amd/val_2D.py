"""Validation of a 2-D network on one volume -- drop-in for the reference's code/val_2D.py.

``test_single_volume(image, label, net, classes, patch_size)`` (:18-39): every slice is resized to ``patch_size``
(nearest, scipy ``zoom(order=0)``), run through ``net`` in eval mode, arg-maxed, resized back; per-class
(dice, hd95) of the stacked prediction.  The resizes, the forward and the channel arg-max run on the device
(``mis_augment2d`` gathers, ``mis_argmax_channels`` -- arg-max of the logits == arg-max of their softmax) with one
upload and one download per volume; the metrics are host-side (utils/metrics.py, medpy-free).
"""
import numpy as np
import torch

from mis_hip import ops
from utils import metrics as metric


def calculate_metric_percase(pred, gt):
    pred[pred > 0] = 1
    gt[gt > 0] = 1
    if pred.sum() > 0:
        dice = metric.dc(pred, gt)
        hd95 = metric.hd95(pred, gt) if gt.sum() > 0 else 0   # medpy raises on an empty reference; score it 0
        return dice, hd95
    else:
        return 0, 0


def predict_slices(image, net, patch_size, slices_per_launch=16):
    """[Z, X, Y] float image -> [Z, X, Y] uint8 label map.  The volume is uploaded once; both nearest resizes
    (scipy ``zoom(order=0)`` in the reference, :24 and :36) are one ``mis_augment2d`` gather launch over all slices
    (bit-exact to scipy, tests/test_augment_gpu.py), the forward runs on batches of slices."""
    from dataloaders.dataset import zoom_slices
    Z, x, y = image.shape
    was_training = net.training
    net.eval()
    try:
        with torch.no_grad():
            vol = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).cuda()
            inp = zoom_slices(vol, patch_size)                                   # [Z, 1, ph, pw]
            pred = torch.empty((Z, patch_size[0], patch_size[1]), dtype=torch.uint8, device="cuda")
            for z0 in range(0, Z, slices_per_launch):
                logits = net.forward_raw(inp[z0:z0 + slices_per_launch].contiguous(), no_backward=True)   # [n, C, 1, ph, pw]
                ops.argmax_channels(logits, pred[z0:z0 + logits.shape[0]].view(-1))
            back = zoom_slices(pred.float(), (x, y))                             # labels are small integers: exact
            prediction = back[:, 0].to(torch.uint8).cpu().numpy()
    finally:
        net.train(was_training)
    return prediction


def test_single_volume(image, label, net, classes, patch_size=[256, 256]):
    image, label = image.squeeze(0).cpu().detach().numpy(), label.squeeze(0).cpu().detach().numpy()
    prediction = predict_slices(image, net, patch_size)
    metric_list = []
    for i in range(1, classes):
        metric_list.append(calculate_metric_percase(prediction == i, label == i))
    return metric_list

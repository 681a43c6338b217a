"""Oracle Mean-Teacher iteration.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates the loop body of the reference scripts on torch CPU fp32:
  2D  code/train_mean_teacher_2D.py:202-236  (consistency forced to 0.0 while iter_num < 1000, :224-228)
  3D  code/train_mean_teacher_3D.py:134-166  (consistency always on, :156-157)
with optim.SGD(momentum=0.9, weight_decay=1e-4) (:189-190), update_ema_variables (:124-128) and the
poly learning-rate rule (:234-236).  The teacher runs in train mode (never .eval()'d) and only
``parameters()`` are EMA'd -- BatchNorm buffers are not.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .losses import consistency_weight, dice_loss


def lr_for_step(k, base_lr, max_iterations, post_increment=False):
    """Learning rate in effect while step k runs (set after step k-1; lr_0 = base_lr)."""
    if k == 0:
        return base_lr
    it = k if post_increment else k - 1
    return base_lr * (1.0 - it / max_iterations) ** 0.9


def ema_alpha(k, ema_decay):
    return min(1 - 1 / (k + 1), ema_decay)


def cross_teaching_step(net1, net2, sd1, sd2, mom1, mom2, volume, label, iter_num, *, labeled_bs, num_classes,
                        base_lr=0.01, max_iterations=30000, consistency=0.1, rampup=200.0, sgd_momentum=0.9,
                        weight_decay=1e-4, drop1=None, drop2=None, apply_update=True, pseudo_ce=False):
    """One iteration of cross teaching between two students (reference
    code/train_cross_teaching_between_cnn_transformer_2D.py:216-263): CE+Dice on the labeled half, Dice against
    the other network's arg-max pseudo labels on the unlabeled half, one backward of loss1+loss2, two SGD steps;
    the learning rate uses the post-increment rule (:255-263).  State dicts are mutated in place."""
    L = labeled_bs
    works, outs = [], []
    for net, sd, drop in ((net1, sd1, drop1), (net2, sd2, drop2)):
        work = OrderedDict((n, t.detach().clone().requires_grad_(True)) if net.is_param(n) else (n, t)
                           for n, t in sd.items())
        works.append(work)
        outs.append(net.forward(work, volume, training=True, drop=drop))
    soft = [torch.softmax(o, dim=1) for o in outs]
    w = consistency_weight(iter_num, consistency, rampup)
    losses, parts = [], []
    for m in (0, 1):
        ce = F.cross_entropy(outs[m][:L], label[:L].long())
        dl = dice_loss(soft[m][:L], label[:L].unsqueeze(1), num_classes)
        pseudo = torch.argmax(soft[1 - m][L:].detach(), dim=1, keepdim=False)
        if pseudo_ce:       # cross pseudo supervision, code/train_cross_pseudo_supervision_3D.py:171-172
            ps = F.cross_entropy(outs[m][L:], pseudo)
        else:
            ps = dice_loss(soft[m][L:], pseudo.unsqueeze(1), num_classes)
        losses.append(0.5 * (ce + dl) + w * ps)
        parts.append((float(ce.detach()), float(dl.detach()), float(ps.detach())))
    loss = losses[0] + losses[1]
    plist = [(m, n) for m, net in enumerate((net1, net2)) for n in works[m] if net.is_param(n)]
    grads = torch.autograd.grad(loss, [works[m][n] for m, n in plist])
    g = [OrderedDict(), OrderedDict()]
    for (m, n), gr in zip(plist, grads):
        g[m][n] = gr
    lr = lr_for_step(iter_num, base_lr, max_iterations, post_increment=True)
    if apply_update:
        with torch.no_grad():
            for sd, mom, gm in ((sd1, mom1, g[0]), (sd2, mom2, g[1])):
                for n, gr in gm.items():
                    d = gr + weight_decay * sd[n]
                    if n in mom:
                        mom[n].mul_(sgd_momentum).add_(d)
                    else:
                        mom[n] = d.clone()
                    sd[n].sub_(lr * mom[n])
    return dict(loss=float(loss.detach()), model1_loss=float(losses[0].detach()), model2_loss=float(losses[1].detach()),
                parts=parts, consistency_weight=w, lr=lr, logits1=outs[0].detach(), logits2=outs[1].detach(), grads=g)


def linear_rampup(current, rampup_length):
    """code/utils/ramps.py:49-55"""
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


def cnn_meet_vit_weights(iter_num, consistency=0.1, rampup=200.0, mt_start_iter=1000):
    """(pseudo-supervision weight, mean-teacher weight) of code/train_cnn_meet_vit_2D.py:322-337:
    ``7 * consistency * linear_rampup(iter_num // 150, rampup)`` and ``consistency * linear_rampup(...)``, the
    latter multiplying a consistency loss that is 0.0 while ``iter_num < 1000``."""
    w = consistency * linear_rampup(iter_num // 150, rampup)
    return 7 * w, (w if iter_num >= mt_start_iter else 0.0)


def cnn_meet_vit_step(net1, net2, sd1, sd2, tsd, mom1, mom2, volume, label, noise, iter_num, *, labeled_bs,
                      num_classes, base_lr=0.01, max_iterations=30000, ema_decay=0.99, consistency=0.1, rampup=200.0,
                      sgd_momentum=0.9, weight_decay=1e-4, drop1=None, drop2=None, drop_t=None, apply_update=True):
    """One iteration of code/train_cnn_meet_vit_2D.py:293-352: a CNN student (net1/sd1) and a Transformer student
    (net2/sd2) cross-teach through Dice on each other's arg-max pseudo labels (weight 7*w), and BOTH are pulled
    (weight w, from iteration 1000) towards an EMA teacher of the Transformer (``tsd``, architecture net2) that sees
    the noised unlabeled half; one backward of model1_loss + model2_loss, two SGD steps, EMA of model2 only
    (:345), learning rate computed before ``iter_num`` is incremented (:347-348, the Mean-Teacher rule)."""
    L = labeled_bs
    works, outs = [], []
    for net, sd, drop in ((net1, sd1, drop1), (net2, sd2, drop2)):
        work = OrderedDict((n, t.detach().clone().requires_grad_(True)) if net.is_param(n) else (n, t)
                           for n, t in sd.items())
        works.append(work)
        outs.append(net.forward(work, volume, training=True, drop=drop))
    soft = [torch.softmax(o, dim=1) for o in outs]
    with torch.no_grad():
        ema_output = net2.forward(tsd, volume[L:] + noise, training=True, drop=drop_t)
        ema_soft = torch.softmax(ema_output, dim=1)
    w_cps, w_mt = cnn_meet_vit_weights(iter_num, consistency, rampup)
    losses, parts = [], []
    for m in (0, 1):
        ce = F.cross_entropy(outs[m][:L], label[:L].long())
        dl = dice_loss(soft[m][:L], label[:L].unsqueeze(1), num_classes)
        pseudo = torch.argmax(soft[1 - m][L:].detach(), dim=1, keepdim=False)
        ps = dice_loss(soft[m][L:], pseudo.unsqueeze(1), num_classes)
        cons = torch.mean((soft[m][L:] - ema_soft) ** 2) if iter_num >= 1000 else torch.zeros(())
        losses.append(0.5 * (ce + dl) + w_cps * ps + w_mt * cons)
        parts.append((float(ce.detach()), float(dl.detach()), float(ps.detach()), float(cons.detach())))
    loss = losses[0] + losses[1]
    plist = [(m, n) for m, net in enumerate((net1, net2)) for n in works[m] if net.is_param(n)]
    grads = torch.autograd.grad(loss, [works[m][n] for m, n in plist])
    g = [OrderedDict(), OrderedDict()]
    for (m, n), gr in zip(plist, grads):
        g[m][n] = gr
    lr = lr_for_step(iter_num, base_lr, max_iterations)
    alpha = ema_alpha(iter_num, ema_decay)
    if apply_update:
        with torch.no_grad():
            for sd, mom, gm in ((sd1, mom1, g[0]), (sd2, mom2, g[1])):
                for n, gr in gm.items():
                    d = gr + weight_decay * sd[n]
                    if n in mom:
                        mom[n].mul_(sgd_momentum).add_(d)
                    else:
                        mom[n] = d.clone()
                    sd[n].sub_(lr * mom[n])
            for n in g[1]:
                tsd[n].mul_(alpha).add_(sd2[n], alpha=1 - alpha)
    return dict(loss=float(loss.detach()), model1_loss=float(losses[0].detach()), model2_loss=float(losses[1].detach()),
                parts=parts, pseudo_weight=w_cps, mt_weight=w_mt, lr=lr, ema_alpha=alpha, logits1=outs[0].detach(),
                logits2=outs[1].detach(), teacher_logits=ema_output.detach(), grads=g)


def mean_teacher_step(net, student, teacher, momentum, volume, label, noise, iter_num, *, labeled_bs,
                      num_classes, base_lr=0.01, max_iterations=30000, ema_decay=0.99, consistency=0.1,
                      rampup=200.0, cons_start_iter=1000, sgd_momentum=0.9, weight_decay=1e-4,
                      drop_student=None, drop_teacher=None, apply_update=True, grad_hook=None):
    """One iteration.  ``student``/``teacher``: state dicts (mutated in place), ``momentum``: dict of
    SGD buffers (mutated; missing entries = first step).  Returns a dict of python floats + tensors.
    """
    params = [n for n in student if net.is_param(n)]
    work = OrderedDict((n, t.detach().clone().requires_grad_(True)) if n in params else (n, t)
                       for n, t in student.items())
    unl = volume[labeled_bs:]
    ema_inputs = unl + noise
    outputs = net.forward(work, volume, training=True, drop=drop_student)
    outputs_soft = torch.softmax(outputs, dim=1)
    with torch.no_grad():
        ema_output = net.forward(teacher, ema_inputs, training=True, drop=drop_teacher)
        ema_output_soft = torch.softmax(ema_output, dim=1)
    loss_ce = F.cross_entropy(outputs[:labeled_bs], label[:labeled_bs].long())
    loss_dice = dice_loss(outputs_soft[:labeled_bs], label[:labeled_bs].unsqueeze(1), num_classes)
    supervised = 0.5 * (loss_dice + loss_ce)
    w = consistency_weight(iter_num, consistency, rampup)
    if iter_num < cons_start_iter:
        cons = torch.zeros(())
    else:
        cons = torch.mean((outputs_soft[labeled_bs:] - ema_output_soft) ** 2)
    loss = supervised + w * cons
    # allow_unused: a parameter that takes no part in the forward (UNETR's ViT cls_token with classification off) gets a
    # zero gradient here.  (torch's optimizer would SKIP such a parameter -- grad None: no weight decay either; the HIP
    # step treats the flat buffer uniformly, i.e. like the zero gradient written here.)
    grads = torch.autograd.grad(loss, [work[n] for n in params], allow_unused=True)
    grads = OrderedDict((n, g if g is not None else torch.zeros_like(work[n])) for n, g in zip(params, grads))
    if grad_hook is not None:       # e.g. data-parallel averaging across shards
        grads = grad_hook(grads)
    lr = lr_for_step(iter_num, base_lr, max_iterations)
    alpha = ema_alpha(iter_num, ema_decay)
    if apply_update:
        with torch.no_grad():
            for n in params:
                d = grads[n] + weight_decay * student[n]
                if n in momentum:
                    momentum[n].mul_(sgd_momentum).add_(d)
                else:
                    momentum[n] = d.clone()
                student[n].sub_(lr * momentum[n])
                teacher[n].mul_(alpha).add_(student[n], alpha=1 - alpha)
    return dict(loss=float(loss.detach()), loss_ce=float(loss_ce.detach()), loss_dice=float(loss_dice.detach()),
                consistency_loss=float(cons.detach()), consistency_weight=w, lr=lr, ema_alpha=alpha,
                logits=outputs.detach(), teacher_logits=ema_output.detach(), grads=grads)


def uamt_threshold(iter_num, max_iterations):
    """(0.75 + 0.25 * ramps.sigmoid_rampup(iter_num, max_iterations)) * ln 2
    (code/train_uncertainty_aware_mean_teacher_3D.py:173-174, code/utils/ramps.py:13-22)."""
    import numpy as np
    cur = np.clip(iter_num, 0.0, max_iterations)
    phase = 1.0 - cur / max_iterations
    return (0.75 + 0.25 * float(np.exp(-5.0 * phase * phase))) * np.log(2)


def uamt_step(net, student, teacher, momentum, volume, label, noise, mc_noise, iter_num, *, labeled_bs,
              num_classes, base_lr=0.01, max_iterations=30000, ema_decay=0.99, consistency=0.1, rampup=200.0,
              sgd_momentum=0.9, weight_decay=1e-4, drop_student=None, drop_teacher=None, apply_update=True):
    """One UA-MT iteration (code/train_uncertainty_aware_mean_teacher_3D.py:134-189, _2D.py:146-201).
    ``mc_noise``: the T//2 = 4 noise tensors added to ``repeat(unlabeled, 2)``.  The teacher state dict is mutated
    by all 5 train-mode forwards (BatchNorm running statistics), as in the reference."""
    T = 8
    params = [n for n in student if net.is_param(n)]
    work = OrderedDict((n, t.detach().clone().requires_grad_(True)) if n in params else (n, t)
                       for n, t in student.items())
    L = labeled_bs
    unl = volume[L:]
    outputs = net.forward(work, volume, training=True, drop=drop_student)
    outputs_soft = torch.softmax(outputs, dim=1)
    with torch.no_grad():
        ema_output = net.forward(teacher, unl + noise, training=True, drop=drop_teacher)
        rep = unl.repeat(2, *([1] * (unl.dim() - 1)))
        stride = rep.shape[0] // 2
        preds = torch.zeros((stride * T,) + tuple(ema_output.shape[1:]))
        for i in range(T // 2):
            preds[2 * stride * i:2 * stride * (i + 1)] = net.forward(teacher, rep + mc_noise[i], training=True,
                                                                     drop=drop_teacher)
        preds = torch.softmax(preds, dim=1)
        preds = preds.reshape((T, stride) + tuple(ema_output.shape[1:])).mean(dim=0)
        uncertainty = -1.0 * torch.sum(preds * torch.log(preds + 1e-6), dim=1, keepdim=True)
    loss_ce = F.cross_entropy(outputs[:L], label[:L].long())
    loss_dice = dice_loss(outputs_soft[:L], label[:L].unsqueeze(1), num_classes)
    supervised = 0.5 * (loss_dice + loss_ce)
    w = consistency_weight(iter_num, consistency, rampup)
    dist = (outputs_soft[L:] - torch.softmax(ema_output, dim=1)) ** 2          # losses.softmax_mse_loss
    threshold = uamt_threshold(iter_num, max_iterations)
    mask = (uncertainty < threshold).float()
    cons = torch.sum(mask * dist) / (2 * torch.sum(mask) + 1e-16)
    loss = supervised + w * cons
    grads = OrderedDict(zip(params, torch.autograd.grad(loss, [work[n] for n in params])))
    lr = lr_for_step(iter_num, base_lr, max_iterations)
    alpha = ema_alpha(iter_num, ema_decay)
    if apply_update:
        with torch.no_grad():
            for n in params:
                d = grads[n] + weight_decay * student[n]
                if n in momentum:
                    momentum[n].mul_(sgd_momentum).add_(d)
                else:
                    momentum[n] = d.clone()
                student[n].sub_(lr * momentum[n])
                teacher[n].mul_(alpha).add_(student[n], alpha=1 - alpha)
    return dict(loss=float(loss.detach()), loss_ce=float(loss_ce.detach()), loss_dice=float(loss_dice.detach()),
                consistency_loss=float(cons.detach()), consistency_weight=w, lr=lr, ema_alpha=alpha,
                threshold=float(threshold), unmasked=float(mask.sum()), uncertainty=uncertainty,
                logits=outputs.detach(), teacher_logits=ema_output.detach(), grads=grads)

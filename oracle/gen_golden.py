"""Generate tests/golden/*.npz from the REAL reference, and pin the oracle against it.

Run in the build container only (it needs /root/reference, which never travels):

    python oracle/gen_golden.py

For every case it (1) runs the reference's own modules (networks.unet.UNet, networks.unet_3D.unet_3D,
utils.losses.DiceLoss, utils.ramps, torch.optim.SGD, the EMA loop) around a restatement of the ~25-line
loop body of train_mean_teacher_{2D,3D}.py that cannot be imported (module-level argparse + absent
deps), (2) runs oracle.step.mean_teacher_step on identical filler inputs, (3) asserts
they agree to <= 1e-5 (relative to scale), and (4) stores the REFERENCE numbers as the golden vector.
Fixtures are data only: scalars, checksums and sampled values -- no reference source text.
"""
import json
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import filler  # noqa: E402
from oracle.nets import OracleUNet2D, OracleUNet3D  # noqa: E402
from oracle.step import mean_teacher_step  # noqa: E402

N_SAMPLES = 64


def sample_idx(numel):
    return np.unique(np.linspace(0, numel - 1, N_SAMPLES).astype(np.int64))


def tensor_summary(t):
    t = t.detach().double().flatten()
    idx = sample_idx(t.numel())
    return dict(sum=float(t.sum()), abssum=float(t.abs().sum()), max=float(t.max()), min=float(t.min()),
                samples=t[idx].numpy())


class MaskDrop(torch.nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask


def _install_timm_shim():
    """The reference's Swin file imports 3 symbols from timm (absent here): container-only stand-ins with
    timm's semantics (DropPath = per-sample Bernoulli(1-p)/(1-p) on the residual branch)."""
    import types
    if "timm.models.layers" in sys.modules:
        return

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.):
            super().__init__()
            self.drop_prob = p

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    tl = types.ModuleType("timm.models.layers")
    tl.DropPath, tl.trunc_normal_ = DropPath, torch.nn.init.trunc_normal_
    tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    sys.modules["timm"] = types.ModuleType("timm")
    sys.modules["timm.models"] = types.ModuleType("timm.models")
    sys.modules["timm.models.layers"] = tl


class SeqScale(torch.nn.Module):
    """Injected DropPath: the block calls its drop_path twice per forward (attention, MLP residual)."""

    def __init__(self, scales):
        super().__init__()
        self.scales, self.i = scales, 0

    def forward(self, x):
        s = self.scales[self.i % 2]
        self.i += 1
        return x * s.to(x.dtype).view(-1, *([1] * (x.dim() - 1)))


def _swin_blocks(model):
    sw = model.swin_unet
    blocks = [b for layer in sw.layers for b in layer.blocks]
    for i in range(1, len(sw.layers_up)):
        blocks += list(sw.layers_up[i].blocks)
    return blocks


def build_reference(kind, in_chns, num_classes):
    sys.path.insert(0, REF)
    if kind == "unet2d":
        from networks.unet import UNet
        return UNet(in_chns=in_chns, class_num=num_classes)
    if kind == "unet2d_deconv":
        # UpBlock(bilinear=False) (unet.py:76-78): the reference's Decoder never passes the flag (:129-136), so the real
        # UNet is built and its four UpBlocks are replaced by real UpBlock(..., bilinear=False) modules of the same widths
        from networks.unet import UNet, UpBlock
        net = UNet(in_chns=in_chns, class_num=num_classes)
        ft = net.decoder.ft_chns
        for i in range(1, 5):
            setattr(net.decoder, f"up{i}", UpBlock(ft[5 - i], ft[4 - i], ft[4 - i], dropout_p=0.0, bilinear=False))
        return net
    if kind in ("swin", "swin_w8"):
        _install_timm_shim()
        from types import SimpleNamespace as NS
        from networks.vision_transformer import SwinUnet
        # "swin_w8": DATA.IMG_SIZE 256 + MODEL.SWIN.WINDOW_SIZE 8 (the --opts overrides of code/config.py:194-195)
        img, win = (256, 8) if kind == "swin_w8" else (224, 7)
        cfg = NS(DATA=NS(IMG_SIZE=img),
                 MODEL=NS(DROP_RATE=0.0, DROP_PATH_RATE=0.2, PRETRAIN_CKPT=None,
                          SWIN=NS(PATCH_SIZE=4, IN_CHANS=3, EMBED_DIM=96, DEPTHS=[2, 2, 2, 2], NUM_HEADS=[3, 6, 12, 24],
                                  WINDOW_SIZE=win, MLP_RATIO=4., QKV_BIAS=True, QK_SCALE=None, APE=False,
                                  PATCH_NORM=True)),
                 TRAIN=NS(USE_CHECKPOINT=False))
        return SwinUnet(cfg, img_size=img, num_classes=num_classes)
    if kind.startswith("vnet"):       # "vnet" = batchnorm (the factory's), "vnet_groupnorm", "vnet_instancenorm", ...
        from networks.vnet import VNet
        norm = kind.split("_", 1)[1] if "_" in kind else "batchnorm"
        return VNet(n_channels=in_chns, n_classes=num_classes, normalization=norm, has_dropout=True)
    from networks.unet_3D import unet_3D
    return unet_3D(n_classes=num_classes, in_channels=in_chns)


class SeqMask(torch.nn.Module):
    """Injected Dropout3d of VNet: ONE module called twice per forward (after block_five, after block_nine)."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.i = masks, 0

    def forward(self, x):
        m = self.masks[self.i % len(self.masks)]
        self.i += 1
        return x * m


def set_reference_dropout(model, kind, drop, sites):
    """drop == 'off': p := 0; dict: inject masks in forward-site order."""
    if kind in ("swin", "swin_w8"):
        for bi, blk in enumerate(_swin_blocks(model)):
            if drop == "off" or (2 * bi) not in drop:
                blk.drop_path = torch.nn.Identity()
            else:
                blk.drop_path = SeqScale([drop[2 * bi], drop[2 * bi + 1]])
        return
    if kind.startswith("unet2d"):
        blocks = [model.encoder.in_conv, model.encoder.down1.maxpool_conv[1], model.encoder.down2.maxpool_conv[1],
                  model.encoder.down3.maxpool_conv[1], model.encoder.down4.maxpool_conv[1]]
        for site, blk in enumerate(blocks):
            blk.conv_conv[3] = torch.nn.Identity() if drop == "off" else MaskDrop(drop[site])
        for up in (model.decoder.up1, model.decoder.up2, model.decoder.up3, model.decoder.up4):
            assert up.conv.conv_conv[3].p == 0.0
    elif kind.startswith("vnet"):
        assert isinstance(model.dropout, (torch.nn.Dropout3d, torch.nn.Identity, SeqMask))
        model.dropout = torch.nn.Identity() if drop == "off" else SeqMask([drop[0], drop[1]])
    else:
        model.dropout1 = torch.nn.Identity() if drop == "off" else MaskDrop(drop[0])
        model.dropout2 = torch.nn.Identity() if drop == "off" else MaskDrop(drop[1])


def reference_step(kind, model, ema_model, optimizer, volume, label, noise, iter_num, cfg):
    """Loop body of train_mean_teacher_2D.py:202-236 / _3D.py:134-166 around the reference modules."""
    from utils import losses, ramps
    L = cfg["labeled_bs"]
    dice = losses.DiceLoss(cfg["num_classes"])
    ce = torch.nn.CrossEntropyLoss()
    ema_inputs = volume[L:] + noise
    outputs = model(volume)
    outputs_soft = torch.softmax(outputs, dim=1)
    with torch.no_grad():
        ema_output = ema_model(ema_inputs)
        ema_output_soft = torch.softmax(ema_output, dim=1)
    loss_ce = ce(outputs[:L], label[:L].long())
    loss_dice = dice(outputs_soft[:L], label[:L].unsqueeze(1))
    supervised = 0.5 * (loss_dice + loss_ce)
    w = cfg["consistency"] * ramps.sigmoid_rampup(iter_num // 150, cfg["rampup"])
    if iter_num < cfg["cons_start_iter"]:
        cons = torch.zeros(())
    else:
        cons = torch.mean((outputs_soft[L:] - ema_output_soft) ** 2)
    loss = supervised + w * cons
    optimizer.zero_grad()
    loss.backward()
    grads = [p.grad.detach().clone() for p in model.parameters()]
    lr_used = optimizer.param_groups[0]["lr"]
    optimizer.step()
    alpha = min(1 - 1 / (iter_num + 1), cfg["ema_decay"])
    for ema_p, p in zip(ema_model.parameters(), model.parameters()):
        ema_p.data.mul_(alpha).add_(p.data, alpha=1 - alpha)
    lr_ = cfg["base_lr"] * (1.0 - iter_num / cfg["max_iterations"]) ** 0.9
    for g in optimizer.param_groups:
        g["lr"] = lr_
    return dict(loss=float(loss), loss_ce=float(loss_ce), loss_dice=float(loss_dice),
                consistency_loss=float(cons), consistency_weight=w, lr=lr_used, logits=outputs.detach(),
                teacher_logits=ema_output.detach(), grads=grads)


def reference_grads64(kind, cfg, sd0, tsd0, drop_s, drop_t, volume, label, noise, iter_num, flips=0):
    """The same loop body with the reference modules in float64: measures the reference's own fp32
    rounding noise per gradient tensor (the tolerance envelope of the gradient-level parity tests).

    ``flips`` > 0: also returns the FLIP ENVELOPE.  At the deepest levels (<= 4^3 voxels per channel) some ReLU
    pre-activation of the student always lies within ~1e-6 (relative) of zero -- measured over ten input draws: 1.6e-7 ...
    5.9e-6 -- i.e. within the rounding error of ANY fp32 convolution: which side of the discontinuity an implementation
    lands on is a coin toss, and one flipped element moves a whole weight-gradient tensor of that level.  The float64 step
    is therefore re-evaluated with the sign of each of the ``flips`` smallest such pre-activations reversed (one at a
    time); the largest change per gradient tensor, relative to the tensor's maximum, is what a single legitimate flip
    costs on this fixture.  Returns (grads, flip_relerr per tensor, the margins that were flipped)."""
    C = cfg["num_classes"]
    m64, e64 = build_reference(kind, 1, C).double(), build_reference(kind, 1, C).double()
    to64 = lambda d: d if d == "off" else {k: v.double() for k, v in d.items()}

    def run(hook_factory=None):
        m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()})
        e64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in tsd0.items()})
        m64.train(); e64.train()
        set_reference_dropout(m64, kind, to64(drop_s), None)
        set_reference_dropout(e64, kind, to64(drop_t), None)
        opt = torch.optim.SGD(m64.parameters(), lr=0.0)
        relus = [m for m in m64.modules() if isinstance(m, (torch.nn.ReLU, torch.nn.LeakyReLU))]
        hooks = [m.register_forward_pre_hook(hook_factory(i)) for i, m in enumerate(relus)] if hook_factory else []
        ref = reference_step(kind, m64, e64, opt, volume.double(), label, noise.double(), iter_num, cfg)
        for h in hooks:
            h.remove()
        return [g.clone() for g in ref["grads"]]

    if not flips:
        return run()
    cands = []          # (margin relative to the layer's largest |x|, relu index, flat element index)

    def probe(i):
        def pre(mod, inp):
            x = inp[0].detach()
            if x.dim() >= 4 and int(np.prod(x.shape[2:])) <= 64:
                a = x.abs().flatten()
                k = torch.topk(a, min(flips, a.numel()), largest=False)
                cands.extend((float(v / a.max()), i, int(j)) for v, j in zip(k.values, k.indices))
        return pre

    g64 = run(probe)
    cands.sort()
    env = [0.0] * len(g64)
    for margin, ri, j in cands[:flips]:
        def flip(i, ri=ri, j=j):
            def pre(mod, inp):
                if i == ri:
                    x = inp[0].clone()
                    x.view(-1)[j] = -x.view(-1)[j]
                    return (x,)
            return pre
        gf = run(flip)
        env = [max(e, float((a - b).abs().max() / (b.abs().max() + 1e-300))) for e, a, b in zip(env, gf, g64)]
    return g64, env, [c[0] for c in cands[:flips]]


def rel_close(a, b, tol, what):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-6)
    assert err <= tol * scale + 1e-9, f"oracle != reference for {what}: err {err:.3e} scale {scale:.3e}"
    return err / scale


def make_inputs(kind, cfg):
    B = cfg["batch_size"]
    sp = tuple(cfg["spatial"])
    tag = cfg.get("tag", "")          # another draw of the closed-form inputs for the same geometry
    volume = filler.image((B, cfg.get("in_channels", 1)) + sp, "volume" + tag)
    ldt = torch.uint8 if kind in ("unet2d", "unet2d_deconv", "swin", "swin_w8") else torch.int64
    label = filler.labels((B,) + sp, cfg["num_classes"], ldt)
    noise = filler.noise((B - cfg["labeled_bs"], cfg.get("in_channels", 1)) + sp, "noise" + tag)
    return volume, label, noise


def run_case(name, kind, cfg, iters, drop_mode, eval_logits=False):
    torch.manual_seed(0)
    C = cfg["num_classes"]
    if kind == "swin":
        from oracle.swin import OracleSwinUnet
        onet = OracleSwinUnet(C)
    elif kind == "swin_w8":
        from oracle.swin import OracleSwinUnet
        onet = OracleSwinUnet(C, img_size=256, window=8)
    elif kind.startswith("vnet"):
        from oracle.nets import OracleVNet
        onet = OracleVNet(C, 1, normalization=kind.split("_", 1)[1] if "_" in kind else "batchnorm")
    elif kind == "unet2d_deconv":
        onet = OracleUNet2D(1, C, bilinear=False)
    else:
        onet = OracleUNet2D(1, C) if kind == "unet2d" else OracleUNet3D(C, 1)
    model = build_reference(kind, 1, C)
    ema_model = build_reference(kind, 1, C)
    for p in ema_model.parameters():
        p.detach_()
    sd0 = filler.fill_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    assert list(sd0.keys()) == [s[0] for s in onet.spec()], "oracle state_dict keys != reference keys"
    model.load_state_dict(sd0)
    tsd0 = filler.fill_state_dict({"t." + k: v.clone() for k, v in model.state_dict().items()})
    tsd0 = {k[2:]: v for k, v in tsd0.items()}
    ema_model.load_state_dict(tsd0)
    volume, label, noise = make_inputs(kind, cfg)
    out = dict(meta=json.dumps(dict(name=name, kind=kind, cfg=cfg, iters=iters, drop_mode=drop_mode)))
    worst = 0.0

    if eval_logits:
        model.eval()
        with torch.no_grad():
            ref_logits = model(volume)
        o_logits = onet.forward({k: v.clone() for k, v in sd0.items()}, volume, training=False)
        worst = max(worst, rel_close(o_logits, ref_logits, 1e-5, "eval logits"))
        for k, v in tensor_summary(ref_logits).items():
            out[f"eval_logits_{k}"] = np.asarray(v)

    model.train()
    ema_model.train()
    in_shape = tuple(volume.shape)
    t_shape = (in_shape[0] - cfg["labeled_bs"],) + in_shape[1:]
    if drop_mode == "off":
        drop_s = drop_t = "off"
    else:
        drop_s = {s: filler.drop_mask(shape, p, f"drop_s{s}") for s, p, shape in onet.drop_sites(in_shape)}
        drop_t = {s: filler.drop_mask(shape, p, f"drop_t{s}") for s, p, shape in onet.drop_sites(t_shape)}
    set_reference_dropout(model, kind, drop_s, None)
    set_reference_dropout(ema_model, kind, drop_t, None)

    for it in iters:
        # every iteration restarts from the same fixture state (parity is per step from identical state)
        model.load_state_dict(sd0)
        ema_model.load_state_dict(tsd0)
        optimizer = torch.optim.SGD(model.parameters(), lr=cfg["base_lr"], momentum=0.9, weight_decay=0.0001)
        if it > 0:
            # momentum buffers + lr as they would be after step it-1: deterministic filler buffers
            for n, p in model.named_parameters():
                optimizer.state[p]["momentum_buffer"] = filler.uniform(p.shape, "mom." + n, -0.01, 0.01)
            lr_prev = cfg["base_lr"] * (1.0 - (it - 1) / cfg["max_iterations"]) ** 0.9
            for g in optimizer.param_groups:
                g["lr"] = lr_prev
        ref = reference_step(kind, model, ema_model, optimizer, volume, label, noise, it, cfg)

        student = {k: v.clone() for k, v in sd0.items()}
        teacher = {k: v.clone() for k, v in tsd0.items()}
        mom = {}
        if it > 0:
            mom = {n: filler.uniform(student[n].shape, "mom." + n, -0.01, 0.01)
                   for n in student if onet.is_param(n)}
        orc = mean_teacher_step(onet, student, teacher, mom, volume, label, noise, it,
                                labeled_bs=cfg["labeled_bs"], num_classes=C, base_lr=cfg["base_lr"],
                                max_iterations=cfg["max_iterations"], ema_decay=cfg["ema_decay"],
                                consistency=cfg["consistency"], rampup=cfg["rampup"],
                                cons_start_iter=cfg["cons_start_iter"], drop_student=drop_s, drop_teacher=drop_t)
        # ---- pin the oracle against the reference ----
        for k in ("loss", "loss_ce", "loss_dice", "consistency_loss", "consistency_weight", "lr"):
            worst = max(worst, rel_close(orc[k], ref[k], 1e-5, f"{name} it{it} {k}"))
        worst = max(worst, rel_close(orc["logits"], ref["logits"], 1e-5, f"{name} it{it} logits"))
        worst = max(worst, rel_close(orc["teacher_logits"], ref["teacher_logits"], 1e-5, "teacher logits"))
        pnames = [n for n, _ in model.named_parameters()]
        for n, g in zip(pnames, ref["grads"]):
            rel_close(orc["grads"][n], g, 2e-4, f"{name} it{it} grad {n}")
        ref_sd = model.state_dict()
        ref_tsd = ema_model.state_dict()
        for n in ref_sd:
            if n.endswith("num_batches_tracked"):
                assert int(ref_sd[n]) == int(student[n])
                continue
            rel_close(student[n], ref_sd[n], 1e-5, f"{name} it{it} post-SGD {n}")
            rel_close(teacher[n], ref_tsd[n], 1e-5, f"{name} it{it} post-EMA {n}")
        # ---- golden = reference numbers ----
        pre = f"it{it}_"
        for k in ("loss", "loss_ce", "loss_dice", "consistency_loss", "consistency_weight", "lr"):
            out[pre + k] = np.float64(ref[k])
        for k, v in tensor_summary(ref["logits"]).items():
            out[pre + "logits_" + k] = np.asarray(v)
        for k, v in tensor_summary(ref["teacher_logits"]).items():
            out[pre + "teacher_logits_" + k] = np.asarray(v)
        out[pre + "grad_norms"] = np.array([float(g.double().norm()) for g in ref["grads"]])
        if cfg.get("flips"):
            g64, flip_env, margins = reference_grads64(kind, cfg, sd0, tsd0, drop_s, drop_t, volume, label, noise, it,
                                                       flips=cfg["flips"])
            out[pre + "grad_flip_relerr"] = np.array(flip_env)
            out[pre + "flip_margins"] = np.array(margins)
            print(f"{name} it{it}: flip envelope over {len(margins)} pre-activations with margins "
                  f"{', '.join('%.1e' % m for m in margins)}: largest per-tensor change {max(flip_env):.2e}")
        else:
            g64 = reference_grads64(kind, cfg, sd0, tsd0, drop_s, drop_t, volume, label, noise, it)
        out[pre + "grad_norms64"] = np.array([float(g.norm()) for g in g64])
        out[pre + "grad_max64"] = np.array([float(g.abs().max()) for g in g64])
        out[pre + "grad_relerr32"] = np.array([float((a.double() - b).abs().max() / (b.abs().max() + 1e-300))
                                               for a, b in zip(ref["grads"], g64)])
        out[pre + "student_sum"] = np.array([float(ref_sd[n].double().sum()) for n in pnames])
        out[pre + "student_abssum"] = np.array([float(ref_sd[n].double().abs().sum()) for n in pnames])
        out[pre + "teacher_sum"] = np.array([float(ref_tsd[n].double().sum()) for n in pnames])
        out[pre + "teacher_abssum"] = np.array([float(ref_tsd[n].double().abs().sum()) for n in pnames])
        bufs = [n for n in ref_sd if n.endswith("running_mean") or n.endswith("running_var")]
        if bufs:
            out[pre + "student_buf_sum"] = np.array([float(ref_sd[n].double().sum()) for n in bufs])
            out[pre + "teacher_buf_sum"] = np.array([float(ref_tsd[n].double().sum()) for n in bufs])
    out["oracle_vs_reference_worst_rel"] = np.float64(worst)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"{name}: oracle vs reference worst rel err {worst:.2e}; wrote {name}.npz")


def run_cross_case(name, cfg, it, kinds=("unet2d", "swin")):
    """Cross teaching UNet <-> SwinUnet (config 5 geometry, 224x224): reference modules in the restated loop of
    train_cross_teaching_between_cnn_transformer_2D.py:216-263 vs oracle.step.cross_teaching_step.
    ``kinds=("swin", "swin")``: train_cross_pseudo_supervision_2D_ViT.py:213-241 -- two SwinUnet students, the same loop
    body (Dice on the other student's arg-max pseudo labels)."""
    from oracle.step import cross_teaching_step
    from oracle.swin import OracleSwinUnet
    from utils import losses as ref_losses, ramps as ref_ramps
    torch.manual_seed(0)
    C, L = cfg["num_classes"], cfg["labeled_bs"]
    nets = [OracleUNet2D(1, C) if k == "unet2d" else OracleSwinUnet(C) for k in kinds]
    models = [build_reference(k, 1, C) for k in kinds]
    sds = []
    for m, (onet, model) in enumerate(zip(nets, models)):
        sd = filler.fill_state_dict({f"m{m}." + k: v.clone() for k, v in model.state_dict().items()})
        sd = {k.split(".", 1)[1]: v for k, v in sd.items()}
        assert list(sd.keys()) == [s[0] for s in onet.spec()]
        model.load_state_dict(sd)
        model.train()
        sds.append(sd)
    set_reference_dropout(models[0], kinds[0], "off", None)
    set_reference_dropout(models[1], kinds[1], "off", None)
    volume, label, _ = make_inputs("swin", cfg)
    # ---- reference loop body ----
    opts = [torch.optim.SGD(m.parameters(), lr=cfg["base_lr"], momentum=0.9, weight_decay=0.0001) for m in models]
    lr_prev = cfg["base_lr"] * (1.0 - it / cfg["max_iterations"]) ** 0.9      # set after step it-1 (post-increment)
    for m, opt in enumerate(opts):
        for n, p in models[m].named_parameters():
            opt.state[p]["momentum_buffer"] = filler.uniform(p.shape, f"mom{m}." + n, -0.01, 0.01)
        for g in opt.param_groups:
            g["lr"] = lr_prev
    dice = ref_losses.DiceLoss(C)
    ce = torch.nn.CrossEntropyLoss()
    o1, o2 = models[0](volume), models[1](volume)
    s1, s2 = torch.softmax(o1, dim=1), torch.softmax(o2, dim=1)
    w = cfg["consistency"] * ref_ramps.sigmoid_rampup(it // 150, cfg["rampup"])
    loss1 = 0.5 * (ce(o1[:L], label[:L].long()) + dice(s1[:L], label[:L].unsqueeze(1)))
    loss2 = 0.5 * (ce(o2[:L], label[:L].long()) + dice(s2[:L], label[:L].unsqueeze(1)))
    p1 = torch.argmax(s1[L:].detach(), dim=1, keepdim=False)
    p2 = torch.argmax(s2[L:].detach(), dim=1, keepdim=False)
    ps1, ps2 = dice(s1[L:], p2.unsqueeze(1)), dice(s2[L:], p1.unsqueeze(1))
    m1, m2 = loss1 + w * ps1, loss2 + w * ps2
    for opt in opts:
        opt.zero_grad()
    (m1 + m2).backward()
    rgrads = [[p.grad.detach().clone() for p in m.parameters()] for m in models]
    for opt in opts:
        opt.step()
    # ---- oracle ----
    osd = [{k: v.clone() for k, v in sd.items()} for sd in sds]
    moms = [{n: filler.uniform(osd[m][n].shape, f"mom{m}." + n, -0.01, 0.01) for n in osd[m] if nets[m].is_param(n)}
            for m in range(2)]
    r = cross_teaching_step(nets[0], nets[1], osd[0], osd[1], moms[0], moms[1], volume, label, it, labeled_bs=L,
                            num_classes=C, base_lr=cfg["base_lr"], max_iterations=cfg["max_iterations"],
                            consistency=cfg["consistency"], rampup=cfg["rampup"], drop1="off", drop2="off")
    worst = 0.0
    for a, b, what in ((r["model1_loss"], float(m1), "model1_loss"), (r["model2_loss"], float(m2), "model2_loss"),
                       (r["lr"], lr_prev, "lr"), (r["logits1"], o1.detach(), "logits1"),
                       (r["logits2"], o2.detach(), "logits2")):
        worst = max(worst, rel_close(a, b, 1e-5, f"{name} {what}"))
    for m in range(2):
        ref_sd = models[m].state_dict()
        for (n, _), g in zip(models[m].named_parameters(), rgrads[m]):
            rel_close(r["grads"][m][n], g, 2e-4, f"{name} grad m{m} {n}")
            rel_close(osd[m][n], ref_sd[n], 1e-5, f"{name} post-SGD m{m} {n}")
    # fp64 reference for the gradient envelope
    m64 = [build_reference(k, 1, C).double() for k in kinds]
    for m in range(2):
        m64[m].load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sds[m].items()})
        m64[m].train()
    set_reference_dropout(m64[0], kinds[0], "off", None)
    set_reference_dropout(m64[1], kinds[1], "off", None)
    q1, q2 = m64[0](volume.double()), m64[1](volume.double())
    t1, t2 = torch.softmax(q1, 1), torch.softmax(q2, 1)
    l1 = 0.5 * (ce(q1[:L], label[:L].long()) + dice(t1[:L], label[:L].unsqueeze(1))) + \
        w * dice(t1[L:], torch.argmax(t2[L:].detach(), 1).unsqueeze(1))
    l2 = 0.5 * (ce(q2[:L], label[:L].long()) + dice(t2[:L], label[:L].unsqueeze(1))) + \
        w * dice(t2[L:], torch.argmax(t1[L:].detach(), 1).unsqueeze(1))
    (l1 + l2).backward()
    out = dict(meta=json.dumps(dict(name=name, kind="cross", cfg=cfg, iters=[it], drop_mode="off", kinds=list(kinds))))
    pre = f"it{it}_"
    out[pre + "model1_loss"], out[pre + "model2_loss"] = np.float64(float(m1)), np.float64(float(m2))
    out[pre + "loss1_ce_dice"] = np.float64(float(loss1))
    out[pre + "loss2_ce_dice"] = np.float64(float(loss2))
    out[pre + "pseudo1"], out[pre + "pseudo2"] = np.float64(float(ps1)), np.float64(float(ps2))
    out[pre + "consistency_weight"], out[pre + "lr"] = np.float64(w), np.float64(lr_prev)
    for m, o in enumerate((o1, o2)):
        for k, v in tensor_summary(o).items():
            out[pre + f"logits{m + 1}_{k}"] = np.asarray(v)
        g64 = [p.grad for p in m64[m].parameters()]
        out[pre + f"grad_norms{m + 1}"] = np.array([float(g.double().norm()) for g in rgrads[m]])
        out[pre + f"grad_norms64_{m + 1}"] = np.array([float(g.norm()) for g in g64])
        out[pre + f"grad_max64_{m + 1}"] = np.array([float(g.abs().max()) for g in g64])
        out[pre + f"grad_relerr32_{m + 1}"] = np.array(
            [float((a.double() - b).abs().max() / (b.abs().max() + 1e-300)) for a, b in zip(rgrads[m], g64)])
        sdm = models[m].state_dict()
        pn = [n for n, _ in models[m].named_parameters()]
        out[pre + f"param_abssum{m + 1}"] = np.array([float(sdm[n].double().abs().sum()) for n in pn])
    out["oracle_vs_reference_worst_rel"] = np.float64(worst)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"{name}: oracle vs reference worst rel err {worst:.2e}; wrote {name}.npz")


def run_cnnvit_case(name, cfg, it):
    """CNN student + Transformer student + EMA Transformer teacher (SURVEY s.8 row n2,
    train_cnn_meet_vit_2D.py:293-352) at 224x224: reference modules in the restated loop vs
    oracle.step.cnn_meet_vit_step."""
    from oracle.step import cnn_meet_vit_step
    from oracle.swin import OracleSwinUnet
    from utils import losses as ref_losses, ramps as ref_ramps
    torch.manual_seed(0)
    C, L = cfg["num_classes"], cfg["labeled_bs"]
    nets = [OracleUNet2D(1, C), OracleSwinUnet(C), OracleSwinUnet(C)]
    kinds = ["unet2d", "swin", "swin"]

    def build(dtype):
        ms, sds = [], []
        for m, kind in enumerate(kinds):
            model = build_reference(kind, 1, C)
            sd = filler.fill_state_dict({f"m{m}." + k: v.clone() for k, v in model.state_dict().items()})
            sd = {k.split(".", 1)[1]: v for k, v in sd.items()}
            assert list(sd.keys()) == [s[0] for s in nets[m].spec()]
            model.load_state_dict(sd)
            model = model.to(dtype)
            model.train()
            set_reference_dropout(model, kind, "off", None)
            ms.append(model)
            sds.append(sd)
        for p in ms[2].parameters():
            p.detach_()
        return ms, sds

    models, sds = build(torch.float32)
    volume, label, noise = make_inputs("swin", cfg)
    dice = ref_losses.DiceLoss(C)
    ce = torch.nn.CrossEntropyLoss()
    w = cfg["consistency"] * ref_ramps.linear_rampup(it // 150, cfg["rampup"])     # :322-323 (both weights)

    def loop_body(ms, vol, nz):
        """:293-337"""
        o1, o2 = ms[0](vol), ms[1](vol)
        s1, s2 = torch.softmax(o1, dim=1), torch.softmax(o2, dim=1)
        with torch.no_grad():
            eo = ms[2](vol[L:] + nz)
            es = torch.softmax(eo, dim=1)
        loss1 = 0.5 * (ce(o1[:L], label[:L].long()) + dice(s1[:L], label[:L].unsqueeze(1)))
        loss2 = 0.5 * (ce(o2[:L], label[:L].long()) + dice(s2[:L], label[:L].unsqueeze(1)))
        p1 = torch.argmax(s1[L:].detach(), dim=1, keepdim=False)
        p2 = torch.argmax(s2[L:].detach(), dim=1, keepdim=False)
        ps1, ps2 = dice(s1[L:], p2.unsqueeze(1)), dice(s2[L:], p1.unsqueeze(1))
        if it < 1000:
            c1 = c2 = 0.0
        else:
            c1 = torch.mean((s1[L:] - es) ** 2)
            c2 = torch.mean((s2[L:] - es) ** 2)
        m1 = loss1 + 7 * w * ps1 + w * c1
        m2 = loss2 + 7 * w * ps2 + w * c2
        return o1, o2, eo, loss1, loss2, ps1, ps2, c1, c2, m1, m2

    opts = [torch.optim.SGD(m.parameters(), lr=cfg["base_lr"], momentum=0.9, weight_decay=0.0001) for m in models[:2]]
    lr_prev = cfg["base_lr"] * (1.0 - (it - 1) / cfg["max_iterations"]) ** 0.9     # set after step it-1 (:347)
    for m, opt in enumerate(opts):
        for n, p in models[m].named_parameters():
            opt.state[p]["momentum_buffer"] = filler.uniform(p.shape, f"mom{m}." + n, -0.01, 0.01)
        for g in opt.param_groups:
            g["lr"] = lr_prev
    o1, o2, eo, loss1, loss2, ps1, ps2, c1, c2, m1, m2 = loop_body(models, volume, noise)
    for opt in opts:
        opt.zero_grad()
    (m1 + m2).backward()
    rgrads = [[p.grad.detach().clone() for p in m.parameters()] for m in models[:2]]
    for opt in opts:
        opt.step()
    alpha = min(1 - 1 / (it + 1), cfg["ema_decay"])                                # update_ema_variables :145-150
    for ema_param, param in zip(models[2].parameters(), models[1].parameters()):
        ema_param.data.mul_(alpha).add_(param.data, alpha=1 - alpha)
    # ---- oracle ----
    osd = [{k: v.clone() for k, v in sd.items()} for sd in sds]
    moms = [{n: filler.uniform(osd[m][n].shape, f"mom{m}." + n, -0.01, 0.01) for n in osd[m] if nets[m].is_param(n)}
            for m in range(2)]
    r = cnn_meet_vit_step(nets[0], nets[1], osd[0], osd[1], osd[2], moms[0], moms[1], volume, label, noise, it,
                          labeled_bs=L, num_classes=C, base_lr=cfg["base_lr"], max_iterations=cfg["max_iterations"],
                          ema_decay=cfg["ema_decay"], consistency=cfg["consistency"], rampup=cfg["rampup"],
                          drop1="off", drop2="off", drop_t="off")
    worst = 0.0
    for a, b, what in ((r["model1_loss"], float(m1), "model1_loss"), (r["model2_loss"], float(m2), "model2_loss"),
                       (r["lr"], lr_prev, "lr"), (r["ema_alpha"], alpha, "alpha"), (r["logits1"], o1.detach(), "logits1"),
                       (r["logits2"], o2.detach(), "logits2"), (r["teacher_logits"], eo.detach(), "teacher_logits"),
                       (r["parts"][0][3], float(c1), "cons1"), (r["parts"][1][3], float(c2), "cons2")):
        worst = max(worst, rel_close(a, b, 1e-5, f"{name} {what}"))
    for m in range(2):
        ref_sd = models[m].state_dict()
        for (n, _), g in zip(models[m].named_parameters(), rgrads[m]):
            rel_close(r["grads"][m][n], g, 2e-4, f"{name} grad m{m} {n}")
            rel_close(osd[m][n], ref_sd[n], 1e-5, f"{name} post-SGD m{m} {n}")
    tsd_ref = models[2].state_dict()
    for n, _ in models[2].named_parameters():
        rel_close(osd[2][n], tsd_ref[n], 1e-5, f"{name} post-EMA {n}")
    # fp64 reference for the gradient envelope
    m64, _ = build(torch.float64)
    r64 = loop_body(m64, volume.double(), noise.double())
    (r64[-2] + r64[-1]).backward()
    out = dict(meta=json.dumps(dict(name=name, kind="cnnvit", cfg=cfg, iters=[it], drop_mode="off")))
    pre = f"it{it}_"
    out[pre + "model1_loss"], out[pre + "model2_loss"] = np.float64(float(m1)), np.float64(float(m2))
    out[pre + "loss1_ce_dice"], out[pre + "loss2_ce_dice"] = np.float64(float(loss1)), np.float64(float(loss2))
    out[pre + "pseudo1"], out[pre + "pseudo2"] = np.float64(float(ps1)), np.float64(float(ps2))
    out[pre + "cons1"], out[pre + "cons2"] = np.float64(float(c1)), np.float64(float(c2))
    out[pre + "weight"], out[pre + "lr"], out[pre + "ema_alpha"] = np.float64(w), np.float64(lr_prev), np.float64(alpha)
    for k, v in tensor_summary(eo).items():
        out[pre + f"teacher_logits_{k}"] = np.asarray(v)
    pn_t = [n for n, _ in models[2].named_parameters()]
    out[pre + "teacher_abssum"] = np.array([float(tsd_ref[n].double().abs().sum()) for n in pn_t])
    for m, o in enumerate((o1, o2)):
        for k, v in tensor_summary(o).items():
            out[pre + f"logits{m + 1}_{k}"] = np.asarray(v)
        g64 = [p.grad for p in m64[m].parameters()]
        out[pre + f"grad_norms{m + 1}"] = np.array([float(g.double().norm()) for g in rgrads[m]])
        out[pre + f"grad_norms64_{m + 1}"] = np.array([float(g.norm()) for g in g64])
        out[pre + f"grad_max64_{m + 1}"] = np.array([float(g.abs().max()) for g in g64])
        out[pre + f"grad_relerr32_{m + 1}"] = np.array(
            [float((a.double() - b).abs().max() / (b.abs().max() + 1e-300)) for a, b in zip(rgrads[m], g64)])
        sdm = models[m].state_dict()
        pn = [n for n, _ in models[m].named_parameters()]
        out[pre + f"param_abssum{m + 1}"] = np.array([float(sdm[n].double().abs().sum()) for n in pn])
    out["oracle_vs_reference_worst_rel"] = np.float64(worst)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"{name}: oracle vs reference worst rel err {worst:.2e}; wrote {name}.npz")


def run_cps_case(name, kind, cfg, it):
    """Cross pseudo supervision between two CNN students of the same architecture (SURVEY s.8 row n2): reference
    modules in the restated loop of train_cross_pseudo_supervision_3D.py:149-185 / _2D.py:166-204 (CE pseudo
    supervision) vs oracle.step.cross_teaching_step(pseudo_ce=True)."""
    from oracle.step import cross_teaching_step
    from utils import losses as ref_losses, ramps as ref_ramps
    torch.manual_seed(0)
    C, L = cfg["num_classes"], cfg["labeled_bs"]
    mk = (lambda: OracleUNet2D(1, C)) if kind == "unet2d" else (lambda: OracleUNet3D(C, 1))
    nets = [mk(), mk()]
    models = [build_reference(kind, 1, C), build_reference(kind, 1, C)]
    sds = []
    for m, (onet, model) in enumerate(zip(nets, models)):
        sd = filler.fill_state_dict({f"m{m}." + k: v.clone() for k, v in model.state_dict().items()})
        sd = {k.split(".", 1)[1]: v for k, v in sd.items()}
        assert list(sd.keys()) == [s[0] for s in onet.spec()]
        model.load_state_dict(sd)
        model.train()
        set_reference_dropout(model, kind, "off", None)
        sds.append(sd)
    volume, label, _ = make_inputs(kind, cfg)
    opts = [torch.optim.SGD(m.parameters(), lr=cfg["base_lr"], momentum=0.9, weight_decay=0.0001) for m in models]
    lr_prev = cfg["base_lr"] * (1.0 - it / cfg["max_iterations"]) ** 0.9      # set after step it-1 (post-increment)
    for m, opt in enumerate(opts):
        for n, p in models[m].named_parameters():
            opt.state[p]["momentum_buffer"] = filler.uniform(p.shape, f"mom{m}." + n, -0.01, 0.01)
        for g in opt.param_groups:
            g["lr"] = lr_prev
    dice = ref_losses.DiceLoss(C)
    ce = torch.nn.CrossEntropyLoss()

    def loop_body(ms, vol):
        o1, o2 = ms[0](vol), ms[1](vol)
        s1, s2 = torch.softmax(o1, dim=1), torch.softmax(o2, dim=1)
        w = cfg["consistency"] * ref_ramps.sigmoid_rampup(it // 150, cfg["rampup"])
        loss1 = 0.5 * (ce(o1[:L], label[:L].long()) + dice(s1[:L], label[:L].unsqueeze(1)))
        loss2 = 0.5 * (ce(o2[:L], label[:L].long()) + dice(s2[:L], label[:L].unsqueeze(1)))
        p1 = torch.argmax(s1[L:].detach(), dim=1, keepdim=False)
        p2 = torch.argmax(s2[L:].detach(), dim=1, keepdim=False)
        ps1, ps2 = ce(o1[L:], p2), ce(o2[L:], p1)
        return o1, o2, loss1, loss2, ps1, ps2, w, loss1 + w * ps1, loss2 + w * ps2

    o1, o2, loss1, loss2, ps1, ps2, w, m1, m2 = loop_body(models, volume)
    for opt in opts:
        opt.zero_grad()
    (m1 + m2).backward()
    rgrads = [[p.grad.detach().clone() for p in m.parameters()] for m in models]
    for opt in opts:
        opt.step()
    osd = [{k: v.clone() for k, v in sd.items()} for sd in sds]
    moms = [{n: filler.uniform(osd[m][n].shape, f"mom{m}." + n, -0.01, 0.01) for n in osd[m] if nets[m].is_param(n)}
            for m in range(2)]
    r = cross_teaching_step(nets[0], nets[1], osd[0], osd[1], moms[0], moms[1], volume, label, it, labeled_bs=L,
                            num_classes=C, base_lr=cfg["base_lr"], max_iterations=cfg["max_iterations"],
                            consistency=cfg["consistency"], rampup=cfg["rampup"], drop1="off", drop2="off",
                            pseudo_ce=True)
    worst = 0.0
    for a, b, what in ((r["model1_loss"], float(m1), "model1_loss"), (r["model2_loss"], float(m2), "model2_loss"),
                       (r["lr"], lr_prev, "lr"), (r["logits1"], o1.detach(), "logits1"),
                       (r["logits2"], o2.detach(), "logits2")):
        worst = max(worst, rel_close(a, b, 1e-5, f"{name} {what}"))
    for m in range(2):
        ref_sd = models[m].state_dict()
        for (n, _), g in zip(models[m].named_parameters(), rgrads[m]):
            rel_close(r["grads"][m][n], g, 2e-4, f"{name} grad m{m} {n}")
            rel_close(osd[m][n], ref_sd[n], 1e-5, f"{name} post-SGD m{m} {n}")
    m64 = [build_reference(kind, 1, C).double(), build_reference(kind, 1, C).double()]
    for m in range(2):
        m64[m].load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sds[m].items()})
        m64[m].train()
        set_reference_dropout(m64[m], kind, "off", None)
    q = loop_body(m64, volume.double())
    (q[7] + q[8]).backward()
    out = dict(meta=json.dumps(dict(name=name, kind=kind, cfg=cfg, iters=[it], drop_mode="off", method="cps")))
    pre = f"it{it}_"
    out[pre + "model1_loss"], out[pre + "model2_loss"] = np.float64(float(m1)), np.float64(float(m2))
    out[pre + "loss1_ce_dice"], out[pre + "loss2_ce_dice"] = np.float64(float(loss1)), np.float64(float(loss2))
    out[pre + "pseudo1"], out[pre + "pseudo2"] = np.float64(float(ps1)), np.float64(float(ps2))
    out[pre + "consistency_weight"], out[pre + "lr"] = np.float64(w), np.float64(lr_prev)
    for m, o in enumerate((o1, o2)):
        for k, v in tensor_summary(o).items():
            out[pre + f"logits{m + 1}_{k}"] = np.asarray(v)
        g64 = [p.grad for p in m64[m].parameters()]
        out[pre + f"grad_norms{m + 1}"] = np.array([float(g.double().norm()) for g in rgrads[m]])
        out[pre + f"grad_norms64_{m + 1}"] = np.array([float(g.norm()) for g in g64])
        out[pre + f"grad_max64_{m + 1}"] = np.array([float(g.abs().max()) for g in g64])
        out[pre + f"grad_relerr32_{m + 1}"] = np.array(
            [float((a.double() - b).abs().max() / (b.abs().max() + 1e-300)) for a, b in zip(rgrads[m], g64)])
        sdm = models[m].state_dict()
        out[pre + f"param_abssum{m + 1}"] = np.array(
            [float(sdm[n].double().abs().sum()) for n, _ in models[m].named_parameters()])
    out["oracle_vs_reference_worst_rel"] = np.float64(worst)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"{name}: oracle vs reference worst rel err {worst:.2e}; wrote {name}.npz")


def reference_uamt_step(model, ema_model, optimizer, volume, label, noise, mc_noise, iter_num, cfg):
    """Loop body of train_uncertainty_aware_mean_teacher_3D.py:134-189 / _2D.py:146-201 around the reference
    modules (model, ema_model, utils.losses.DiceLoss / softmax_mse_loss, utils.ramps), with the noise tensors
    injected instead of torch.randn_like."""
    from utils import losses, ramps
    L, C = cfg["labeled_bs"], cfg["num_classes"]
    dice = losses.DiceLoss(C)
    ce = torch.nn.CrossEntropyLoss()
    unlabeled_volume_batch = volume[L:]
    ema_inputs = unlabeled_volume_batch + noise
    outputs = model(volume)
    outputs_soft = torch.softmax(outputs, dim=1)
    with torch.no_grad():
        ema_output = ema_model(ema_inputs)
    T = 8
    sp = list(unlabeled_volume_batch.shape[2:])
    volume_batch_r = unlabeled_volume_batch.repeat(2, *([1] * (volume.dim() - 1)))
    stride = volume_batch_r.shape[0] // 2
    preds = torch.zeros([stride * T, C] + sp, dtype=volume.dtype)
    for i in range(T // 2):
        with torch.no_grad():
            preds[2 * stride * i:2 * stride * (i + 1)] = ema_model(volume_batch_r + mc_noise[i])
    preds = torch.softmax(preds, dim=1)
    preds = preds.reshape([T, stride, C] + sp)
    preds = torch.mean(preds, dim=0)
    uncertainty = -1.0 * torch.sum(preds * torch.log(preds + 1e-6), dim=1, keepdim=True)
    loss_ce = ce(outputs[:L], label[:L].long())
    loss_dice = dice(outputs_soft[:L], label[:L].unsqueeze(1))
    supervised = 0.5 * (loss_dice + loss_ce)
    w = cfg["consistency"] * ramps.sigmoid_rampup(iter_num // 150, cfg["rampup"])
    consistency_dist = losses.softmax_mse_loss(outputs[L:], ema_output)
    threshold = (0.75 + 0.25 * ramps.sigmoid_rampup(iter_num, cfg["max_iterations"])) * np.log(2)
    mask = (uncertainty < threshold).float()
    cons = torch.sum(mask * consistency_dist) / (2 * torch.sum(mask) + 1e-16)
    loss = supervised + w * cons
    optimizer.zero_grad()
    loss.backward()
    grads = [p.grad.detach().clone() for p in model.parameters()]
    lr_used = optimizer.param_groups[0]["lr"]
    optimizer.step()
    alpha = min(1 - 1 / (iter_num + 1), cfg["ema_decay"])
    for ema_p, p in zip(ema_model.parameters(), model.parameters()):
        ema_p.data.mul_(alpha).add_(p.data, alpha=1 - alpha)
    return dict(loss=float(loss), loss_ce=float(loss_ce), loss_dice=float(loss_dice), consistency_loss=float(cons),
                consistency_weight=w, lr=lr_used, threshold=float(threshold), unmasked=float(mask.sum()),
                logits=outputs.detach(), teacher_logits=ema_output.detach(), grads=grads,
                margin=(uncertainty - threshold).abs().min().item())


def run_uamt_case(name, kind, cfg, it):
    """UA-MT: reference modules in the restated UA-MT loop vs oracle.step.uamt_step; dropout p := 0, noises injected."""
    from oracle.step import uamt_step
    torch.manual_seed(0)
    C, L = cfg["num_classes"], cfg["labeled_bs"]
    if kind == "swin":       # train_uncertainty_aware_mean_teacher_ViT_2D.py:183-245 == the 2-D loop on two SwinUnets
        from oracle.swin import OracleSwinUnet
        onet = OracleSwinUnet(C)
    else:
        onet = OracleUNet2D(1, C) if kind == "unet2d" else OracleUNet3D(C, 1)
    model, ema_model = build_reference(kind, 1, C), build_reference(kind, 1, C)
    for p in ema_model.parameters():
        p.detach_()
    sd0 = filler.fill_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    tsd0 = filler.fill_state_dict({"t." + k: v.clone() for k, v in model.state_dict().items()})
    tsd0 = {k[2:]: v for k, v in tsd0.items()}
    # the filler teacher predicts near-uniform probabilities (entropy > ln 2 everywhere = empty mask): sharpen its
    # output layer so that the entropy threshold splits the voxels.  Recorded in the fixture's cfg.
    head = {"unet2d": "decoder.out_conv.weight", "swin": "swin_unet.output.weight"}.get(kind, "final.weight")
    tsd0[head] = tsd0[head] * cfg["teacher_head_scale"]
    model.load_state_dict(sd0)
    ema_model.load_state_dict(tsd0)
    model.train(); ema_model.train()
    set_reference_dropout(model, kind, "off", None)
    set_reference_dropout(ema_model, kind, "off", None)
    volume, label, noise = make_inputs(kind, cfg)
    U = cfg["batch_size"] - L
    mc_noise = [filler.noise((2 * U, 1) + tuple(cfg["spatial"]), f"mc_noise{i}") for i in range(4)]
    optimizer = torch.optim.SGD(model.parameters(), lr=cfg["base_lr"], momentum=0.9, weight_decay=0.0001)
    for n, p in model.named_parameters():
        optimizer.state[p]["momentum_buffer"] = filler.uniform(p.shape, "mom." + n, -0.01, 0.01)
    lr_prev = cfg["base_lr"] * (1.0 - (it - 1) / cfg["max_iterations"]) ** 0.9
    for g in optimizer.param_groups:
        g["lr"] = lr_prev
    ref = reference_uamt_step(model, ema_model, optimizer, volume, label, noise, mc_noise, it, cfg)
    frac = ref["unmasked"] / (U * np.prod(cfg["spatial"]))
    print(f"{name}: threshold {ref['threshold']:.4f}, unmasked fraction {frac:.3f}, closest |u - thr| {ref['margin']:.2e}")
    assert 0.05 < frac < 0.95, "fixture must exercise both sides of the uncertainty mask"
    student = {k: v.clone() for k, v in sd0.items()}
    teacher = {k: v.clone() for k, v in tsd0.items()}
    mom = {n: filler.uniform(student[n].shape, "mom." + n, -0.01, 0.01) for n in student if onet.is_param(n)}
    orc = uamt_step(onet, student, teacher, mom, volume, label, noise, mc_noise, it, labeled_bs=L, num_classes=C,
                    base_lr=cfg["base_lr"], max_iterations=cfg["max_iterations"], ema_decay=cfg["ema_decay"],
                    consistency=cfg["consistency"], rampup=cfg["rampup"], drop_student="off", drop_teacher="off")
    worst = 0.0
    for k in ("loss", "loss_ce", "loss_dice", "consistency_loss", "consistency_weight", "lr", "threshold", "unmasked"):
        worst = max(worst, rel_close(orc[k], ref[k], 1e-5, f"{name} {k}"))
    worst = max(worst, rel_close(orc["logits"], ref["logits"], 1e-5, f"{name} logits"))
    worst = max(worst, rel_close(orc["teacher_logits"], ref["teacher_logits"], 1e-5, f"{name} teacher logits"))
    pnames = [n for n, _ in model.named_parameters()]
    for n, g in zip(pnames, ref["grads"]):
        rel_close(orc["grads"][n], g, 2e-4, f"{name} grad {n}")
    ref_sd, ref_tsd = model.state_dict(), ema_model.state_dict()
    for n in ref_sd:
        if n.endswith("num_batches_tracked"):
            assert int(ref_sd[n]) == int(student[n]) and int(ref_tsd[n]) == int(teacher[n])
            continue
        rel_close(student[n], ref_sd[n], 1e-5, f"{name} post-SGD {n}")
        rel_close(teacher[n], ref_tsd[n], 1e-5, f"{name} post-EMA {n}")
    # float64 run of the same loop: the reference's own fp32 rounding noise per gradient tensor
    m64, e64 = build_reference(kind, 1, C).double(), build_reference(kind, 1, C).double()
    m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()})
    e64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in tsd0.items()})
    m64.train(); e64.train()
    set_reference_dropout(m64, kind, "off", None)
    set_reference_dropout(e64, kind, "off", None)
    g64 = reference_uamt_step(m64, e64, torch.optim.SGD(m64.parameters(), lr=0.0), volume.double(), label,
                              noise.double(), [m.double() for m in mc_noise], it, cfg)["grads"]
    out = dict(meta=json.dumps(dict(name=name, kind=kind, cfg=cfg, iters=[it], drop_mode="off", method="uamt")))
    pre = f"it{it}_"
    for k in ("loss", "loss_ce", "loss_dice", "consistency_loss", "consistency_weight", "lr", "threshold", "unmasked"):
        out[pre + k] = np.float64(ref[k])
    for k, v in tensor_summary(ref["logits"]).items():
        out[pre + "logits_" + k] = np.asarray(v)
    for k, v in tensor_summary(ref["teacher_logits"]).items():
        out[pre + "teacher_logits_" + k] = np.asarray(v)
    out[pre + "grad_norms"] = np.array([float(g.double().norm()) for g in ref["grads"]])
    out[pre + "grad_norms64"] = np.array([float(g.norm()) for g in g64])
    out[pre + "grad_max64"] = np.array([float(g.abs().max()) for g in g64])
    out[pre + "grad_relerr32"] = np.array([float((a.double() - b).abs().max() / (b.abs().max() + 1e-300))
                                           for a, b in zip(ref["grads"], g64)])
    out[pre + "student_abssum"] = np.array([float(ref_sd[n].double().abs().sum()) for n in pnames])
    out[pre + "teacher_abssum"] = np.array([float(ref_tsd[n].double().abs().sum()) for n in pnames])
    bufs = [n for n in ref_sd if n.endswith("running_mean") or n.endswith("running_var")]
    if bufs:
        out[pre + "student_buf_sum"] = np.array([float(ref_sd[n].double().sum()) for n in bufs])
        out[pre + "teacher_buf_sum"] = np.array([float(ref_tsd[n].double().sum()) for n in bufs])
    out["oracle_vs_reference_worst_rel"] = np.float64(worst)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"{name}: oracle vs reference worst rel err {worst:.2e}; wrote {name}.npz")


CFG2D = dict(num_classes=4, base_lr=0.01, max_iterations=30000, ema_decay=0.99, consistency=0.1, rampup=200.0,
             cons_start_iter=1000)
CFG3D = dict(num_classes=2, base_lr=0.01, max_iterations=30000, ema_decay=0.99, consistency=0.1, rampup=200.0,
             cons_start_iter=0)


def load_from_checkpoints(key_shapes, which):
    """The two checkpoint formats SwinUnet.load_from reads (vision_transformer.py:54-89), filled in closed form so that
    the GPU test can rebuild them from the key / shape lists stored in the golden (no reference needed there).
    key_shapes: [(name, shape, dtype str)] of SwinTransformerSys.state_dict()."""
    def val(tag, k, shape, dt):
        if "int" in dt:
            return torch.full(shape, 7, dtype=getattr(torch, dt))
        return filler.uniform(shape, tag + ":" + k, -0.5, 0.5)
    if which == "split":
        # a checkpoint of a whole (DataParallel-wrapped) SwinUnet: "module.swin_unet." + key, head included
        return {"module.swin_unet." + k: val("split", k, sh, dt) for k, sh, dt in key_shapes}
    # an ImageNet Swin-T encoder checkpoint {"model": ...}: encoder keys only (depths [2, 2, 6, 2]: stage 2 carries four
    # blocks the network does not have), a 1000-class head, and one relative-position table of a window-12 model
    # (shape mismatch: dropped)
    enc = {}
    for k, sh, dt in key_shapes:
        if k.startswith(("patch_embed.", "layers.")) or k in ("norm.weight", "norm.bias"):
            enc[k] = val("enc", k, sh, dt)
    for k, sh, dt in key_shapes:
        m = re.match(r"layers\.2\.blocks\.([01])\.(.*)", k)
        if m:
            for extra in (2, 4):
                k2 = f"layers.2.blocks.{int(m.group(1)) + extra}.{m.group(2)}"
                enc[k2] = val("enc", k2, sh, dt)
    enc["head.weight"] = val("enc", "head.weight", (1000, 768), "float32")
    enc["head.bias"] = val("enc", "head.bias", (1000,), "float32")
    enc["layers.1.blocks.0.attn.relative_position_bias_table"] = val("enc", "rpb12", (23 * 23, 6), "float32")
    return {"model": enc}


def run_load_from_case(name):
    """SwinUnet.load_from on both checkpoint formats, by the REAL class: which entries of the network end up with
    which checkpoint tensor (prefix stripping, encoder -> decoder mirroring, head / shape-mismatch drops)."""
    import tempfile
    from types import SimpleNamespace as NS
    _install_timm_shim()
    out = {}
    model = build_reference("swin", 1, 4)
    ks = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in model.swin_unet.state_dict().items()]
    out["meta"] = json.dumps(dict(name=name, kind="load_from", key_shapes=ks))
    for which in ("split", "encoder"):
        model = build_reference("swin", 1, 4)
        model.load_state_dict(filler.fill_state_dict(model.state_dict()))
        before = {k: v.clone() for k, v in model.state_dict().items()}
        ck = load_from_checkpoints(ks, which)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "ck.pth")
            torch.save(ck, path)
            model.load_from(NS(MODEL=NS(PRETRAIN_CKPT=path)))
        after = model.state_dict()
        names = list(after.keys())
        changed = [bool((after[k] != before[k]).any()) for k in names]
        out[which + "_names"] = np.array(names)
        out[which + "_changed"] = np.array(changed)
        out[which + "_sum"] = np.array([float(after[k].double().sum()) for k in names])
        out[which + "_abssum"] = np.array([float(after[k].double().abs().sum()) for k in names])
        out[which + "_first"] = np.array([float(after[k].double().flatten()[0]) for k in names])
        out[which + "_last"] = np.array([float(after[k].double().flatten()[-1]) for k in names])
        print(f"{name}/{which}: {sum(changed)} of {len(names)} entries replaced by the checkpoint")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"wrote {name}.npz")


def main():
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    small2d = dict(CFG2D, batch_size=4, labeled_bs=2, spatial=[64, 64])
    # 3-D small case is 64^3 so that the deepest InstanceNorm still sees 4^3 voxels per channel
    # (at 32^3 it would normalise over 8 values, where every fp32 implementation is chaotic)
    small3d = dict(CFG3D, batch_size=2, labeled_bs=1, spatial=[64, 64, 64])
    cases = [
        ("unet2d_64_dropoff", "unet2d", small2d, [0, 1000, 1001], "off", True),
        ("unet2d_64_masks", "unet2d", small2d, [1500], "masks", False),
        # the transposed-convolution decoder: UpBlock(bilinear=False), nn.ConvTranspose2d(k=2, s=2) (unet.py:76-78)
        ("unet2d_deconv_64_masks", "unet2d_deconv", dict(small2d, flips=4), [1500], "masks", True),
        ("unet3d_64_dropoff", "unet3d", small3d, [0, 7], "off", True),
        ("unet3d_64_masks", "unet3d", small3d, [450], "masks", False),
        # V-Net (--model vnet): BatchNorm3d + Dropout3d, stride-2 / transposed convolutions; batch 2+2 so that
        # the batch statistics of both networks see two samples
        ("vnet_64_dropoff", "vnet", dict(CFG3D, batch_size=4, labeled_bs=2, spatial=[64, 64, 64]), [0, 7], "off",
         True),
        ("vnet_64_masks", "vnet", dict(CFG3D, batch_size=4, labeled_bs=2, spatial=[64, 64, 64]), [450], "masks",
         False),
        # the other blocks of vnet.py:15-22: conv + GroupNorm(16) + ReLU (the north-star's conv+GN+ReLU block),
        # conv + InstanceNorm3d + ReLU, conv + ReLU
        ("vnet_gn_64_dropoff", "vnet_groupnorm", dict(CFG3D, batch_size=2, labeled_bs=1, spatial=[64, 64, 64]), [0, 7],
         "off", True),
        # flips=4: the float64 gate's flip envelope (reference_grads64): this fixture has a 4^3-level ReLU within rounding of 0
        ("vnet_gn_64_masks", "vnet_groupnorm", dict(CFG3D, batch_size=2, labeled_bs=1, spatial=[64, 64, 64], flips=4),
         [450], "masks", False),
        ("vnet_in_64_dropoff", "vnet_instancenorm", dict(CFG3D, batch_size=2, labeled_bs=1, spatial=[64, 64, 64]), [7],
         "off", False),
        ("vnet_none_64_masks", "vnet_none", dict(CFG3D, batch_size=2, labeled_bs=1, spatial=[64, 64, 64]), [450],
         "masks", False),
        # BASELINE shapes: config 1 (2D 256^2, 4+4) and config 3 geometry at batch 1+1 (96^3)
        ("unet2d_256_cfg1", "unet2d", dict(CFG2D, batch_size=8, labeled_bs=4, spatial=[256, 256]), [1000], "off",
         False),
        ("unet3d_96_cfg3_b2", "unet3d", dict(CFG3D, batch_size=2, labeled_bs=1, spatial=[96, 96, 96]), [3], "off",
         False),
        # config 4 geometry (SwinUnet cannot shrink below 224 with window 7), batch 1+1
        ("swin_224_dropoff", "swin", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224]), [1000], "off", True),
        ("swin_224_masks", "swin", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224]), [1200], "masks",
         False),
        # SwinUnet input variants: 3-channel input (vision_transformer.py:48-50 passes it through un-repeated) and
        # the IMG_SIZE 256 / WINDOW_SIZE 8 configuration (config.py:194-195) that runs BASELINE config 5's 256 x 256
        ("swin_224_rgb", "swin", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224], in_channels=3), [1200],
         "off", False),
        ("swin_256_w8", "swin_w8", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[256, 256]), [1200], "off", True),
    ]
    for name, kind, cfg, iters, mode, ev in cases:
        if not only or name in only:
            run_case(name, kind, cfg, iters, mode, eval_logits=ev)
    # UA-MT (SURVEY s.8 row n1): 2-D UNet (BatchNorm: 5 teacher forwards update the running statistics) and
    # unet_3D; iterations late in the schedule so that the entropy threshold splits the voxels
    for uname, ukind, ucfg, uit in (
            ("uamt_unet2d_64", "unet2d", dict(small2d, max_iterations=3000, teacher_head_scale=40.0), 2500),
            ("uamt_unet3d_64", "unet3d", dict(small3d, max_iterations=3000, teacher_head_scale=40.0), 2500)):
        if not only or uname in only:
            sys.path.insert(0, REF)
            run_uamt_case(uname, ukind, ucfg, uit)
    # the ViT variant of UA-MT (train_uncertainty_aware_mean_teacher_ViT_2D.py): two SwinUnets at 224 x 224, batch 1+1
    if not only or "uamt_swin_224" in only:
        _install_timm_shim()
        sys.path.insert(0, REF)
        run_uamt_case("uamt_swin_224", "swin", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224],
                                                    max_iterations=3000, teacher_head_scale=40.0), 2500)
    # ... and of cross pseudo supervision (train_cross_pseudo_supervision_2D_ViT.py): two SwinUnet students
    if not only or "cps_vit_224" in only:
        _install_timm_shim()
        sys.path.insert(0, REF)
        run_cross_case("cps_vit_224", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224]), 1300,
                       kinds=("swin", "swin"))
    # cross pseudo supervision (SURVEY s.8 row n2), CNN scripts: CE pseudo-supervision between two students
    for cname, ckind, ccfg, cit in (("cps_unet2d_64", "unet2d", small2d, 1300), ("cps_unet3d_64", "unet3d", small3d, 460)):
        if not only or cname in only:
            sys.path.insert(0, REF)
            run_cps_case(cname, ckind, ccfg, cit)
    # config 5 geometry: cross teaching UNet <-> SwinUnet at 224x224, batch 1+1
    if not only or "cross_224" in only:
        _install_timm_shim()
        sys.path.insert(0, REF)
        run_cross_case("cross_224", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224]), 1300)
    # CNN student + Transformer student + EMA Transformer teacher (train_cnn_meet_vit_2D.py), same geometry;
    # iteration 3100: linear ramp 20/200, mean-teacher term on
    if not only or "cnnvit_224" in only:
        _install_timm_shim()
        sys.path.insert(0, REF)
        run_cnnvit_case("cnnvit_224", dict(CFG2D, batch_size=2, labeled_bs=1, spatial=[224, 224]), 3100)
    # SwinUnet.load_from (vision_transformer.py:54-89): both checkpoint formats through the real class
    if not only or "swin_load_from" in only:
        sys.path.insert(0, REF)
        run_load_from_case("swin_load_from")


if __name__ == "__main__":
    main()

"""SwinUnet (``ViT_seg``) on hand-written gfx950 kernels.

Drop-in for the reference's ``networks.vision_transformer.SwinUnet`` (code/networks/vision_transformer.py:24-52)
wrapping ``SwinTransformerSys`` (code/networks/swin_transformer_unet_skip_expand_decoder_sys.py:599-793):
same constructor (``config, img_size, num_classes, zero_head, vis``), ``forward(x[B,1|3,224,224])`` and the
same 238 state_dict entries (``swin_unet.*``, including the ``relative_position_index`` / ``attn_mask``
buffers) so checkpoints interchange.

Architecture as instantiated by the reference yaml (swin_tiny_patch4_window7_224_lite.yaml): embed 96,
depths [2,2,2,2] for encoder AND decoder (``DECODER_DEPTHS`` is never read), heads [3,6,12,24], window 7,
mlp x4, qkv bias, drop_path linspace(0, 0.2, 8), patch-expand decoder, 1x1 output conv without bias.
Single-channel inputs are repeated to 3 channels (vision_transformer.py:49-50) inside the im2col kernel; 3-channel
inputs are taken as they are (the other branch of :48-50).  ``MODEL.SWIN.WINDOW_SIZE`` 7 (``DATA.IMG_SIZE`` a
multiple of 224) or 8 (a multiple of 256: ``--opts DATA.IMG_SIZE 256 MODEL.SWIN.WINDOW_SIZE 8``, config.py:194-195 --
the way the reference runs SwinUnet on the 256 x 256 inputs of the cross-teaching configuration).
"""
import math
import re

import torch

from mis_hip.plan import HipNet
from mis_hip import swin_plan as sp

def _trunc_normal(*shape, std=0.02):
    t = torch.empty(*shape)
    torch.nn.init.trunc_normal_(t, std=std)
    return t


def _rel_pos_index(WS):
    coords = torch.stack(torch.meshgrid([torch.arange(WS), torch.arange(WS)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += WS - 1
    rel[:, :, 1] += WS - 1
    rel[:, :, 0] *= 2 * WS - 1
    return rel.sum(-1)


def _attn_mask(H, W, shift, WS):
    """The reference's attn_mask buffer (swin...sys.py:216-238); kept for state_dict parity only --
    the attention kernel derives the mask from the token coordinates."""
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for hs in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
        for ws_ in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
            img[:, hs, ws_, :] = cnt
            cnt += 1
    mw = img.view(1, H // WS, WS, W // WS, WS, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, WS * WS)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


class SwinUnet(HipNet):
    ndim_spatial = 2

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False):
        super().__init__()
        self.num_classes, self.zero_head, self.config = num_classes, zero_head, config
        sw = config.MODEL.SWIN
        self.img = config.DATA.IMG_SIZE       # the reference ignores the img_size argument too (:31)
        self.embed, self.depths, self.heads = sw.EMBED_DIM, list(sw.DEPTHS), list(sw.NUM_HEADS)
        self.in_chans, self.mlp_ratio = sw.IN_CHANS, sw.MLP_RATIO
        self.ws = WS = int(sw.WINDOW_SIZE)
        if sw.PATCH_SIZE != 4 or WS not in (7, 8) or sw.APE or not sw.PATCH_NORM or not sw.QKV_BIAS or \
                sw.QK_SCALE is not None or config.MODEL.DROP_RATE != 0.0 or self.embed // self.heads[0] != 32:
            raise NotImplementedError("HIP SwinUnet covers the configurations the reference instantiates "
                                      "(patch 4, window 7 or 8, head_dim 32, no APE, patch norm, drop_rate 0)")
        if self.img % (4 * WS * 2 ** (len(self.depths) - 1)):
            # same failure point as the reference: window_partition cannot view e.g. 256 with window 7
            raise RuntimeError(f"img_size {self.img} is not divisible by patch*window*2^stages "
                               f"(224 with window 7, 256 with window 8)")
        self.dpr = [x.item() for x in torch.linspace(0, config.MODEL.DROP_PATH_RATE, sum(self.depths))]
        self._declare_all()
        self._materialize()

    # ---- parameters / buffers in the reference's state_dict order ----
    def _declare_linear(self, name, out_f, in_f, bias=True):
        self._declare(name + ".weight", _trunc_normal(out_f, in_f))
        if bias:
            self._declare(name + ".bias", torch.zeros(out_f))

    def _declare_ln(self, name, C):
        self._declare(name + ".weight", torch.ones(C))
        self._declare(name + ".bias", torch.zeros(C))

    def _declare_block(self, p, dim, res, heads, shift):
        WS = self.ws
        if shift > 0:
            self._declare(p + ".attn_mask", _attn_mask(res, res, shift, WS), "buffer")
        self._declare_ln(p + ".norm1", dim)
        self._declare(p + ".attn.relative_position_bias_table", _trunc_normal((2 * WS - 1) ** 2, heads))
        self._declare(p + ".attn.relative_position_index", _rel_pos_index(WS), "buffer")
        self._declare_linear(p + ".attn.qkv", 3 * dim, dim)
        self._declare_linear(p + ".attn.proj", dim, dim)
        self._declare_ln(p + ".norm2", dim)
        self._declare_linear(p + ".mlp.fc1", int(dim * self.mlp_ratio), dim)
        self._declare_linear(p + ".mlp.fc2", dim, int(dim * self.mlp_ratio))

    def _shift(self, res, blk):
        return 0 if (blk % 2 == 0 or res <= self.ws) else self.ws // 2

    def _declare_all(self):
        E, nl = self.embed, len(self.depths)
        pr = self.img // 4
        w = torch.empty(E, self.in_chans, 4, 4)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(self.in_chans * 16)
        self._declare("swin_unet.patch_embed.proj.weight", w)
        self._declare("swin_unet.patch_embed.proj.bias", torch.empty(E).uniform_(-bound, bound))
        self._declare_ln("swin_unet.patch_embed.norm", E)
        for i in range(nl):
            dim, res = E * 2 ** i, pr // 2 ** i
            for b in range(self.depths[i]):
                self._declare_block(f"swin_unet.layers.{i}.blocks.{b}", dim, res, self.heads[i], self._shift(res, b))
            if i < nl - 1:
                self._declare_linear(f"swin_unet.layers.{i}.downsample.reduction", 2 * dim, 4 * dim, bias=False)
                self._declare_ln(f"swin_unet.layers.{i}.downsample.norm", 4 * dim)
        for i in range(nl):
            k = nl - 1 - i
            dim, res = E * 2 ** k, pr // 2 ** k
            if i == 0:
                self._declare_linear("swin_unet.layers_up.0.expand", 2 * dim, dim, bias=False)
                self._declare_ln("swin_unet.layers_up.0.norm", dim // 2)
            else:
                for b in range(self.depths[k]):
                    self._declare_block(f"swin_unet.layers_up.{i}.blocks.{b}", dim, res, self.heads[k],
                                        self._shift(res, b))
                if i < nl - 1:
                    self._declare_linear(f"swin_unet.layers_up.{i}.upsample.expand", 2 * dim, dim, bias=False)
                    self._declare_ln(f"swin_unet.layers_up.{i}.upsample.norm", dim // 2)
        for i in range(1, nl):
            dim = E * 2 ** (nl - 1 - i)
            self._declare_linear(f"swin_unet.concat_back_dim.{i}", dim, 2 * dim)
        self._declare_ln("swin_unet.norm", E * 2 ** (nl - 1))
        self._declare_ln("swin_unet.norm_up", E)
        self._declare_linear("swin_unet.up.expand", 16 * E, E, bias=False)
        self._declare_ln("swin_unet.up.norm", E)
        w = torch.empty(self.num_classes, E, 1, 1)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self._declare("swin_unet.output.weight", w)

    def load_from(self, config):
        """reference vision_transformer.py:54-89: loads MODEL.PRETRAIN_CKPT when given (the checkpoint is not
        part of the reference repository); 'none pretrain' otherwise."""
        path = config.MODEL.PRETRAIN_CKPT
        if path is None:
            print("none pretrain")
            return
        print("pretrained_path:{}".format(path))
        sd = torch.load(path, map_location="cuda")
        own = self.state_dict()
        if "model" not in sd:
            # :60-68: a checkpoint of a whole wrapped SwinUnet -- the reference strips a 17-character prefix
            # ("module.swin_unet.") and drops the segmentation head ("output")
            print("---start load pretrained modle by splitting---")
            full = {}
            for k, v in sd.items():
                k = k[17:]
                if "output" in k:
                    print("delete key:{}".format(k))
                    continue
                if "swin_unet." + k in own:
                    full["swin_unet." + k] = v
            self.load_state_dict(full, strict=False)
            return
        # :69-87: an ImageNet Swin encoder checkpoint {"model": {...}}; encoder stage weights are mirrored into the
        # decoder stages (layers.N -> layers_up.(3-N)); shape mismatches are dropped
        print("---start load pretrained modle of swin encoder---")
        sd = sd["model"]
        full = dict(sd)
        for k, v in sd.items():
            if "layers." in k:
                m = re.match(r"layers\.(\d)(.*)", k)      # the reference indexes k[7:8]: keys start with "layers.N"
                if m:
                    full["layers_up." + str(3 - int(m.group(1))) + m.group(2)] = v
        keep = {}
        for k, v in full.items():
            k2 = "swin_unet." + k
            if k2 in own:
                if own[k2].shape != v.shape:
                    print("delete:{};shape pretrain:{};shape model:{}".format(k, v.shape, own[k2].shape))
                    continue
                keep[k2] = v
        self.load_state_dict(keep, strict=False)

    # ---- layer graph ----
    def _new_plan(self, key):
        return sp.SwinPlan(self, key)

    def _block(self, plan, p, x, out, B, res, dim, heads, shift, dp):
        P = self.P
        rows, L = B * res * res, res * res
        n1 = plan.new(rows, dim)
        plan.add(sp.LayerNormOp(x, n1, P(p + ".norm1.weight"), P(p + ".norm1.bias")))
        qkv = plan.new(rows, 3 * dim)
        plan.add(sp.LinearOp(n1, qkv, P(p + ".attn.qkv.weight"), P(p + ".attn.qkv.bias")))
        att = plan.new(rows, dim)
        plan.add(sp.AttnOp(qkv, att, P(p + ".attn.relative_position_bias_table"), B, res, res, heads, shift,
                           window=self.ws))
        pr = plan.new(rows, dim)
        proj = plan.add(sp.LinearOp(att, pr, P(p + ".attn.proj.weight"), P(p + ".attn.proj.bias")))
        x1 = plan.new(rows, dim)
        proj.res = plan.add(sp.ResidualOp(x, pr, x1, L, dp, plan.next_site()))    # fused into proj's epilogue
        n2 = plan.new(rows, dim)
        plan.add(sp.LayerNormOp(x1, n2, P(p + ".norm2.weight"), P(p + ".norm2.bias")))
        hdim = int(dim * self.mlp_ratio)
        h = plan.new(rows, hdim)
        fc1 = plan.add(sp.LinearOp(n2, h, P(p + ".mlp.fc1.weight"), P(p + ".mlp.fc1.bias")))
        hg = plan.new(rows, hdim)
        fc1.gelu = gelu = plan.add(sp.GeluOp(h, hg))                               # GELU in fc1's epilogue
        m = plan.new(rows, dim)
        fc2 = plan.add(sp.LinearOp(hg, m, P(p + ".mlp.fc2.weight"), P(p + ".mlp.fc2.bias")))
        fc2.dx_gelu = gelu                                                         # gelu' in fc2's dX epilogue
        x2 = out if out is not None else plan.new(rows, dim)
        fc2.res = plan.add(sp.ResidualOp(x1, m, x2, L, dp, plan.next_site()))      # residual add in fc2's epilogue
        return x2

    def _expand(self, plan, p, x, out, B, res, dim, P_, norm=True):
        """PatchExpand (P_=2: dim -> 2dim -> shuffle -> dim/2) / FinalPatchExpand_X4 (P_=4: dim -> 16dim -> dim);
        ``norm=False``: the caller applies the LayerNorm (fused with the output head)."""
        rows = B * res * res
        cout = (2 * dim) if P_ == 2 else 16 * dim
        c = cout // (P_ * P_)
        e = plan.new(rows, cout)               # token-major expansion: only its gradient is ever materialised
        sh = plan.new(rows * P_ * P_, c)
        plan.add(sp.ExpandLinearOp(x, e, sh, self.P(p + ".expand.weight"), (B, res, res, c, P_)))
        if not norm:
            return sh
        y = out if out is not None else plan.new(rows * P_ * P_, c)
        plan.add(sp.LayerNormOp(sh, y, self.P(p + ".norm.weight"), self.P(p + ".norm.bias")))
        return y

    def _build(self, plan):
        N, C, D, H, W = plan.in_shape
        # the reference asserts this in PatchEmbed.forward (swin...sys.py:583-584)
        assert D == 1 and H == self.img and W == self.img, \
            f"Input image size ({H}*{W}) doesn't match model ({self.img}*{self.img})."
        if C != 1 and C != self.in_chans:
            raise RuntimeError(f"input must have 1 channel (repeated to {self.in_chans} inside the patch embedding, "
                               f"vision_transformer.py:49-50) or {self.in_chans}; got {C}")
        E, nl, B = self.embed, len(self.depths), N
        pr = self.img // 4
        Pn = self.P
        # concat buffers of the decoder stages i = 1..nl-1: [x_up | skip]
        cat = {}
        for i in range(1, nl):
            k = nl - 1 - i
            cat[k] = plan.new(B * (pr // 2 ** k) ** 2, 2 * E * 2 ** k)
        cols = plan.new(B * pr * pr, self.in_chans * 16)
        plan.add(sp.Im2colOp(plan, cols, self.in_chans))
        e0 = plan.new(B * pr * pr, E)
        plan.add(sp.LinearOp(cols, e0, Pn("swin_unet.patch_embed.proj.weight"), Pn("swin_unet.patch_embed.proj.bias"),
                             need_dx=False))
        x = plan.cols(cat[0], E, E)                       # x_downsample[0]
        plan.add(sp.LayerNormOp(e0, x, Pn("swin_unet.patch_embed.norm.weight"), Pn("swin_unet.patch_embed.norm.bias")))
        di = 0
        for i in range(nl):
            dim, res = E * 2 ** i, pr // 2 ** i
            for b in range(self.depths[i]):
                x = self._block(plan, f"swin_unet.layers.{i}.blocks.{b}", x, None, B, res, dim, self.heads[i],
                                self._shift(res, b), self.dpr[di])
                di += 1
            if i < nl - 1:                               # PatchMerging (:325-346)
                p = f"swin_unet.layers.{i}.downsample"
                g = plan.new(B * (res // 2) ** 2, 4 * dim)
                plan.add(sp.RearrangeOp(x, g, B, res, res, dim, 2, 0))
                n = plan.new(B * (res // 2) ** 2, 4 * dim)
                plan.add(sp.LayerNormOp(g, n, Pn(p + ".norm.weight"), Pn(p + ".norm.bias")))
                nxt = plan.cols(cat[i + 1], 2 * dim, 2 * dim) if (i + 1) in cat else plan.new(B * (res // 2) ** 2, 2 * dim)
                plan.add(sp.LinearOp(n, nxt, Pn(p + ".reduction.weight"), None))
                x = nxt
        dimb, resb = E * 2 ** (nl - 1), pr // 2 ** (nl - 1)
        xn = plan.new(B * resb * resb, dimb)
        plan.add(sp.LayerNormOp(x, xn, Pn("swin_unet.norm.weight"), Pn("swin_unet.norm.bias")))
        # decoder
        k = nl - 2
        x = self._expand(plan, "swin_unet.layers_up.0", xn, plan.cols(cat[k], 0, dimb // 2), B, resb, dimb, 2)
        for i in range(1, nl):
            k = nl - 1 - i
            dim, res = E * 2 ** k, pr // 2 ** k
            c = plan.new(B * res * res, dim)
            plan.add(sp.LinearOp(cat[k], c, Pn(f"swin_unet.concat_back_dim.{i}.weight"),
                                 Pn(f"swin_unet.concat_back_dim.{i}.bias")))
            x = c
            base = sum(self.depths[:k])
            for b in range(self.depths[k]):
                x = self._block(plan, f"swin_unet.layers_up.{i}.blocks.{b}", x, None, B, res, dim, self.heads[k],
                                self._shift(res, b), self.dpr[base + b])
            if i < nl - 1:
                x = self._expand(plan, f"swin_unet.layers_up.{i}.upsample", x, plan.cols(cat[k - 1], 0, dim // 2), B,
                                 res, dim, 2)
        xu = plan.new(B * pr * pr, E)
        plan.add(sp.LayerNormOp(x, xu, Pn("swin_unet.norm_up.weight"), Pn("swin_unet.norm_up.bias")))
        plan.out = sp.new_logits(B, self.num_classes, H, W)
        if sp.LnHeadOp.eligible(E, self.num_classes):      # up.norm + output in one pass over the 16x expanded tokens
            sh = self._expand(plan, "swin_unet.up", xu, None, B, pr, E, 4, norm=False)
            expand = plan.ops[-1]                           # the ExpandLinearOp that wrote `sh` through the pixel shuffle
            head = sp.LnHeadOp(sh, Pn("swin_unet.up.norm.weight"), Pn("swin_unet.up.norm.bias"),
                               Pn("swin_unet.output.weight"), plan.out)
            head.expand = expand if isinstance(expand, sp.ExpandLinearOp) else None
            if head.expand is not None:
                head.expand.ln_head = head
            plan.add(head)
        else:
            xf = self._expand(plan, "swin_unet.up", xu, None, B, pr, E, 4)
            plan.add(sp.HeadOp(xf, Pn("swin_unet.output.weight"), plan.out))


ViT_seg = SwinUnet

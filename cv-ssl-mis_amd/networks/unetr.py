"""UNETR (``--model unetr``) on hand-written gfx950 kernels.

Drop-in for the network ``net_factory_3d('unetr')`` builds (reference code/networks/net_factory_3d.py:23-36 ->
code/networks/unetr.py:22-230): same constructor arguments, ``forward(x[N,1,96,96,96]) -> logits[N,C,96,96,96]``.

PARITY UNPINNED.  The reference assembles UNETR from MONAI blocks; MONAI is an un-vendored, un-versioned dependency of
the reference and is absent from the build image, so the arithmetic below follows the published MONAI blocks
(restated in oracle/unetr.py, which the GPU tests compare against) and the state_dict keys follow MONAI 0.8-era
module names (``vit.patch_embedding.*``, ``vit.blocks.N.{mlp.linear1,mlp.linear2,norm1,attn.out_proj,attn.qkv,norm2}``,
``encoderK.*``, ``decoderK.*``, ``out.conv.conv.*``); neither could be checked against the reference itself.

Execution: one static plan (mis_hip.plan.Plan) mixing the token-major ops of the ViT encoder (patch gather, Linear /
LayerNorm / GELU / residual on the SwinUnet kernels, full 216-token attention in csrc/attention_full.hip) with the
NCDHW conv ops of the decoder (3x3x3 / 1x1x1 MFMA convs, InstanceNorm + LeakyReLU, k2s2 transposed convs as 1x1x1 conv
+ depth-to-space, skips written straight into the concat buffers); ``proj_feat`` is a per-sample transpose.
"""
import math

import torch

from mis_hip import swin_plan as sp
from mis_hip.plan import HipNet


def _trunc_normal(*shape, std=0.02):
    t = torch.empty(*shape)
    torch.nn.init.trunc_normal_(t, std=std)
    return t


def _conv_default(*shape):
    """torch's default (transposed) conv init: kaiming_uniform(a=sqrt(5))"""
    w = torch.empty(*shape)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    return w


class UNETR(HipNet):
    ndim_spatial = 3

    def __init__(self, in_channels, out_channels, img_size, feature_size=16, hidden_size=768, mlp_dim=3072,
                 num_heads=12, pos_embed="perceptron", norm_name="instance", conv_block=False, res_block=True,
                 dropout_rate=0.0):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")                      # unetr.py:70-71
        if hidden_size % num_heads != 0:
            raise AssertionError("hidden size should be divisible by num_heads.")                # :73-74
        if pos_embed not in ["conv", "perceptron"]:
            raise KeyError(f"Position embedding layer of type {pos_embed} is not supported.")    # :76-77
        if (in_channels != 1 or pos_embed != "perceptron" or norm_name != "instance" or not res_block or
                dropout_rate != 0.0 or hidden_size // num_heads != 64 or any(s % 16 for s in img_size)):
            raise NotImplementedError("the HIP UNETR covers the configuration net_factory_3d builds: in_channels 1, "
                                      "'perceptron' embedding, instance norm, res_block, dropout 0, head_dim 64")
        self.in_channels, self.n_classes, self.img = in_channels, out_channels, tuple(img_size)
        self.fs, self.hid, self.mlp, self.heads, self.num_layers = feature_size, hidden_size, mlp_dim, num_heads, 12
        self.patch_size = (16, 16, 16)
        self.feat_size = tuple(s // 16 for s in self.img)
        self.L = self.feat_size[0] * self.feat_size[1] * self.feat_size[2]
        if self.L > 256:
            raise NotImplementedError("attention_full.hip holds one sample's tokens per workgroup: at most 256 patches")
        self.transposed_convs = set()
        H, f = hidden_size, feature_size
        D = self._declare
        D("vit.patch_embedding.position_embeddings", _trunc_normal(1, self.L, H))
        D("vit.patch_embedding.cls_token", torch.zeros(1, 1, H))
        D("vit.patch_embedding.patch_embeddings.1.weight", _trunc_normal(H, 4096))
        D("vit.patch_embedding.patch_embeddings.1.bias", torch.zeros(H))
        for i in range(self.num_layers):
            p = f"vit.blocks.{i}"
            D(p + ".mlp.linear1.weight", _trunc_normal(mlp_dim, H)); D(p + ".mlp.linear1.bias", torch.zeros(mlp_dim))
            D(p + ".mlp.linear2.weight", _trunc_normal(H, mlp_dim)); D(p + ".mlp.linear2.bias", torch.zeros(H))
            D(p + ".norm1.weight", torch.ones(H)); D(p + ".norm1.bias", torch.zeros(H))
            D(p + ".attn.out_proj.weight", _trunc_normal(H, H)); D(p + ".attn.out_proj.bias", torch.zeros(H))
            D(p + ".attn.qkv.weight", _trunc_normal(3 * H, H))
            D(p + ".norm2.weight", torch.ones(H)); D(p + ".norm2.bias", torch.zeros(H))
        D("vit.norm.weight", torch.ones(H)); D("vit.norm.bias", torch.zeros(H))
        self._declare_res("encoder1.layer", 1, f)
        self.enc_cfg = (("encoder2", 2 * f, 2), ("encoder3", 4 * f, 1), ("encoder4", 8 * f, 0))
        for name, cout, nl in self.enc_cfg:
            self._declare_up(name + ".transp_conv_init.conv.weight", H, cout)
            for b in range(nl):
                self._declare_up(f"{name}.blocks.{b}.0.conv.weight", cout, cout)
                self._declare_res(f"{name}.blocks.{b}.1", cout, cout)
        self.dec_cfg = (("decoder5", H, 8 * f), ("decoder4", 8 * f, 4 * f), ("decoder3", 4 * f, 2 * f),
                        ("decoder2", 2 * f, f))
        for name, cin, cout in self.dec_cfg:
            self._declare_up(name + ".transp_conv.conv.weight", cin, cout)
            self._declare_res(name + ".conv_block", 2 * cout, cout)
        D("out.conv.conv.weight", _conv_default(out_channels, f, 1, 1, 1))
        bound = 1.0 / math.sqrt(f)
        D("out.conv.conv.bias", torch.empty(out_channels).uniform_(-bound, bound))
        self._materialize()

    def _declare_up(self, name, cin, cout):
        self._declare(name, _conv_default(cin, cout, 2, 2, 2))        # ConvTranspose3d weight layout [Cin][Cout][2][2][2]
        self.transposed_convs.add(name)

    def _declare_res(self, p, cin, cout):
        self._declare(p + ".conv1.conv.weight", _conv_default(cout, cin, 3, 3, 3))
        self._declare(p + ".conv2.conv.weight", _conv_default(cout, cout, 3, 3, 3))
        if cin != cout:
            self._declare(p + ".conv3.conv.weight", _conv_default(cout, cin, 1, 1, 1))

    # ---- layer graph ----
    def _resblock(self, plan, p, x, cin, cout, spatial, out, need_dx=True):
        """MONAI UnetResBlock: conv3-IN-lrelu-conv3-IN (+ conv1-IN shortcut when cin != cout) -> add -> lrelu."""
        P = self.P
        res = x
        if cin != cout:      # the shortcut first: conv2 stays adjacent to its normalisation (statistics from its epilogue)
            t3 = plan.new(cout, spatial)
            plan.conv(x, t3, P(p + ".conv3.conv.weight"), None, (1, 1, 1), need_dx=need_dx, bias_grad=False)
            res = plan.norm_act(t3, plan.new(cout, spatial), per_sample=True, slope=1.0)
        t1 = plan.new(cout, spatial)
        plan.conv(x, t1, P(p + ".conv1.conv.weight"), None, (3, 3, 3), need_dx=need_dx, bias_grad=False)
        n1 = plan.norm_act(t1, plan.new(cout, spatial), per_sample=True, slope=0.01)
        t2 = plan.new(cout, spatial)
        plan.conv(n1, t2, P(p + ".conv2.conv.weight"), None, (3, 3, 3), bias_grad=False)
        if plan.can_norm_res_act(t2):       # IN(t2) + shortcut -> LeakyReLU in one pass over t2 and the shortcut
            return plan.norm_res_act(t2, res, out, per_sample=True, slope=0.01)
        n2 = plan.norm_act(t2, plan.new(cout, spatial), per_sample=True, slope=1.0)      # IN only (slope 1 = identity)
        s = plan.add(n2, res, plan.new(cout, spatial))
        return plan.norm_act(s, out, per_sample=False, slope=0.01, no_norm=True)          # LeakyReLU only

    def _build(self, plan):
        N, C, Dd, Hh, Ww = plan.in_shape
        if C != 1 or (Dd, Hh, Ww) != self.img:
            raise RuntimeError(f"UNETR input must be [N,1,{self.img[0]},{self.img[1]},{self.img[2]}]; got {plan.in_shape}")
        P, H, f, L, B = self.P, self.hid, self.fs, self.L, N
        rows = B * L

        def tok(C_):
            a = sp.TAct(rows, C_)
            plan.acts.append(a)
            return a

        def add(op):
            plan.ops.append(op)
            return op

        # ---- ViT encoder (token-major) ----
        cols = tok(4096)
        add(sp.Patch3dOp(plan, cols))
        e0 = tok(H)
        add(sp.LinearOp(cols, e0, P("vit.patch_embedding.patch_embeddings.1.weight"),
                        P("vit.patch_embedding.patch_embeddings.1.bias"), need_dx=False))
        x = tok(H)
        add(sp.PosAddOp(e0, P("vit.patch_embedding.position_embeddings"), x, L))
        hidden = []
        for i in range(self.num_layers):
            p = f"vit.blocks.{i}"
            n1 = tok(H)
            add(sp.LayerNormOp(x, n1, P(p + ".norm1.weight"), P(p + ".norm1.bias")))
            qkv = tok(3 * H)
            add(sp.LinearOp(n1, qkv, P(p + ".attn.qkv.weight"), None))
            att = tok(H)
            add(sp.FullAttnOp(qkv, att, B, L, self.heads))
            pr = tok(H)
            proj = add(sp.LinearOp(att, pr, P(p + ".attn.out_proj.weight"), P(p + ".attn.out_proj.bias")))
            x1 = tok(H)
            proj.res = add(sp.ResidualOp(x, pr, x1, L, 0.0, 2 * i))
            n2 = tok(H)
            add(sp.LayerNormOp(x1, n2, P(p + ".norm2.weight"), P(p + ".norm2.bias")))
            h = tok(self.mlp)
            fc1 = add(sp.LinearOp(n2, h, P(p + ".mlp.linear1.weight"), P(p + ".mlp.linear1.bias")))
            hg = tok(self.mlp)
            fc1.gelu = gelu = add(sp.GeluOp(h, hg))
            m = tok(H)
            fc2 = add(sp.LinearOp(hg, m, P(p + ".mlp.linear2.weight"), P(p + ".mlp.linear2.bias")))
            fc2.dx_gelu = gelu
            x2 = tok(H)
            fc2.res = add(sp.ResidualOp(x1, m, x2, L, 0.0, 2 * i + 1))
            x = x2
            hidden.append(x)
        xn = tok(H)
        add(sp.LayerNormOp(x, xn, P("vit.norm.weight"), P("vit.norm.bias")))

        # ---- conv decoder (NCDHW); skips are written straight into the decoders' concat buffers [up | skip] ----
        sp0 = self.feat_size
        scale = lambda s, k: tuple(v * k for v in s)
        cat = {name: plan.new(2 * cout, scale(sp0, 2 ** (i + 1))) for i, (name, _, cout) in enumerate(self.dec_cfg)}
        skip_of = {"decoder5": "encoder4", "decoder4": "encoder3", "decoder3": "encoder2", "decoder2": "encoder1"}
        skip_view = {skip_of[n]: plan.view(cat[n], cout, cout) for n, _, cout in self.dec_cfg}

        self._resblock(plan, "encoder1.layer", plan.inp, 1, f, self.img, skip_view["encoder1"], need_dx=False)
        for (name, cout, nl), hs in zip(self.enc_cfg, (3, 6, 9)):          # unetr.py:216-221
            v = plan.new(H, sp0)
            add(sp.TokToVolOp(hidden[hs], v, B, L))
            s1 = scale(sp0, 2)
            y = skip_view[name] if nl == 0 else plan.new(cout, s1)
            plan.up_conv(v, y, P(name + ".transp_conv_init.conv.weight"), None)
            cur, cs = y, s1
            for b in range(nl):
                cs = scale(cs, 2)
                u = plan.new(cout, cs)
                plan.up_conv(cur, u, P(f"{name}.blocks.{b}.0.conv.weight"), None)
                cur = self._resblock(plan, f"{name}.blocks.{b}.1", u, cout, cout, cs,
                                     skip_view[name] if b == nl - 1 else plan.new(cout, cs))
        v = plan.new(H, sp0)
        add(sp.TokToVolOp(xn, v, B, L))                                   # dec4 = proj_feat(x)  (:222)
        cur, cs = v, sp0
        for name, cin, cout in self.dec_cfg:                              # decoder5, 4, 3, 2  (:223-226)
            cs = scale(cs, 2)
            plan.up_conv(cur, plan.view(cat[name], 0, cout), P(name + ".transp_conv.conv.weight"), None)
            cur = self._resblock(plan, name + ".conv_block", cat[name], 2 * cout, cout, cs, plan.new(cout, cs))
        plan.out = plan.new(self.n_classes, self.img)
        plan.conv(cur, plan.out, P("out.conv.conv.weight"), P("out.conv.conv.bias"), (1, 1, 1), bias_grad=True)

    def load_from(self, weights):
        """reference unetr.py:188-212: copy a pre-trained ViT ('module.transformer.*' keys) into ``self.vit``."""
        sd = weights["state_dict"]
        own = dict(self.named_parameters())
        pre = "module.transformer."
        mapping = {"vit.patch_embedding.position_embeddings": pre + "patch_embedding.position_embeddings_3d",
                   "vit.patch_embedding.cls_token": pre + "patch_embedding.cls_token",
                   "vit.patch_embedding.patch_embeddings.1.weight": pre + "patch_embedding.patch_embeddings.1.weight",
                   "vit.patch_embedding.patch_embeddings.1.bias": pre + "patch_embedding.patch_embeddings.1.bias",
                   "vit.norm.weight": pre + "norm.weight", "vit.norm.bias": pre + "norm.bias"}
        for i in range(self.num_layers):
            for leaf in ("mlp.linear1.weight", "mlp.linear1.bias", "mlp.linear2.weight", "mlp.linear2.bias", "norm1.weight",
                         "norm1.bias", "attn.out_proj.weight", "attn.out_proj.bias", "attn.qkv.weight", "norm2.weight",
                         "norm2.bias"):
                mapping[f"vit.blocks.{i}.{leaf}"] = f"{pre}blocks.{i}.{leaf}"
        with torch.no_grad():
            for k, src in mapping.items():
                own[k].copy_(sd[src])

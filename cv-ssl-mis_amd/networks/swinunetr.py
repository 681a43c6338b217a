"""SwinUNETR (``--model swinunetr``) on hand-written gfx950 kernels.

Drop-in for the network ``net_factory_3d('swinunetr')`` builds (reference code/networks/net_factory_3d.py:7,37-38:
``monai.networks.nets.SwinUNETR(img_size=(64, 64, 64), in_channels=in_chns, out_channels=class_num, feature_size=48)``):
same constructor arguments, ``forward(x[N,1,D,H,W]) -> logits[N,C,D,H,W]`` for D, H, W multiples of 32.

PARITY UNPINNED.  The reference repository holds no source of this network -- it is imported from MONAI, an un-vendored,
un-versioned dependency that is absent from the build image.  The arithmetic follows the published MONAI 1.x
``networks/nets/swin_unetr.py`` with the defaults the reference's call relies on (depths (2,2,2,2), heads (3,6,12,24),
window 7, patch 2, mlp_ratio 4, qkv_bias, instance norm, all dropout rates 0, normalize=True, downsample="merging"),
restated in oracle/swinunetr.py, which the GPU tests compare against; the state_dict keys follow MONAI's module names
(``swinViT.patch_embed.proj``, ``swinViT.layersK.0.blocks.B.{norm1,attn.{relative_position_bias_table,qkv,proj},norm2,
mlp.{linear1,linear2}}``, ``swinViT.layersK.0.downsample.{reduction,norm}``, ``encoderK.layer.*``, ``decoderK.{transp_conv,
conv_block}.*``, ``out.conv.conv.*``).  Neither could be checked against the reference itself.

Execution: one static plan (mis_hip.plan.Plan) mixing the token-major ops of the Swin encoder -- LayerNorm / Linear / GELU
/ residual on the SwinUnet kernels, padded + shifted 3-D window gather and scatter, 343-token window attention with heads
of 16 channels and the v0.9 patch merging in csrc/swin3d.hip -- with the NCDHW conv ops of the decoder (UNETR's res / up
blocks: 3x3x3 Winograd / MFMA convs, InstanceNorm + LeakyReLU, k2s2 transposed convs, skips written straight into the
concat buffers).  The five hidden states reach the conv side through an affine-free channel LayerNorm (``proj_out``) and a
per-sample transpose.
"""
import math

import torch

from mis_hip import swin_plan as sp
from mis_hip.plan import DownConvOp, HipNet
from networks.unetr import UNETR, _conv_default, _trunc_normal

WS = 7


def window_geometry(dims, ws=WS):
    """MONAI get_window_size: window clipped (and shift dropped) per axis when the volume is not larger than it."""
    return tuple(d if d <= ws else ws for d in dims), tuple(0 if d <= ws else ws // 2 for d in dims)


def region_ids(pdims, win, shift):
    """MONAI compute_mask's region image, window-partitioned: int32 [nW, n]."""
    img = torch.zeros(pdims, dtype=torch.int32)
    cnt = 0
    for dsl in (slice(-win[0]), slice(-win[0], -shift[0]), slice(-shift[0], None)):
        for hsl in (slice(-win[1]), slice(-win[1], -shift[1]), slice(-shift[1], None)):
            for wsl in (slice(-win[2]), slice(-win[2], -shift[2]), slice(-shift[2], None)):
                img[dsl, hsl, wsl] = cnt
                cnt += 1
    d, h, w = pdims
    x = img.view(d // win[0], win[0], h // win[1], win[1], w // win[2], win[2])
    return x.permute(0, 2, 4, 1, 3, 5).contiguous().view(-1, win[0] * win[1] * win[2])


class SwinUNETR(HipNet):
    ndim_spatial = 3

    def __init__(self, img_size, in_channels, out_channels, depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24), feature_size=24,
                 norm_name="instance", drop_rate=0.0, attn_drop_rate=0.0, dropout_path_rate=0.0, normalize=True,
                 use_checkpoint=False, spatial_dims=3, downsample="merging", use_v2=False):
        super().__init__()
        for r, nm in ((drop_rate, "dropout"), (attn_drop_rate, "attention dropout"), (dropout_path_rate, "drop path")):
            if not (0 <= r <= 1):
                raise ValueError(f"{nm} rate should be between 0 and 1.")             # MONAI swin_unetr.py __init__
        if feature_size % 12 != 0:
            raise ValueError("feature_size should be divisible by 12.")
        if any(int(s) % 32 for s in (img_size if isinstance(img_size, (tuple, list)) else (img_size,) * 3)):
            raise ValueError("input image size (img_size) should be divisible by stage-wise image resolution.")
        if (in_channels != 1 or tuple(depths) != (2, 2, 2, 2) or norm_name != "instance" or drop_rate or attn_drop_rate or
                dropout_path_rate or not normalize or spatial_dims != 3 or downsample != "merging" or use_v2 or
                any(feature_size * 2 ** i // h != 16 for i, h in enumerate(num_heads))):
            raise NotImplementedError("the HIP SwinUNETR covers the configuration net_factory_3d builds: in_channels 1, "
                                      "depths (2,2,2,2), heads of 16 channels, instance norm, all dropout 0, 'merging'")
        self.in_channels, self.n_classes, self.fs = in_channels, out_channels, feature_size
        self.depths, self.heads = tuple(depths), tuple(num_heads)
        self.transposed_convs = set()
        f, D = feature_size, self._declare
        D("swinViT.patch_embed.proj.weight", _conv_default(f, in_channels, 2, 2, 2))
        b = 1.0 / math.sqrt(in_channels * 8)
        D("swinViT.patch_embed.proj.bias", torch.empty(f).uniform_(-b, b))
        for i, (depth, nh) in enumerate(zip(self.depths, self.heads)):
            dim = f * 2 ** i
            for blk in range(depth):
                p = f"swinViT.layers{i + 1}.0.blocks.{blk}"
                D(p + ".norm1.weight", torch.ones(dim)); D(p + ".norm1.bias", torch.zeros(dim))
                D(p + ".attn.relative_position_bias_table", _trunc_normal((2 * WS - 1) ** 3, nh))
                D(p + ".attn.qkv.weight", _trunc_normal(3 * dim, dim)); D(p + ".attn.qkv.bias", torch.zeros(3 * dim))
                D(p + ".attn.proj.weight", _trunc_normal(dim, dim)); D(p + ".attn.proj.bias", torch.zeros(dim))
                D(p + ".norm2.weight", torch.ones(dim)); D(p + ".norm2.bias", torch.zeros(dim))
                D(p + ".mlp.linear1.weight", _trunc_normal(4 * dim, dim)); D(p + ".mlp.linear1.bias", torch.zeros(4 * dim))
                D(p + ".mlp.linear2.weight", _trunc_normal(dim, 4 * dim)); D(p + ".mlp.linear2.bias", torch.zeros(dim))
            p = f"swinViT.layers{i + 1}.0.downsample"
            D(p + ".reduction.weight", _trunc_normal(2 * dim, 8 * dim))
            D(p + ".norm.weight", torch.ones(8 * dim)); D(p + ".norm.bias", torch.zeros(8 * dim))
        self._declare_res("encoder1.layer", in_channels, f)
        self._declare_res("encoder2.layer", f, f)
        self._declare_res("encoder3.layer", 2 * f, 2 * f)
        self._declare_res("encoder4.layer", 4 * f, 4 * f)
        self._declare_res("encoder10.layer", 16 * f, 16 * f)
        self.dec_cfg = (("decoder5", 16 * f, 8 * f), ("decoder4", 8 * f, 4 * f), ("decoder3", 4 * f, 2 * f),
                        ("decoder2", 2 * f, f), ("decoder1", f, f))
        for name, cin, cout in self.dec_cfg:
            self._declare_up(name + ".transp_conv.conv.weight", cin, cout)
            self._declare_res(name + ".conv_block", 2 * cout, cout)
        D("out.conv.conv.weight", _conv_default(out_channels, f, 1, 1, 1))
        bound = 1.0 / math.sqrt(f)
        D("out.conv.conv.bias", torch.empty(out_channels).uniform_(-bound, bound))
        self._materialize()

    _declare_up = UNETR._declare_up
    _declare_res = UNETR._declare_res
    _resblock = UNETR._resblock

    # ---- layer graph ----
    def _build(self, plan):
        N, Cin, D0, H0, W0 = plan.in_shape
        if Cin != 1 or D0 % 32 or H0 % 32 or W0 % 32:
            raise RuntimeError("SwinUNETR input must be [N,1,D,H,W] with spatial dimensions divisible by 2 ** 5 "
                               f"(MONAI _check_input_size); got {plan.in_shape}")
        P, f, B = self.P, self.fs, N

        def tok(rows, C_):
            a = sp.TAct(rows, C_)
            plan.acts.append(a)
            return a

        def add(op):
            plan.ops.append(op)
            return op

        # ---- Swin encoder (token-major [B, d, h, w, C]) ----
        dims = (D0 // 2, H0 // 2, W0 // 2)
        pe = plan.new(f, dims)
        add(DownConvOp(plan.inp, pe, P("swinViT.patch_embed.proj.weight"), P("swinViT.patch_embed.proj.bias"),
                       bias_grad=True, need_dx=False, in_shape=plan.in_shape))
        L = dims[0] * dims[1] * dims[2]
        x = tok(B * L, f)
        add(sp.VolToTokOp(pe, x, B, L))
        hidden = []          # (tokens, dims, channels) of the 5 hidden states BEFORE proj_out

        # the decoders' concat buffers [up | skip]; skips are written straight into their channel slices
        full = (D0, H0, W0)
        sz = lambda k: tuple(s // k for s in full)
        cat = {"decoder5": plan.new(16 * f, sz(16)), "decoder4": plan.new(8 * f, sz(8)), "decoder3": plan.new(4 * f, sz(4)),
               "decoder2": plan.new(2 * f, sz(2)), "decoder1": plan.new(2 * f, sz(1))}
        skip = {n: plan.view(cat[n], cout, cout) for n, _, cout in self.dec_cfg}

        def proj_out(t, dm, C_, out=None):
            """F.layer_norm over the channels (no affine), then tokens -> volume for the conv side."""
            n = tok(t.rows, C_)
            add(sp.LayerNormOp(t, n, sp.ConstRef(torch.ones(C_, device="cuda")), sp.ConstRef(torch.zeros(C_, device="cuda"))))
            v = plan.new(C_, dm) if out is None else out
            add(sp.TokToVolOp(n, v, B, dm[0] * dm[1] * dm[2]))
            return v

        hs_vol = [proj_out(x, dims, f)]
        site = 0
        for i, (depth, nh) in enumerate(zip(self.depths, self.heads)):
            dim = f * 2 ** i
            L = dims[0] * dims[1] * dims[2]
            win, shift0 = window_geometry(dims)
            n = win[0] * win[1] * win[2]
            pdims = tuple(-(-s // win[k]) * win[k] for k, s in enumerate(dims))
            nW = (pdims[0] // win[0]) * (pdims[1] // win[1]) * (pdims[2] // win[2])
            region = region_ids(pdims, win, shift0).cuda() if any(shift0) else None
            for blk in range(depth):
                p = f"swinViT.layers{i + 1}.0.blocks.{blk}"
                shift = shift0 if blk % 2 == 1 else (0, 0, 0)
                n1 = tok(B * L, dim)
                add(sp.LayerNormOp(x, n1, P(p + ".norm1.weight"), P(p + ".norm1.bias")))
                xw = tok(B * nW * n, dim)
                add(sp.Win3dOp(n1, xw, B, dims, win, shift, True))
                qkv = tok(B * nW * n, 3 * dim)
                add(sp.LinearOp(xw, qkv, P(p + ".attn.qkv.weight"), P(p + ".attn.qkv.bias")))
                att = tok(B * nW * n, dim)
                add(sp.Win3dAttnOp(qkv, att, P(p + ".attn.relative_position_bias_table"),
                                   region if any(shift) else None, B * nW, nW, n, nh))
                pr = tok(B * nW * n, dim)
                add(sp.LinearOp(att, pr, P(p + ".attn.proj.weight"), P(p + ".attn.proj.bias")))
                back = tok(B * L, dim)
                add(sp.Win3dOp(pr, back, B, dims, win, shift, False))
                x1 = tok(B * L, dim)
                add(sp.ResidualOp(x, back, x1, L, 0.0, site)); site += 1
                n2 = tok(B * L, dim)
                add(sp.LayerNormOp(x1, n2, P(p + ".norm2.weight"), P(p + ".norm2.bias")))
                h = tok(B * L, 4 * dim)
                fc1 = add(sp.LinearOp(n2, h, P(p + ".mlp.linear1.weight"), P(p + ".mlp.linear1.bias")))
                hg = tok(B * L, 4 * dim)
                fc1.gelu = gelu = add(sp.GeluOp(h, hg))
                m = tok(B * L, dim)
                fc2 = add(sp.LinearOp(hg, m, P(p + ".mlp.linear2.weight"), P(p + ".mlp.linear2.bias")))
                fc2.dx_gelu = gelu
                x2 = tok(B * L, dim)
                fc2.res = add(sp.ResidualOp(x1, m, x2, L, 0.0, site)); site += 1
                x = x2
            # PatchMerging ("merging"): v0.9 slot order -> LayerNorm(8 dim) -> Linear(8 dim, 2 dim, no bias)
            p = f"swinViT.layers{i + 1}.0.downsample"
            mg = tok(B * L // 8, 8 * dim)
            add(sp.Merge3dOp(x, mg, B, dims))
            mn = tok(B * L // 8, 8 * dim)
            add(sp.LayerNormOp(mg, mn, P(p + ".norm.weight"), P(p + ".norm.bias")))
            x = tok(B * L // 8, 2 * dim)
            add(sp.LinearOp(mn, x, P(p + ".reduction.weight"), None))
            dims = tuple(s // 2 for s in dims)
            # hidden state 3 is decoder5's skip as it is (no conv block): straight into the concat buffer
            hs_vol.append(proj_out(x, dims, 2 * dim, out=skip["decoder5"] if i == 2 else None))

        # ---- conv side (NCDHW) ----
        self._resblock(plan, "encoder1.layer", plan.inp, 1, f, sz(1), skip["decoder1"], need_dx=False)      # enc0
        self._resblock(plan, "encoder2.layer", hs_vol[0], f, f, sz(2), skip["decoder2"])                    # enc1
        self._resblock(plan, "encoder3.layer", hs_vol[1], 2 * f, 2 * f, sz(4), skip["decoder3"])            # enc2
        self._resblock(plan, "encoder4.layer", hs_vol[2], 4 * f, 4 * f, sz(8), skip["decoder4"])            # enc3
        cur = self._resblock(plan, "encoder10.layer", hs_vol[4], 16 * f, 16 * f, sz(32), plan.new(16 * f, sz(32)))   # dec4
        for (name, cin, cout), k in zip(self.dec_cfg, (16, 8, 4, 2, 1)):
            plan.up_conv(cur, plan.view(cat[name], 0, cout), P(name + ".transp_conv.conv.weight"), None)
            cur = self._resblock(plan, name + ".conv_block", cat[name], 2 * cout, cout, sz(k), plan.new(cout, sz(k)))
        plan.out = plan.new(self.n_classes, full)
        plan.conv(cur, plan.out, P("out.conv.conv.weight"), P("out.conv.conv.bias"), (1, 1, 1), bias_grad=True)

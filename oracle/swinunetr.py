"""Oracle SwinUNETR: functional torch-CPU restatement of the network ``net_factory_3d('swinunetr')`` builds (reference
code/networks/net_factory_3d.py:7,37-38: ``monai.networks.nets.SwinUNETR(img_size=(64, 64, 64), in_channels=in_chns,
out_channels=class_num, feature_size=48)``).  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED.  The reference repository contains no source of this network at all: it is imported from MONAI, an
un-vendored, un-versioned dependency that is not installed in the build image.  This file restates the PUBLISHED algorithm
of ``monai/networks/nets/swin_unetr.py`` (MONAI 1.x; the defaults the reference's call relies on: depths (2, 2, 2, 2),
num_heads (3, 6, 12, 24), window 7, patch 2, mlp_ratio 4, qkv_bias, instance norm, all dropout rates 0, normalize=True,
downsample="merging") and of the dynunet / unetr blocks oracle/unetr.py already restates.  What it pins is the call site
(net_factory_3d.py:38) -- nothing here could be checked against the reference's own arithmetic.

Published algorithm, as restated below:
  PatchEmbed        Conv3d(in, 48, kernel 2, stride 2) with bias, no norm
  BasicLayer i      dim 48 * 2^i at (D/2^(i+1))^3: 2 x SwinTransformerBlock (shift 0, then window // 2 where the
                    volume is larger than the window), THEN PatchMerging
  SwinBlock         x = x + Attn(LN(x)); x = x + MLP(LN(x)).  Attn: zero-pad (after the LayerNorm) to a multiple of the
                    window, cyclic shift, 7^3-token windows (window clipped to the volume when the volume is not larger:
                    4^3 -> 64 tokens, no shift), qkv Linear with bias, heads of 16 channels, q * 16^-0.5, + relative position
                    bias table[(2*7-1)^3][heads] through ``relative_position_index[:n, :n]`` of the FULL 7^3 window (also
                    for clipped windows), + the shift mask (0 / -100 from the 27 regions; only in shifted blocks -- padded
                    tokens are ordinary keys), softmax, proj Linear, window reverse, un-shift, un-pad
  PatchMerging      "merging" = the v0.9 order: cat of x[0::2,0::2,0::2], x[1::2,0::2,0::2], x[0::2,1::2,0::2],
                    x[0::2,0::2,1::2], x[1::2,0::2,1::2], x[0::2,1::2,0::2], x[0::2,0::2,1::2], x[1::2,1::2,1::2] (two slices
                    appear twice, two never), LayerNorm(8 dim), Linear(8 dim -> 2 dim) without bias
  proj_out          F.layer_norm over the channels (no affine) of each of the 5 hidden states
  encoder1/2/3/4/10 UnetResBlock on the input / hidden states 0, 1, 2 / hidden state 4
  decoder5..1       ConvTranspose3d(k2, s2) -> cat([up, skip]) -> UnetResBlock; skips: hidden state 3, enc3, enc2, enc1, enc0
  out               Conv3d 1x1x1 with bias
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

WS = 7


def window_geometry(dims, ws=WS):
    """(window, shift) MONAI's get_window_size gives a volume: the window is clipped (and the shift dropped) per axis
    when the volume is not larger than it."""
    win = tuple(d if d <= ws else ws for d in dims)
    shift = tuple(0 if d <= ws else ws // 2 for d in dims)
    return win, shift


def relative_position_index(ws=WS):
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rc = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rc += ws - 1
    return rc[:, :, 0] * (2 * ws - 1) ** 2 + rc[:, :, 1] * (2 * ws - 1) + rc[:, :, 2]


def window_partition(x, win):
    b, d, h, w, c = x.shape
    x = x.view(b, d // win[0], win[0], h // win[1], win[1], w // win[2], win[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, win[0] * win[1] * win[2], c)


def window_reverse(windows, win, dims):
    b, d, h, w = dims
    x = windows.view(b, d // win[0], h // win[1], w // win[2], win[0], win[1], win[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def region_ids(pdims, win, shift):
    """MONAI compute_mask's img_mask, window-partitioned: [nW, n] region id of every token of every window."""
    img = torch.zeros((1, *pdims, 1))
    cnt = 0
    for dsl in (slice(-win[0]), slice(-win[0], -shift[0]), slice(-shift[0], None)):
        for hsl in (slice(-win[1]), slice(-win[1], -shift[1]), slice(-shift[1], None)):
            for wsl in (slice(-win[2]), slice(-win[2], -shift[2]), slice(-shift[2], None)):
                img[:, dsl, hsl, wsl, :] = cnt
                cnt += 1
    return window_partition(img, win).squeeze(-1)


MERGE_OFFSETS = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1))


class OracleSwinUNETR:
    def __init__(self, num_classes=2, in_channels=1, feature_size=48, depths=(2, 2, 2, 2), heads=(3, 6, 12, 24)):
        self.nc, self.cin, self.fs, self.depths, self.heads = num_classes, in_channels, feature_size, depths, heads

    # ---- state (MONAI parameter names, registration order) ----
    def _res(self, p, cin, cout):
        k = [(p + ".conv1.conv.weight", (cout, cin, 3, 3, 3)), (p + ".conv2.conv.weight", (cout, cout, 3, 3, 3))]
        if cin != cout:
            k.append((p + ".conv3.conv.weight", (cout, cin, 1, 1, 1)))
        return k

    def spec(self):
        f = self.fs
        keys = [("swinViT.patch_embed.proj.weight", (f, self.cin, 2, 2, 2)), ("swinViT.patch_embed.proj.bias", (f,))]
        for i, (depth, nh) in enumerate(zip(self.depths, self.heads)):
            dim = f * 2 ** i
            for b in range(depth):
                p = f"swinViT.layers{i + 1}.0.blocks.{b}"
                keys += [(p + ".norm1.weight", (dim,)), (p + ".norm1.bias", (dim,)),
                         (p + ".attn.relative_position_bias_table", ((2 * WS - 1) ** 3, nh)),
                         (p + ".attn.qkv.weight", (3 * dim, dim)), (p + ".attn.qkv.bias", (3 * dim,)),
                         (p + ".attn.proj.weight", (dim, dim)), (p + ".attn.proj.bias", (dim,)),
                         (p + ".norm2.weight", (dim,)), (p + ".norm2.bias", (dim,)),
                         (p + ".mlp.linear1.weight", (4 * dim, dim)), (p + ".mlp.linear1.bias", (4 * dim,)),
                         (p + ".mlp.linear2.weight", (dim, 4 * dim)), (p + ".mlp.linear2.bias", (dim,))]
            p = f"swinViT.layers{i + 1}.0.downsample"
            keys += [(p + ".reduction.weight", (2 * dim, 8 * dim)), (p + ".norm.weight", (8 * dim,)),
                     (p + ".norm.bias", (8 * dim,))]
        keys += self._res("encoder1.layer", self.cin, f)
        keys += self._res("encoder2.layer", f, f) + self._res("encoder3.layer", 2 * f, 2 * f)
        keys += self._res("encoder4.layer", 4 * f, 4 * f) + self._res("encoder10.layer", 16 * f, 16 * f)
        for name, cin, cout in (("decoder5", 16 * f, 8 * f), ("decoder4", 8 * f, 4 * f), ("decoder3", 4 * f, 2 * f),
                                ("decoder2", 2 * f, f), ("decoder1", f, f)):
            keys.append((name + ".transp_conv.conv.weight", (cin, cout, 2, 2, 2)))
            keys += self._res(name + ".conv_block", 2 * cout, cout)
        keys += [("out.conv.conv.weight", (self.nc, f, 1, 1, 1)), ("out.conv.conv.bias", (self.nc,))]
        return keys

    def new_state(self):
        sd = OrderedDict()
        for name, shape in self.spec():
            sd[name] = torch.ones(shape) if (".norm" in name and name.endswith("weight")) else torch.zeros(shape)
        return sd

    @staticmethod
    def is_param(name):
        return True

    # ---- blocks ----
    @staticmethod
    def _in(x):
        return F.instance_norm(x, eps=1e-5)

    def _resblock(self, sd, p, x):
        out = F.leaky_relu(self._in(F.conv3d(x, sd[p + ".conv1.conv.weight"], padding=1)), 0.01)
        out = self._in(F.conv3d(out, sd[p + ".conv2.conv.weight"], padding=1))
        res = x
        if p + ".conv3.conv.weight" in sd:
            res = self._in(F.conv3d(x, sd[p + ".conv3.conv.weight"]))
        return F.leaky_relu(out + res, 0.01)

    def _up(self, sd, name, x, skip):
        x = F.conv_transpose3d(x, sd[name + ".transp_conv.conv.weight"], stride=2)
        return self._resblock(sd, name + ".conv_block", torch.cat((x, skip), dim=1))

    def _attention(self, sd, p, xw, nh, mask):
        b, n, c = xw.shape
        qkv = F.linear(xw, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(b, n, 3, nh, c // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (c // nh) ** -0.5, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        idx = relative_position_index()[:n, :n].reshape(-1)
        bias = sd[p + ".relative_position_bias_table"][idx].reshape(n, n, -1).permute(2, 0, 1)
        attn = attn + bias.unsqueeze(0)
        if mask is not None:
            nw = mask.shape[0]
            attn = (attn.view(b // nw, nw, nh, n, n) + mask.to(attn.dtype).unsqueeze(1).unsqueeze(0)).view(-1, nh, n, n)
        attn = torch.softmax(attn, dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(b, n, c)
        return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])

    def _block(self, sd, p, x, nh, shifted, mask):
        b, d, h, w, c = x.shape
        win, shift = window_geometry((d, h, w))
        if not shifted:
            shift = (0, 0, 0)
        shortcut = x
        x = F.layer_norm(x, (c,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
        pd, ph, pw = ((win[i] - s % win[i]) % win[i] for i, s in enumerate((d, h, w)))
        x = F.pad(x, (0, 0, 0, pw, 0, ph, 0, pd))
        dims = (b, d + pd, h + ph, w + pw)
        use_mask = None
        if any(s > 0 for s in shift):
            x = torch.roll(x, shifts=(-shift[0], -shift[1], -shift[2]), dims=(1, 2, 3))
            use_mask = mask
        aw = self._attention(sd, p + ".attn", window_partition(x, win), nh, use_mask)
        x = window_reverse(aw, win, dims)
        if any(s > 0 for s in shift):
            x = torch.roll(x, shifts=shift, dims=(1, 2, 3))
        x = x[:, :d, :h, :w, :].contiguous()
        x = shortcut + x
        y = F.layer_norm(x, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
        y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.linear1.weight"], sd[p + ".mlp.linear1.bias"])),
                     sd[p + ".mlp.linear2.weight"], sd[p + ".mlp.linear2.bias"])
        return x + y

    def _layer(self, sd, i, x):
        """BasicLayer i on x [b, c, d, h, w]: the blocks, then the patch merging."""
        b, c, d, h, w = x.shape
        win, shift = window_geometry((d, h, w))
        x = x.permute(0, 2, 3, 4, 1).contiguous()
        pdims = tuple(-(-s // win[k]) * win[k] for k, s in enumerate((d, h, w)))
        mask = None
        if any(s > 0 for s in shift):
            r = region_ids(pdims, win, shift)
            mask = r.unsqueeze(1) - r.unsqueeze(2)
            mask = mask.masked_fill(mask != 0, -100.0).masked_fill(mask == 0, 0.0)
        for blk in range(self.depths[i]):
            x = self._block(sd, f"swinViT.layers{i + 1}.0.blocks.{blk}", x, self.heads[i], blk % 2 == 1, mask)
        p = f"swinViT.layers{i + 1}.0.downsample"
        x = torch.cat([x[:, o[0]::2, o[1]::2, o[2]::2, :] for o in MERGE_OFFSETS], -1)
        x = F.layer_norm(x, (8 * c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
        x = F.linear(x, sd[p + ".reduction.weight"])
        return x.permute(0, 4, 1, 2, 3).contiguous()

    @staticmethod
    def _proj_out(x):
        c = x.shape[1]
        return F.layer_norm(x.permute(0, 2, 3, 4, 1), (c,)).permute(0, 4, 1, 2, 3).contiguous()

    def hidden_states(self, sd, x_in):
        x = F.conv3d(x_in, sd["swinViT.patch_embed.proj.weight"], sd["swinViT.patch_embed.proj.bias"], stride=2)
        outs = [self._proj_out(x)]
        for i in range(4):
            x = self._layer(sd, i, x)
            outs.append(self._proj_out(x))
        return outs

    def forward(self, sd, x_in, training=True, drop=None):
        if any(s % 32 for s in x_in.shape[2:]):
            raise ValueError("spatial dimensions of input image must be divisible by 2 ** 5")      # MONAI _check_input_size
        hs = self.hidden_states(sd, x_in)
        enc0 = self._resblock(sd, "encoder1.layer", x_in)
        enc1 = self._resblock(sd, "encoder2.layer", hs[0])
        enc2 = self._resblock(sd, "encoder3.layer", hs[1])
        enc3 = self._resblock(sd, "encoder4.layer", hs[2])
        dec4 = self._resblock(sd, "encoder10.layer", hs[4])
        dec3 = self._up(sd, "decoder5", dec4, hs[3])
        dec2 = self._up(sd, "decoder4", dec3, enc3)
        dec1 = self._up(sd, "decoder3", dec2, enc2)
        dec0 = self._up(sd, "decoder2", dec1, enc1)
        out = self._up(sd, "decoder1", dec0, enc0)
        return F.conv3d(out, sd["out.conv.conv.weight"], sd["out.conv.conv.bias"])

    def drop_sites(self, in_shape):
        return []          # drop_rate = attn_drop_rate = dropout_path_rate = 0 (SwinUNETR defaults)

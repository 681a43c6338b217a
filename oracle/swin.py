"""Oracle SwinUnet: functional torch-CPU restatement of the reference's SwinUnet / SwinTransformerSys
(code/networks/vision_transformer.py:24-52, code/networks/swin_transformer_unet_skip_expand_decoder_sys.py).
TEST INFRASTRUCTURE (see oracle/__init__.py).  State = ordered ``name -> tensor`` dict with the reference's
238 state_dict keys.

DropPath sites are numbered in forward order (two per block: attention residual, MLP residual).
``drop``: None -> stock per-sample Bernoulli (timm DropPath semantics, training only); "off" -> p := 0;
dict {site: per-sample scale tensor [B] (0 or 1/(1-p))} -> injected.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from einops import rearrange

def _rel_pos_index(WS):
    coords = torch.stack(torch.meshgrid([torch.arange(WS), torch.arange(WS)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += WS - 1
    rel[:, :, 1] += WS - 1
    rel[:, :, 0] *= 2 * WS - 1
    return rel.sum(-1)


def _window_partition(x, WS):                                   # swin...sys.py:28-41
    B, H, W, C = x.shape
    x = x.view(B, H // WS, WS, W // WS, WS, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, WS, WS, C)


def _window_reverse(windows, H, W, WS):                         # :44-60
    B = int(windows.shape[0] / (H * W / WS / WS))
    x = windows.view(B, H // WS, W // WS, WS, WS, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def _attn_mask(H, W, shift, WS):                                # :216-238
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for hs in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
        for ws_ in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
            img[:, hs, ws_, :] = cnt
            cnt += 1
    mw = _window_partition(img, WS).view(-1, WS * WS)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


class OracleSwinUnet:
    def __init__(self, num_classes=4, img_size=224, embed_dim=96, depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24),
                 in_chans=3, mlp_ratio=4.0, drop_path_rate=0.2, window=7):
        # window 7 / img 224 (the reference yaml) or window 8 / img 256 (config.py:194-195 overrides)
        assert img_size % (32 * window) == 0
        self.ws = window
        self.nc, self.img, self.E = num_classes, img_size, embed_dim
        self.depths, self.heads, self.in_chans, self.mlp = list(depths), list(num_heads), in_chans, mlp_ratio
        self.dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.pr = img_size // 4

    def _shift(self, res, b):
        return 0 if (b % 2 == 0 or res <= self.ws) else self.ws // 2

    # ---- state ----
    def spec(self):
        E, nl, keys, WS = self.E, len(self.depths), [], self.ws

        def lin(n, o, i, bias=True):
            keys.append((n + ".weight", (o, i), "p"))
            if bias:
                keys.append((n + ".bias", (o,), "p"))

        def ln(n, c):
            keys.append((n + ".weight", (c,), "p"))
            keys.append((n + ".bias", (c,), "p"))

        def block(p, dim, res, heads, shift):
            if shift > 0:
                keys.append((p + ".attn_mask", ((res // WS) ** 2, WS * WS, WS * WS), "mask"))
            ln(p + ".norm1", dim)
            keys.append((p + ".attn.relative_position_bias_table", ((2 * WS - 1) ** 2, heads), "p"))
            keys.append((p + ".attn.relative_position_index", (WS * WS, WS * WS), "index"))
            lin(p + ".attn.qkv", 3 * dim, dim)
            lin(p + ".attn.proj", dim, dim)
            ln(p + ".norm2", dim)
            lin(p + ".mlp.fc1", int(dim * self.mlp), dim)
            lin(p + ".mlp.fc2", dim, int(dim * self.mlp))

        keys.append(("swin_unet.patch_embed.proj.weight", (E, self.in_chans, 4, 4), "p"))
        keys.append(("swin_unet.patch_embed.proj.bias", (E,), "p"))
        ln("swin_unet.patch_embed.norm", E)
        for i in range(nl):
            dim, res = E * 2 ** i, self.pr // 2 ** i
            for b in range(self.depths[i]):
                block(f"swin_unet.layers.{i}.blocks.{b}", dim, res, self.heads[i], self._shift(res, b))
            if i < nl - 1:
                lin(f"swin_unet.layers.{i}.downsample.reduction", 2 * dim, 4 * dim, bias=False)
                ln(f"swin_unet.layers.{i}.downsample.norm", 4 * dim)
        for i in range(nl):
            k = nl - 1 - i
            dim, res = E * 2 ** k, self.pr // 2 ** k
            if i == 0:
                lin("swin_unet.layers_up.0.expand", 2 * dim, dim, bias=False)
                ln("swin_unet.layers_up.0.norm", dim // 2)
            else:
                for b in range(self.depths[k]):
                    block(f"swin_unet.layers_up.{i}.blocks.{b}", dim, res, self.heads[k], self._shift(res, b))
                if i < nl - 1:
                    lin(f"swin_unet.layers_up.{i}.upsample.expand", 2 * dim, dim, bias=False)
                    ln(f"swin_unet.layers_up.{i}.upsample.norm", dim // 2)
        for i in range(1, nl):
            dim = E * 2 ** (nl - 1 - i)
            lin(f"swin_unet.concat_back_dim.{i}", dim, 2 * dim)
        ln("swin_unet.norm", E * 2 ** (nl - 1))
        ln("swin_unet.norm_up", E)
        lin("swin_unet.up.expand", 16 * E, E, bias=False)
        ln("swin_unet.up.norm", E)
        keys.append(("swin_unet.output.weight", (self.nc, E, 1, 1), "p"))
        return keys

    def new_state(self):
        sd = OrderedDict()
        WS = self.ws
        for name, shape, kind in self.spec():
            if kind == "index":
                sd[name] = _rel_pos_index(WS)
            elif kind == "mask":
                res = int(round((shape[0]) ** 0.5)) * WS
                sd[name] = _attn_mask(res, res, WS // 2, WS)
            else:
                sd[name] = torch.ones(shape) if (name.endswith("norm.weight") or ".norm1.weight" in name or
                                                 ".norm2.weight" in name or name.endswith("norm_up.weight")) \
                    else torch.zeros(shape)
        return sd

    @staticmethod
    def is_param(name):
        return not (name.endswith("attn_mask") or name.endswith("relative_position_index"))

    # ---- forward ----
    def _droppath(self, x, p, training, drop, site):
        if not training or p == 0.0 or drop == "off":
            return x
        if isinstance(drop, dict):
            return x * drop[site].to(x.dtype).view(-1, *([1] * (x.dim() - 1)))
        keep = 1 - p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep

    def _block(self, sd, p, x, res, dim, heads, shift, dp, training, drop, sites):
        B, L, C = x.shape
        WS = self.ws
        shortcut = x
        x = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5).view(B, res, res, C)
        if shift > 0:
            x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        xw = _window_partition(x, WS).view(-1, WS * WS, C)
        B_, N, _ = xw.shape
        qkv = F.linear(xw, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
        qkv = qkv.reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * ((C // heads) ** -0.5), qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        table = sd[p + ".attn.relative_position_bias_table"]
        idx = sd[p + ".attn.relative_position_index"]
        attn = attn + table[idx.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
        if shift > 0:
            mask = sd[p + ".attn_mask"].to(attn.dtype)
            nW = mask.shape[0]
            attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
        attn = attn.softmax(dim=-1)
        xw = (attn @ v).transpose(1, 2).reshape(B_, N, C)
        xw = F.linear(xw, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
        x = _window_reverse(xw.view(-1, WS, WS, C), res, res, WS)
        if shift > 0:
            x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
        x = x.view(B, res * res, C)
        x = shortcut + self._droppath(x, dp, training, drop, next(sites))
        h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])),
                     sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
        return x + self._droppath(h, dp, training, drop, next(sites))

    def _expand(self, sd, p, x, res, P_):
        x = F.linear(x, sd[p + ".expand.weight"])
        B, L, C = x.shape
        x = rearrange(x.view(B, res, res, C), 'b h w (p1 p2 c)-> b (h p1) (w p2) c', p1=P_, p2=P_, c=C // (P_ * P_))
        x = x.reshape(B, -1, C // (P_ * P_))
        return F.layer_norm(x, (x.shape[-1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)

    def forward(self, sd, x, training=True, drop=None):
        import itertools
        sites = itertools.count(0)
        E, nl = self.E, len(self.depths)
        if x.size(1) == 1:
            x = x.repeat(1, 3, 1, 1)                                             # vision_transformer.py:49-50
        x = F.conv2d(x, sd["swin_unet.patch_embed.proj.weight"], sd["swin_unet.patch_embed.proj.bias"], stride=4)
        x = x.flatten(2).transpose(1, 2)
        x = F.layer_norm(x, (E,), sd["swin_unet.patch_embed.norm.weight"], sd["swin_unet.patch_embed.norm.bias"], 1e-5)
        downs, di = [], 0
        for i in range(nl):
            downs.append(x)
            dim, res = E * 2 ** i, self.pr // 2 ** i
            for b in range(self.depths[i]):
                x = self._block(sd, f"swin_unet.layers.{i}.blocks.{b}", x, res, dim, self.heads[i],
                                self._shift(res, b), self.dpr[di], training, drop, sites)
                di += 1
            if i < nl - 1:                                                       # PatchMerging :325-346
                p = f"swin_unet.layers.{i}.downsample"
                B = x.shape[0]
                x = x.view(B, res, res, dim)
                x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
                x = x.view(B, -1, 4 * dim)
                x = F.layer_norm(x, (4 * dim,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
                x = F.linear(x, sd[p + ".reduction.weight"])
        dimb = E * 2 ** (nl - 1)
        x = F.layer_norm(x, (dimb,), sd["swin_unet.norm.weight"], sd["swin_unet.norm.bias"], 1e-5)
        for i in range(nl):
            k = nl - 1 - i
            dim, res = E * 2 ** k, self.pr // 2 ** k
            if i == 0:
                x = self._expand(sd, "swin_unet.layers_up.0", x, res, 2)
            else:
                x = torch.cat([x, downs[3 - i]], -1)                             # :767
                x = F.linear(x, sd[f"swin_unet.concat_back_dim.{i}.weight"], sd[f"swin_unet.concat_back_dim.{i}.bias"])
                base = sum(self.depths[:k])
                for b in range(self.depths[k]):
                    x = self._block(sd, f"swin_unet.layers_up.{i}.blocks.{b}", x, res, dim, self.heads[k],
                                    self._shift(res, b), self.dpr[base + b], training, drop, sites)
                if i < nl - 1:
                    x = self._expand(sd, f"swin_unet.layers_up.{i}.upsample", x, res, 2)
        x = F.layer_norm(x, (E,), sd["swin_unet.norm_up.weight"], sd["swin_unet.norm_up.bias"], 1e-5)
        x = self._expand(sd, "swin_unet.up", x, self.pr, 4)
        B = x.shape[0]
        x = x.view(B, 4 * self.pr, 4 * self.pr, -1).permute(0, 3, 1, 2)
        return F.conv2d(x, sd["swin_unet.output.weight"])

    def drop_sites(self, in_shape):
        """[(site, p, (B,))] for the residuals whose DropPath rate is > 0."""
        B = in_shape[0]
        out, site, nl = [], 0, len(self.depths)
        rates = list(self.dpr)
        for i in range(1, nl):
            k = nl - 1 - i
            rates += self.dpr[sum(self.depths[:k]):sum(self.depths[:k + 1])]
        for r in rates:
            for _ in range(2):
                if r > 0:
                    out.append((site, r, (B,)))
                site += 1
        return out

"""Oracle UNETR: functional torch-CPU restatement of the network ``net_factory_3d('unetr')`` builds (reference
code/networks/net_factory_3d.py:23-36, code/networks/unetr.py:22-230).  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED.  code/networks/unetr.py assembles the network from MONAI blocks (``monai.networks.nets.ViT``,
``UnetrBasicBlock`` / ``UnetrPrUpBlock`` / ``UnetrUpBlock``, ``UnetOutBlock``); MONAI is an un-vendored dependency of
the reference (no version is pinned in the repository and the package is not installed in the build image), so this
file restates the PUBLISHED algorithm of those blocks (MONAI 0.8-era ``networks/blocks/{patchembedding,
transformerblock,selfattention,mlp,unetr_block,dynunet_block}.py``) and nothing here could be checked against the
reference's own arithmetic.  What it pins: the call-site arguments of net_factory_3d.py:23-36 (in_channels 1,
img_size 96^3, feature_size 16, hidden 768, mlp 3072, 12 heads, 'perceptron' position embedding, instance norm,
conv_block / res_block True, dropout 0) and unetr.py's wiring (hidden states 3 / 6 / 9, proj_feat, decoder order).

Blocks:
  ViT            Rearrange('b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)') -> Linear(4096, 768) -> + position
                 embeddings; 12 x [x += SA(LN(x)); x += MLP(LN(x))] (SA: bias-free qkv Linear, 12 heads of 64, softmax(q k^T
                 / 8) v, out_proj Linear; MLP: Linear-GELU-Linear); final LayerNorm; hidden state after every block
  UnetResBlock   conv3-IN-LeakyReLU(0.01)-conv3-IN, + (conv1-IN of the input when channels differ), LeakyReLU; convs
                 without bias, InstanceNorm3d without affine
  UnetrPrUpBlock ConvTranspose3d(k2, s2) then num_layer x [ConvTranspose3d(k2, s2) -> UnetResBlock]
  UnetrUpBlock   ConvTranspose3d(k2, s2) -> cat([up, skip]) -> UnetResBlock
  UnetOutBlock   Conv3d 1x1x1 with bias
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F


class OracleUNETR:
    def __init__(self, num_classes=2, img_size=(96, 96, 96), feature_size=16, hidden=768, mlp_dim=3072, heads=12,
                 layers=12):
        self.nc, self.img, self.fs, self.hid, self.mlp, self.heads, self.layers = \
            num_classes, tuple(img_size), feature_size, hidden, mlp_dim, heads, layers
        self.feat = tuple(s // 16 for s in self.img)
        self.L = self.feat[0] * self.feat[1] * self.feat[2]

    # ---- state (MONAI 0.8-era parameter names, registration order) ----
    def _res(self, p, cin, cout):
        k = [(p + ".conv1.conv.weight", (cout, cin, 3, 3, 3)), (p + ".conv2.conv.weight", (cout, cout, 3, 3, 3))]
        if cin != cout:
            k.append((p + ".conv3.conv.weight", (cout, cin, 1, 1, 1)))
        return k

    def spec(self):
        H, f = self.hid, self.fs
        keys = [("vit.patch_embedding.position_embeddings", (1, self.L, H)), ("vit.patch_embedding.cls_token", (1, 1, H)),
                ("vit.patch_embedding.patch_embeddings.1.weight", (H, 4096)),
                ("vit.patch_embedding.patch_embeddings.1.bias", (H,))]
        for i in range(self.layers):
            p = f"vit.blocks.{i}"
            keys += [(p + ".mlp.linear1.weight", (self.mlp, H)), (p + ".mlp.linear1.bias", (self.mlp,)),
                     (p + ".mlp.linear2.weight", (H, self.mlp)), (p + ".mlp.linear2.bias", (H,)),
                     (p + ".norm1.weight", (H,)), (p + ".norm1.bias", (H,)),
                     (p + ".attn.out_proj.weight", (H, H)), (p + ".attn.out_proj.bias", (H,)),
                     (p + ".attn.qkv.weight", (3 * H, H)),
                     (p + ".norm2.weight", (H,)), (p + ".norm2.bias", (H,))]
        keys += [("vit.norm.weight", (H,)), ("vit.norm.bias", (H,))]
        keys += self._res("encoder1.layer", 1, f)
        for name, cout, nl in (("encoder2", 2 * f, 2), ("encoder3", 4 * f, 1), ("encoder4", 8 * f, 0)):
            keys.append((name + ".transp_conv_init.conv.weight", (H, cout, 2, 2, 2)))
            for b in range(nl):
                keys.append((f"{name}.blocks.{b}.0.conv.weight", (cout, cout, 2, 2, 2)))
                keys += self._res(f"{name}.blocks.{b}.1", cout, cout)
        for name, cin, cout in (("decoder5", H, 8 * f), ("decoder4", 8 * f, 4 * f), ("decoder3", 4 * f, 2 * f),
                                ("decoder2", 2 * f, f)):
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

    def _prup(self, sd, name, x, nl):
        x = F.conv_transpose3d(x, sd[name + ".transp_conv_init.conv.weight"], stride=2)
        for b in range(nl):
            x = F.conv_transpose3d(x, sd[f"{name}.blocks.{b}.0.conv.weight"], stride=2)
            x = self._resblock(sd, f"{name}.blocks.{b}.1", x)
        return x

    def _up(self, sd, name, x, skip):
        x = F.conv_transpose3d(x, sd[name + ".transp_conv.conv.weight"], stride=2)
        return self._resblock(sd, name + ".conv_block", torch.cat((x, skip), dim=1))

    def _proj(self, t):
        B = t.shape[0]
        return t.view(B, *self.feat, self.hid).permute(0, 4, 1, 2, 3).contiguous()

    def forward(self, sd, x_in, training=True, drop=None):
        B, H, nh = x_in.shape[0], self.hid, self.heads
        h, w, d = self.feat
        # Rearrange('b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)')
        t = x_in.view(B, 1, h, 16, w, 16, d, 16).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, self.L, 4096)
        t = F.linear(t, sd["vit.patch_embedding.patch_embeddings.1.weight"], sd["vit.patch_embedding.patch_embeddings.1.bias"])
        t = t + sd["vit.patch_embedding.position_embeddings"]
        hidden = []
        for i in range(self.layers):
            p = f"vit.blocks.{i}"
            n = F.layer_norm(t, (H,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
            qkv = F.linear(n, sd[p + ".attn.qkv.weight"]).view(B, self.L, 3, nh, H // nh).permute(2, 0, 3, 1, 4)
            att = torch.softmax((qkv[0] @ qkv[1].transpose(-2, -1)) * (H // nh) ** -0.5, dim=-1)
            o = (att @ qkv[2]).transpose(1, 2).reshape(B, self.L, H)
            t = t + F.linear(o, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
            n = F.layer_norm(t, (H,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
            t = t + F.linear(F.gelu(F.linear(n, sd[p + ".mlp.linear1.weight"], sd[p + ".mlp.linear1.bias"])),
                             sd[p + ".mlp.linear2.weight"], sd[p + ".mlp.linear2.bias"])
            hidden.append(t)
        x = F.layer_norm(t, (H,), sd["vit.norm.weight"], sd["vit.norm.bias"], 1e-5)
        enc1 = self._resblock(sd, "encoder1.layer", x_in)
        enc2 = self._prup(sd, "encoder2", self._proj(hidden[3]), 2)
        enc3 = self._prup(sd, "encoder3", self._proj(hidden[6]), 1)
        enc4 = self._prup(sd, "encoder4", self._proj(hidden[9]), 0)
        dec3 = self._up(sd, "decoder5", self._proj(x), enc4)
        dec2 = self._up(sd, "decoder4", dec3, enc3)
        dec1 = self._up(sd, "decoder3", dec2, enc2)
        out = self._up(sd, "decoder2", dec1, enc1)
        return F.conv3d(out, sd["out.conv.conv.weight"], sd["out.conv.conv.bias"])

    def drop_sites(self, in_shape):
        return []          # dropout_rate = 0.0 (net_factory_3d.py:35)

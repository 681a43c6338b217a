"""Oracle networks: functional torch-CPU restatement of the reference's UNet and unet_3D.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters/buffers live in an ordered
``name -> tensor`` dict with exactly the reference's state_dict keys, so a reference state_dict,
an oracle state and a HIP network's state_dict are interchangeable.

Dropout sites are numbered in forward order.  ``drop`` selects their behaviour:
  None      -> stock F.dropout (training) -- what the reference does; used for CPU-baseline timing
  "off"     -> p := 0 everywhere (fixture mode (ii) of SURVEY.md s.8c)
  dict      -> {site index: scale mask (0 or 1/(1-p))} injected masks
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F


# Pre-activation hook (tests/test_parity_gpu.py, float64 gate): ``PRE_ACT(x) -> x`` is called on the input of every ReLU /
# LeakyReLU of a forward that records gradients (the student's; the teacher runs under no_grad) in call order.  The gate uses
# it to find the pre-activations that lie within fp32 rounding of zero and to re-evaluate the float64 step with their signs
# reversed -- the same measurement oracle/gen_golden.py::reference_grads64 makes on the reference modules with forward hooks.
PRE_ACT = None


def _pre_act(x):
    if PRE_ACT is None or not torch.is_grad_enabled():
        return x
    return PRE_ACT(x)


def _dropout(x, p, training, drop, site):
    if not training or p == 0.0 or drop == "off":
        return x
    if isinstance(drop, dict):
        return x * drop[site]
    return F.dropout(x, p, True)


class OracleUNet2D:
    """reference code/networks/unet.py:304-321 (UNet) with Encoder :89-116, Decoder :119-153."""

    FT = [16, 32, 64, 128, 256]
    DROPOUT = [0.05, 0.1, 0.2, 0.3, 0.5]   # unet.py:310

    def __init__(self, in_chns, class_num, bilinear=True):
        # bilinear=False: UpBlock's other branch, nn.ConvTranspose2d(C1, C2, 2, stride=2) (unet.py:76-78, :81-84)
        self.in_chns, self.class_num, self.bilinear = in_chns, class_num, bilinear

    def _block_keys(self, prefix, cin, cout):
        out = []
        for idx, ci in ((0, cin), (4, cout)):
            out += [(f"{prefix}.{idx}.weight", (cout, ci, 3, 3)), (f"{prefix}.{idx}.bias", (cout,)),
                    (f"{prefix}.{idx + 1}.weight", (cout,)), (f"{prefix}.{idx + 1}.bias", (cout,)),
                    (f"{prefix}.{idx + 1}.running_mean", (cout,)), (f"{prefix}.{idx + 1}.running_var", (cout,)),
                    (f"{prefix}.{idx + 1}.num_batches_tracked", ())]
        return out

    def spec(self):
        ft = self.FT
        keys = self._block_keys("encoder.in_conv.conv_conv", self.in_chns, ft[0])
        for i in range(1, 5):
            keys += self._block_keys(f"encoder.down{i}.maxpool_conv.1.conv_conv", ft[i - 1], ft[i])
        for i in range(1, 5):
            c1, c2 = ft[5 - i], ft[4 - i]
            if self.bilinear:
                keys += [(f"decoder.up{i}.conv1x1.weight", (c2, c1, 1, 1)), (f"decoder.up{i}.conv1x1.bias", (c2,))]
            else:
                keys += [(f"decoder.up{i}.up.weight", (c1, c2, 2, 2)), (f"decoder.up{i}.up.bias", (c2,))]
            keys += self._block_keys(f"decoder.up{i}.conv.conv_conv", 2 * c2, c2)
        keys += [("decoder.out_conv.weight", (self.class_num, ft[0], 3, 3)),
                 ("decoder.out_conv.bias", (self.class_num,))]
        return keys

    def new_state(self):
        sd = OrderedDict()
        for name, shape in self.spec():
            if name.endswith("num_batches_tracked"):
                sd[name] = torch.zeros((), dtype=torch.long)
            elif name.endswith("running_var"):
                sd[name] = torch.ones(shape)
            else:
                sd[name] = torch.zeros(shape)
        return sd

    @staticmethod
    def is_param(name):
        return not (name.endswith("running_mean") or name.endswith("running_var")
                    or name.endswith("num_batches_tracked"))

    # ConvBlock, unet.py:31-47
    def _conv_block(self, sd, prefix, x, p, training, drop, site):
        for idx, pp in ((0, p), (4, None)):
            x = F.conv2d(x, sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"], padding=1)
            bn = idx + 1
            if training:
                sd[f"{prefix}.{bn}.num_batches_tracked"] += 1
            x = F.batch_norm(x, sd[f"{prefix}.{bn}.running_mean"], sd[f"{prefix}.{bn}.running_var"],
                             sd[f"{prefix}.{bn}.weight"], sd[f"{prefix}.{bn}.bias"], training, 0.1, 1e-5)
            x = F.leaky_relu(_pre_act(x), 0.01)
            if pp is not None:
                x = _dropout(x, pp, training, drop, site)
        return x

    def forward(self, sd, x, training=True, drop=None):
        ft = self.FT
        feats = []
        site = 0
        x = self._conv_block(sd, "encoder.in_conv.conv_conv", x, self.DROPOUT[0], training, drop, site)
        feats.append(x)
        for i in range(1, 5):
            site += 1
            x = F.max_pool2d(x, 2)                                                   # unet.py:56
            x = self._conv_block(sd, f"encoder.down{i}.maxpool_conv.1.conv_conv", x, self.DROPOUT[i], training,
                                 drop, site)
            feats.append(x)
        for i in range(1, 5):                                                        # UpBlock, unet.py:65-86
            skip = feats[4 - i]
            if self.bilinear:
                x = F.conv2d(x, sd[f"decoder.up{i}.conv1x1.weight"], sd[f"decoder.up{i}.conv1x1.bias"])
                x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
            else:
                x = F.conv_transpose2d(x, sd[f"decoder.up{i}.up.weight"], sd[f"decoder.up{i}.up.bias"], stride=2)
            x = torch.cat([skip, x], dim=1)
            x = self._conv_block(sd, f"decoder.up{i}.conv.conv_conv", x, 0.0, training, "off", -1)
        return F.conv2d(x, sd["decoder.out_conv.weight"], sd["decoder.out_conv.bias"], padding=1)

    def drop_sites(self, in_shape):
        """[(site, p, activation shape)] for an input [N,C,H,W]."""
        N, _, H, W = in_shape
        return [(l, self.DROPOUT[l], (N, self.FT[l], H >> l, W >> l)) for l in range(5)]


class OracleUNet3D:
    """reference code/networks/unet_3D.py:20-94 as built by net_factory_3d.py:11-12."""

    def __init__(self, n_classes=2, in_channels=1, feature_scale=4):
        self.n_classes, self.in_channels = n_classes, in_channels
        self.f = [int(x / feature_scale) for x in (64, 128, 256, 512, 1024)]

    def _uc_keys(self, prefix, cin, cout):
        return [(f"{prefix}.conv1.0.weight", (cout, cin, 3, 3, 3)), (f"{prefix}.conv1.0.bias", (cout,)),
                (f"{prefix}.conv2.0.weight", (cout, cout, 3, 3, 3)), (f"{prefix}.conv2.0.bias", (cout,))]

    def spec(self):
        f = self.f
        keys, cin = [], self.in_channels
        for n, co in zip(["conv1", "conv2", "conv3", "conv4", "center"], f):
            keys += self._uc_keys(n, cin, co)
            cin = co
        for lvl in (4, 3, 2, 1):
            keys += self._uc_keys(f"up_concat{lvl}.conv", f[lvl] + f[lvl - 1], f[lvl - 1])
        keys += [("final.weight", (self.n_classes, f[0], 1, 1, 1)), ("final.bias", (self.n_classes,))]
        return keys

    def new_state(self):
        return OrderedDict((n, torch.zeros(s)) for n, s in self.spec())

    @staticmethod
    def is_param(name):
        return True

    # UnetConv3, networks/utils.py:99-123 (is_batchnorm=True branch: InstanceNorm3d + ReLU)
    def _unetconv(self, sd, prefix, x):
        for sub in ("conv1", "conv2"):
            x = F.conv3d(x, sd[f"{prefix}.{sub}.0.weight"], sd[f"{prefix}.{sub}.0.bias"], padding=1)
            x = F.relu(_pre_act(F.instance_norm(x, eps=1e-5)))
        return x

    def forward(self, sd, x, training=True, drop=None):
        skips = []
        for n in ("conv1", "conv2", "conv3", "conv4"):
            x = self._unetconv(sd, n, x)
            skips.append(x)
            x = F.max_pool3d(x, 2)
        x = self._unetconv(sd, "center", x)
        x = _dropout(x, 0.3, training, drop, 0)                                      # unet_3D.py:85
        for lvl in (4, 3, 2, 1):                                                     # UnetUp3_CT, utils.py:270-276
            up = F.interpolate(x, scale_factor=(2, 2, 2), mode="trilinear")
            skip = skips[lvl - 1]
            off = up.size(2) - skip.size(2)
            skip = F.pad(skip, 2 * [off // 2, off // 2, 0])
            x = self._unetconv(sd, f"up_concat{lvl}.conv", torch.cat([skip, up], 1))
        x = _dropout(x, 0.3, training, drop, 1)                                      # unet_3D.py:90
        return F.conv3d(x, sd["final.weight"], sd["final.bias"])

    def drop_sites(self, in_shape):
        N, _, D, H, W = in_shape
        return [(0, 0.3, (N, self.f[4], D >> 4, H >> 4, W >> 4)), (1, 0.3, (N, self.f[0], D, H, W))]


class OracleVNet:
    """reference code/networks/vnet.py:145-239 (has_dropout=True); ``normalization`` as in vnet.py:15-22:
    'batchnorm' (what net_factory_3d.py:18-20 builds), 'groupnorm' (GroupNorm(16)), 'instancenorm', 'none'."""

    def __init__(self, n_classes=2, n_channels=1, n_filters=16, normalization='batchnorm'):
        assert normalization in ('batchnorm', 'groupnorm', 'instancenorm', 'none')
        self.normalization = normalization
        self.n_classes, self.n_channels, self.nf = n_classes, n_channels, n_filters
        f = n_filters
        # (module name, kind, n_stages, cin, cout) in registration order, vnet.py:150-175
        self.layout = [
            ("block_one", "conv", 1, n_channels, f), ("block_one_dw", "down", 1, f, 2 * f),
            ("block_two", "conv", 2, 2 * f, 2 * f), ("block_two_dw", "down", 1, 2 * f, 4 * f),
            ("block_three", "conv", 3, 4 * f, 4 * f), ("block_three_dw", "down", 1, 4 * f, 8 * f),
            ("block_four", "conv", 3, 8 * f, 8 * f), ("block_four_dw", "down", 1, 8 * f, 16 * f),
            ("block_five", "conv", 3, 16 * f, 16 * f), ("block_five_up", "up", 1, 16 * f, 8 * f),
            ("block_six", "conv", 3, 8 * f, 8 * f), ("block_six_up", "up", 1, 8 * f, 4 * f),
            ("block_seven", "conv", 3, 4 * f, 4 * f), ("block_seven_up", "up", 1, 4 * f, 2 * f),
            ("block_eight", "conv", 2, 2 * f, 2 * f), ("block_eight_up", "up", 1, 2 * f, f),
            ("block_nine", "conv", 1, f, f),
        ]

    def spec(self):
        keys = []
        for name, kind, stages, cin, cout in self.layout:
            for s in range(stages):
                ci = cin if s == 0 else cout
                wshape = {"conv": (cout, ci, 3, 3, 3), "down": (cout, ci, 2, 2, 2), "up": (ci, cout, 2, 2, 2)}[kind]
                c, bn = self._keys(name, s)
                keys += [(c + ".weight", wshape), (c + ".bias", (cout,))]
                if self.normalization in ('batchnorm', 'groupnorm'):
                    keys += [(bn + ".weight", (cout,)), (bn + ".bias", (cout,))]
                if self.normalization == 'batchnorm':
                    keys += [(bn + ".running_mean", (cout,)), (bn + ".running_var", (cout,)),
                             (bn + ".num_batches_tracked", ())]
        keys += [("out_conv.weight", (self.n_classes, self.nf, 1, 1, 1)), ("out_conv.bias", (self.n_classes,))]
        return keys

    new_state = OracleUNet2D.new_state
    is_param = staticmethod(OracleUNet2D.is_param)

    def _keys(self, name, s):
        k = 2 if self.normalization == 'none' else 3      # modules per stage of the nn.Sequential
        return f"{name}.conv.{k * s}", f"{name}.conv.{k * s + 1}"

    def _bn_relu(self, sd, bn, x, training):
        if self.normalization == 'batchnorm':
            if training:
                sd[bn + ".num_batches_tracked"] += 1
            x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                             sd[bn + ".bias"], training, 0.1, 1e-5)
        elif self.normalization == 'groupnorm':                                       # vnet.py:19-20
            x = F.group_norm(x, 16, sd[bn + ".weight"], sd[bn + ".bias"], 1e-5)
        elif self.normalization == 'instancenorm':                                    # vnet.py:21-22
            x = F.instance_norm(x, eps=1e-5)
        return F.relu(_pre_act(x))

    def _block(self, sd, name, kind, stages, x, training):
        for s in range(stages):
            c, bn = self._keys(name, s)
            if kind == "conv":                                                        # ConvBlock, vnet.py:5-31
                x = F.conv3d(x, sd[c + ".weight"], sd[c + ".bias"], padding=1)
            elif kind == "down":                                                      # DownsamplingConvBlock :67-91
                x = F.conv3d(x, sd[c + ".weight"], sd[c + ".bias"], stride=2)
            else:                                                                     # UpsamplingDeconvBlock :94-118
                x = F.conv_transpose3d(x, sd[c + ".weight"], sd[c + ".bias"], stride=2)
            x = self._bn_relu(sd, bn, x, training)
        return x

    def forward(self, sd, x, training=True, drop=None):
        feats = {}
        skip_of = {"block_six": "block_four", "block_seven": "block_three", "block_eight": "block_two",
                   "block_nine": "block_one"}
        for name, kind, stages, cin, cout in self.layout:
            if name in skip_of:
                x = x + feats[skip_of[name]]                                          # vnet.py:210,214,218,222
            x = self._block(sd, name, kind, stages, x, training)
            if name == "block_five":
                x = self._dropout3d(x, training, drop, 0)                             # vnet.py:195-196
            if name == "block_nine":
                x = self._dropout3d(x, training, drop, 1)                             # vnet.py:225-226
            feats[name] = x
        return F.conv3d(x, sd["out_conv.weight"], sd["out_conv.bias"])

    @staticmethod
    def _dropout3d(x, training, drop, site):
        if not training or drop == "off":
            return x
        if isinstance(drop, dict):
            return x * drop[site]                      # [N,C,1,1,1] scale mask: whole feature maps
        return F.dropout3d(x, 0.5, True)

    def drop_sites(self, in_shape):
        """Dropout3d draws one Bernoulli per (sample, channel) feature map: masks are [N,C,1,1,1]."""
        N = in_shape[0]
        return [(0, 0.5, (N, 16 * self.nf, 1, 1, 1)), (1, 0.5, (N, self.nf, 1, 1, 1))]

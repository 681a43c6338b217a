"""V-Net (``--model vnet``) on hand-written gfx950 kernels.

Drop-in for the reference's ``networks.vnet.VNet`` (code/networks/vnet.py:145-239): same constructor, same
``forward(x[N,C,D,H,W])``, same state_dict keys (``block_one.conv.{0,1}.*``, ``block_one_dw.conv.{0,1}.*``,
``block_two.conv.{0,1,3,4}.*`` ... ``block_five_up.conv.{0,1}.*`` ... ``out_conv.*``).  ``net_factory_3d`` builds it
with ``normalization='batchnorm', has_dropout=True`` (net_factory_3d.py:18-20).

Layers: 3x3x3 conv + norm + ReLU stages (1/2/3/3/3 encoder, 3/3/2/1 decoder), kernel-2 stride-2 down convolutions and
transposed convolutions (executed as space/depth re-layout + the 1x1x1 MFMA conv), additive skips, ``Dropout3d(0.5)``
after block_five and block_nine, 1x1x1 output conv.

``normalization`` selects the block of vnet.py:15-22 (and :73-84, :100-110):
  'batchnorm'     conv + BatchNorm3d + ReLU (keys ``conv.{3s}``, ``conv.{3s+1}.{weight,bias,running_*}``)
  'groupnorm'     conv + GroupNorm(16) + ReLU -- the conv+GN+ReLU block; per-(sample, group) statistics, per-channel
                  affine (keys ``conv.{3s+1}.{weight,bias}``); statistics come from the conv epilogue
  'instancenorm'  conv + InstanceNorm3d (no affine, no running statistics: no keys) + ReLU
  'none'          conv + ReLU (ConvBlock keys ``conv.{2s}``); here the conv biases get their real gradient
"""
import math

import torch

from mis_hip.plan import HipNet


def _default_init(shape):
    """torch's default (transposed) conv init: kaiming_uniform(a=sqrt(5)); fan_in = shape[1] * prod(kernel)."""
    w = torch.empty(*shape)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    fan_in = shape[1] * math.prod(shape[2:])
    bound = 1.0 / math.sqrt(fan_in)
    return w, bound


class VNet(HipNet):
    ndim_spatial = 3

    def __init__(self, n_channels=3, n_classes=2, n_filters=16, normalization='none', has_dropout=False):
        super().__init__()
        if normalization not in ('batchnorm', 'groupnorm', 'instancenorm', 'none'):
            raise AssertionError(f"normalization={normalization!r}")          # `assert False` in vnet.py:24
        if normalization == 'groupnorm' and n_filters % 16:
            raise ValueError("num_channels must be divisible by num_groups")  # nn.GroupNorm(16, n_filters)
        self.normalization = normalization
        self.n_channels, self.n_classes, self.nf, self.has_dropout = n_channels, n_classes, n_filters, has_dropout
        f = n_filters
        # (name, kind, stages, cin, cout) in the reference's registration order
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
        for name, kind, stages, cin, cout in self.layout:
            for s in range(stages):
                ci = cin if s == 0 else cout
                if kind == "conv":
                    shape = (cout, ci, 3, 3, 3)
                elif kind == "down":
                    shape = (cout, ci, 2, 2, 2)
                else:
                    shape = (ci, cout, 2, 2, 2)            # ConvTranspose3d weight layout
                w, bound = _default_init(shape)
                p, bn = self._keys(name, s)
                self._declare(p + ".weight", w)
                self._declare(p + ".bias", torch.empty(cout).uniform_(-bound, bound))
                if normalization in ('batchnorm', 'groupnorm'):
                    self._declare(bn + ".weight", torch.ones(cout))
                    self._declare(bn + ".bias", torch.zeros(cout))
                if normalization == 'batchnorm':
                    self._declare(bn + ".running_mean", torch.zeros(cout), "buffer")
                    self._declare(bn + ".running_var", torch.ones(cout), "buffer")
                    self._declare(bn + ".num_batches_tracked", torch.zeros((), dtype=torch.long), "buffer")
        w, bound = _default_init((n_classes, f, 1, 1, 1))
        self._declare("out_conv.weight", w)
        self._declare("out_conv.bias", torch.empty(n_classes).uniform_(-bound, bound))
        self._materialize()
        # nn.ConvTranspose3d weights (not touched by the reference's kaiming / xavier re-initialisation helpers)
        self.transposed_convs = {f"{name}.conv.0.weight" for name, kind, *_ in self.layout if kind == "up"}

    def _keys(self, name, s):
        """(conv key, norm key) of stage s: nn.Sequential indices, 3 modules per stage (2 without a norm)."""
        k = 2 if self.normalization == 'none' else 3
        return f"{name}.conv.{k * s}", f"{name}.conv.{k * s + 1}"

    def _bn_relu(self, plan, prefix, t, y, drop_p=0.0):
        P, B = self.P, self.B
        if self.normalization == 'batchnorm':
            return plan.norm_act(t, y, per_sample=False, gamma=P(prefix + ".weight"), beta=P(prefix + ".bias"),
                                 running=(B(prefix + ".running_mean"), B(prefix + ".running_var"),
                                          B(prefix + ".num_batches_tracked")),
                                 slope=0.0, drop_p=drop_p, drop3d=True)
        if self.normalization == 'groupnorm':     # nn.GroupNorm(16, C): C/16 consecutive channels per group
            return plan.norm_act(t, y, per_sample=True, gamma=P(prefix + ".weight"), beta=P(prefix + ".bias"),
                                 slope=0.0, drop_p=drop_p, drop3d=True, cg=t.shape[1] // 16)
        if self.normalization == 'instancenorm':
            return plan.norm_act(t, y, per_sample=True, slope=0.0, drop_p=drop_p, drop3d=True)
        return plan.norm_act(t, y, per_sample=False, slope=0.0, drop_p=drop_p, drop3d=True, no_norm=True)

    def _build(self, plan):
        N, C, D, H, W = plan.in_shape
        if C != self.n_channels or D % 16 or H % 16 or W % 16:
            raise RuntimeError(f"VNet input must be [N,{self.n_channels},D,H,W] with D,H,W multiples of 16; got "
                               f"{plan.in_shape}")
        P = self.P
        sp = (D, H, W)
        x = plan.inp
        feats = {}
        drop = 0.5 if self.has_dropout else 0.0
        # a conv bias has a real gradient unless the normalisation group lies inside one channel (BatchNorm,
        # InstanceNorm, GroupNorm with one channel per group): a GroupNorm group of cg > 1 channels is NOT invariant
        # to a per-channel shift
        def bias_grad(cout):
            return self.normalization == 'none' or (self.normalization == 'groupnorm' and cout // 16 > 1)
        for name, kind, stages, cin, cout in self.layout:
            if kind == "conv":
                if name in ("block_six", "block_seven", "block_eight", "block_nine"):
                    skip = {"block_six": "block_four", "block_seven": "block_three", "block_eight": "block_two",
                            "block_nine": "block_one"}[name]
                    x = plan.add(x, feats[skip], fuse=True)       # x_up + skip (vnet.py:210-222)
                for s in range(stages):
                    t = plan.new(cout, sp)
                    ck, nk = self._keys(name, s)
                    plan.conv(x, t, P(ck + ".weight"), P(ck + ".bias"), (3, 3, 3),
                              need_dx=not (name == "block_one" and s == 0), bias_grad=bias_grad(cout))
                    last = s == stages - 1
                    dp = drop if (last and name in ("block_five", "block_nine")) else 0.0   # Dropout3d :176-177,224-225
                    x = self._bn_relu(plan, nk, t, plan.new(cout, sp), dp)
                feats[name] = x
            elif kind == "down":
                sp = tuple(v // 2 for v in sp)
                t = plan.new(cout, sp)
                plan.down_conv(x, t, P(f"{name}.conv.0.weight"), P(f"{name}.conv.0.bias"), bias_grad=bias_grad(cout))
                x = self._bn_relu(plan, f"{name}.conv.1", t, plan.new(cout, sp))
            else:
                sp = tuple(v * 2 for v in sp)
                t = plan.new(cout, sp)
                plan.up_conv(x, t, P(f"{name}.conv.0.weight"), P(f"{name}.conv.0.bias"), bias_grad=bias_grad(cout))
                x = self._bn_relu(plan, f"{name}.conv.1", t, plan.new(cout, sp))
        plan.out = plan.new(self.n_classes, sp)
        plan.conv(x, plan.out, P("out_conv.weight"), P("out_conv.bias"), (1, 1, 1), bias_grad=True)

    def forward(self, input, turnoff_drop=False):
        """``turnoff_drop`` as in the reference (vnet.py:230-238)."""
        if not turnoff_drop:
            return super().forward(input)
        keep = self.dropout_enabled
        self.dropout_enabled = False
        try:
            return super().forward(input)
        finally:
            self.dropout_enabled = keep

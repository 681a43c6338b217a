"""2-D U-Net (``--model unet``) on hand-written gfx950 kernels.

Drop-in for the reference's ``networks.unet.UNet`` (code/networks/unet.py:304-321): same
constructor, same ``forward(x[N,1,H,W]) -> logits[N,C,H,W]``, same state_dict keys/shapes
(``encoder.in_conv.conv_conv.{0,1,4,5}.*``, ``encoder.down{1-4}.maxpool_conv.1.conv_conv.*``,
``decoder.up{1-4}.conv1x1.*``, ``decoder.up{1-4}.conv.conv_conv.*``, ``decoder.out_conv.*``).

Architecture as the reference *instantiates* it: feature channels
[16,32,64,128,256]; every 3x3 conv is followed by BatchNorm2d + LeakyReLU(0.01); Dropout
p = [.05,.1,.2,.3,.5] after the first activation of each encoder block; decoder blocks are
1x1 conv -> bilinear x2 (align_corners=True) -> cat([skip, up]) -> ConvBlock(p=0); 3x3 out_conv.
(UpBlock's ``bilinear`` defaults to True in the reference, unet.py:68-69,129-136, and its Decoder never passes the flag.)
``UNet(..., bilinear=False)`` builds the other branch of UpBlock (unet.py:76-78): ``decoder.up{i}.up`` =
``nn.ConvTranspose2d(C1, C2, kernel_size=2, stride=2)`` in place of ``conv1x1`` + bilinear up-sampling.

The layer graph is executed by ``mis_hip.plan``; nothing here computes on the CPU.
"""
import math

import torch

from mis_hip.plan import HipNet

_FT = [16, 32, 64, 128, 256]
_DROPOUT = [0.05, 0.1, 0.2, 0.3, 0.5]


def _conv_init(cout, cin, k):
    """torch's default Conv init: kaiming_uniform(a=sqrt(5)) weight, U(+-1/sqrt(fan_in)) bias."""
    w = torch.empty(cout, cin, *k)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(cin * math.prod(k))
    b = torch.empty(cout).uniform_(-bound, bound)
    return w, b


class UNet(HipNet):
    ndim_spatial = 2

    def __init__(self, in_chns, class_num, bilinear=True):
        super().__init__()
        self.in_chns, self.class_num, self.bilinear = in_chns, class_num, bool(bilinear)
        ft = _FT
        self._blocks = []   # (prefix, cin, cout, dropout)
        self._declare_block("encoder.in_conv.conv_conv", in_chns, ft[0])
        for i in range(1, 5):
            self._declare_block(f"encoder.down{i}.maxpool_conv.1.conv_conv", ft[i - 1], ft[i])
        for i in range(1, 5):
            c1, c2 = ft[5 - i], ft[4 - i]
            if self.bilinear:
                w, b = _conv_init(c2, c1, (1, 1))
                self._declare(f"decoder.up{i}.conv1x1.weight", w)
                self._declare(f"decoder.up{i}.conv1x1.bias", b)
            else:       # nn.ConvTranspose2d(c1, c2, 2, stride=2): weight [c1][c2][2][2]; torch's fan_in = size(1) * 4
                w = torch.empty(c1, c2, 2, 2)
                torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                bound = 1.0 / math.sqrt(c2 * 4)
                self._declare(f"decoder.up{i}.up.weight", w)
                self._declare(f"decoder.up{i}.up.bias", torch.empty(c2).uniform_(-bound, bound))
            self._declare_block(f"decoder.up{i}.conv.conv_conv", 2 * c2, c2)
        w, b = _conv_init(class_num, ft[0], (3, 3))
        self._declare("decoder.out_conv.weight", w)
        self._declare("decoder.out_conv.bias", b)
        self._materialize()

    def _declare_block(self, prefix, cin, cout):
        for idx, (ci, co) in ((0, (cin, cout)), (4, (cout, cout))):
            w, b = _conv_init(co, ci, (3, 3))
            self._declare(f"{prefix}.{idx}.weight", w)
            self._declare(f"{prefix}.{idx}.bias", b)
            bn = idx + 1
            self._declare(f"{prefix}.{bn}.weight", torch.ones(co))
            self._declare(f"{prefix}.{bn}.bias", torch.zeros(co))
            self._declare(f"{prefix}.{bn}.running_mean", torch.zeros(co), "buffer")
            self._declare(f"{prefix}.{bn}.running_var", torch.ones(co), "buffer")
            self._declare(f"{prefix}.{bn}.num_batches_tracked", torch.zeros((), dtype=torch.long), "buffer")

    # ConvBlock (unet.py:31-47): conv-BN-LeakyReLU-Dropout-conv-BN-LeakyReLU
    def _conv_block(self, plan, prefix, x, cout, sp, drop_p, out, need_dx=True):
        P, B = self.P, self.B
        for idx, dst, p in ((0, None, drop_p), (4, out, 0.0)):
            bn = idx + 1
            t = plan.new(cout, sp)
            plan.conv(x, t, P(f"{prefix}.{idx}.weight"), P(f"{prefix}.{idx}.bias"), (3, 3),
                      need_dx=need_dx, bias_grad=False)
            y = dst if dst is not None else plan.new(cout, sp)
            plan.norm_act(t, y, per_sample=False, gamma=P(f"{prefix}.{bn}.weight"), beta=P(f"{prefix}.{bn}.bias"),
                          running=(B(f"{prefix}.{bn}.running_mean"), B(f"{prefix}.{bn}.running_var"),
                                   B(f"{prefix}.{bn}.num_batches_tracked")),
                          slope=0.01, drop_p=p)
            x, need_dx = y, True
        return x

    def _build(self, plan):
        N, C, D, H, W = plan.in_shape
        if C != self.in_chns or D != 1 or H % 16 or W % 16:
            raise RuntimeError(f"UNet input must be [N,{self.in_chns},H,W] with H,W multiples of 16; got "
                               f"{(N, C, H, W)}")
        ft = _FT
        sp = [(1, H >> l, W >> l) for l in range(5)]
        # concat buffers of the 4 decoder levels: [skip | upsampled]
        cat = [plan.new(2 * ft[l], sp[l]) for l in range(4)]
        skip = [plan.view(cat[l], 0, ft[l]) for l in range(4)]
        upv = [plan.view(cat[l], ft[l], ft[l]) for l in range(4)]
        # encoder
        x = self._conv_block(plan, "encoder.in_conv.conv_conv", plan.inp, ft[0], sp[0], _DROPOUT[0], skip[0],
                             need_dx=False)
        for l in range(1, 5):
            pooled = plan.new(ft[l - 1], sp[l])
            plan.maxpool(x, pooled)
            out = skip[l] if l < 4 else plan.new(ft[4], sp[4])
            x = self._conv_block(plan, f"encoder.down{l}.maxpool_conv.1.conv_conv", pooled, ft[l], sp[l],
                                 _DROPOUT[l], out)
        # decoder (UpBlock, unet.py:65-86)
        for i in range(1, 5):
            l = 4 - i
            if self.bilinear:
                c1 = plan.new(ft[l], sp[l + 1])
                plan.conv(x, c1, self.P(f"decoder.up{i}.conv1x1.weight"), self.P(f"decoder.up{i}.conv1x1.bias"),
                          (1, 1), bias_grad=True)
                plan.upsample(c1, upv[l], align_corners=True)
            else:
                plan.up_conv2d(x, upv[l], self.P(f"decoder.up{i}.up.weight"), self.P(f"decoder.up{i}.up.bias"))
            x = self._conv_block(plan, f"decoder.up{i}.conv.conv_conv", cat[l], ft[l], sp[l], 0.0,
                                 plan.new(ft[l], sp[l]))
        plan.out = plan.new(self.class_num, sp[0])
        plan.conv(x, plan.out, self.P("decoder.out_conv.weight"), self.P("decoder.out_conv.bias"), (3, 3),
                  bias_grad=True)

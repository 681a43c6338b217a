"""Static forward/backward executor for the convolutional segmentation nets on the hot path.

Instead of recording a torch autograd graph per op, a network describes its layers once as a
*plan*: a list of ops over pre-allocated activation buffers.  ``Plan.forward`` walks the list,
``Plan.backward`` walks it in reverse; every op is one or two launches of the gfx950 kernels
through the C-ABI (``mis_hip.ops``).  Consequences:

* no per-step allocation, so the whole training step can be captured in one hipGraph;
* skip connections are channel-slices of the decoder's concat buffer, so ``torch.cat``
  (reference unet.py:85, networks/utils.py:276) never runs;
* parameter gradients land directly in one flat fp32 buffer (the DDP all-reduce bucket and the
  input of the fused SGD+EMA kernel).

torch provides device memory, streams and the ``nn.Module``/state_dict surface only.
"""
import itertools
import math
import os

import torch
import torch.nn as nn

from . import lib as _lib
from . import ops

_net_ids = itertools.count(1)

# conv -> norm pairs: the conv epilogue emits the statistics (mis_conv_fwd_stats); MIS_FUSE_STATS=0 keeps the
# separate statistics pass (A/B timing)
FUSE_CONV_STATS = os.environ.get("MIS_FUSE_STATS", "1") != "0"
# norm + act + dropout + 1x1x1 classifier of the 3-D nets as one pass over the last conv's output (norm_act.hip, mis_norm_head_*)
FUSE_HEAD = os.environ.get("MIS_FUSE_HEAD", "1") != "0"
# first layer (1 -> 16 channels, its input needs no gradient): the normalisation's backward apply pass on the load path of
# the weight-gradient kernel (conv_wgrad_cin1.hip, mis_conv_wgrad_cin1_norm)
FUSE_FIRST = os.environ.get("MIS_FUSE_FIRST", "1") != "0"
# max-pool backward on the load path of the producing block's normalisation backward (mis_norm_act_bwd_pool)
FUSE_POOL = os.environ.get("MIS_FUSE_POOL", "1") != "0"
FUSE_POOL_FWD = os.environ.get("MIS_FUSE_POOL_FWD", "1") != "0"     # ... and the pool's forward inside the apply pass
# weight gradients on a side stream: dW of a layer needs only (x, dy) and nothing of the backward needs dW, so it runs
# beside the data gradient / normalisation backward of the layers below and fills the CUs their launch tails leave idle
WGRAD_STREAM = os.environ.get("MIS_WGRAD_STREAM", "1") != "0"
# first stage of the InstanceNorm + ReLU backward (sum dz, sum dz * x) in the epilogue of the Winograd data-gradient launch
# that produces the gradient at the activation (mis_conv3d_wino_dgrad_norm): the partial-sum pass over da and x is not run
FUSE_DGRAD_NORM = os.environ.get("MIS_FUSE_DGRAD_NORM", "1") != "0"
PROGRESS_SYNC_MAIN = os.environ.get("MIS_PROGRESS_SYNC_MAIN", "0") == "1"
# residual blocks (UNETR / SwinUNETR): normalise + add the shortcut + activate in one pass, forward and backward
NORM_RES = os.environ.get("MIS_NORM_RES", "1") != "0"


class Act:
    """Activation buffer ``[N,C,D,H,W]`` (fp32), optionally a channel-slice of a wider buffer."""

    def __init__(self, shape=None, parent=None, c0=0, C=None, tensor=None):
        self.parent, self.c0 = parent, c0
        self._written = False
        self.g = None
        if tensor is not None:
            self.t = tensor
        elif parent is not None:
            self.t = parent.t[:, c0:c0 + C]
        else:
            self.t = torch.empty(shape, dtype=torch.float32, device="cuda")

    @property
    def shape(self):
        return tuple(self.t.shape)

    def slice(self, c0, C):
        return Act(parent=self, c0=c0, C=C)

    def grad(self):
        if self.g is None:
            if self.parent is not None:
                self.g = self.parent.grad()[:, self.c0:self.c0 + self.t.shape[1]]
            else:
                self.g = torch.empty(self.shape, dtype=torch.float32, device="cuda")
        return self.g

    def _root(self):
        a = self
        while a.parent is not None:
            a = a.parent
        return a

    @property
    def written(self):
        return self._written or self._root()._written

    def mark_written(self):
        self._written = True

    def reset(self):
        self._written = False


class Ctx:
    """Per-call execution context."""

    def __init__(self, training, state=None, drop_masks=None, dropout=True, rng_stream=0):
        self.training = training
        self.rng_stream = rng_stream    # Philox sub-stream of this network (student / teacher differ)
        self.dropout = dropout          # False: every dropout p := 0 (fixture mode)
        self.state = state              # device MisStepState (needed for Philox dropout)
        self.drop_masks = drop_masks    # optional {salt_index: mask tensor} (parity tests)


class ConvOp:
    def __init__(self, x, y, w, b, ksize, need_dx, bias_grad):
        self.x, self.y, self.w, self.b = x, y, w, b
        self.ksize, self.need_dx, self.bias_grad = tuple(ksize), need_dx, bias_grad
        self.cout, self.cin = w.data.shape[0], w.data.shape[1]
        # Winograd F(2^3, 3^3) variant of the forward / of the data gradient (conv_wino.hip), or -1: direct kernel
        N, _, D, H, W = y.shape          # stride 1, 'same': the geometry of x (the plan's input has no buffer yet)
        self.wino_f = ops.conv_wino_select(N, self.cin, self.cout, D, H, W, self.ksize)
        self.wino_b = ops.conv_wino_select(N, self.cout, self.cin, D, H, W, self.ksize) if need_dx else -1
        self.wp = None
        self.wpd = None
        self.batched = False     # True: the plan repacks all conv weights in one launch (Plan._pack)
        self.stat = None         # (partials, stride_channel, stride_image): statistics for the NormActOp that follows
        self.stat_norm = None
        self._dx_tmp = None
        self.fused_into = None   # NormActOp that computes this 1x1x1 classifier in its own pass (Plan._fuse_head)
        self.norm_bwd = None     # NormActOp whose backward apply pass runs on this (first) conv's wgrad load path
        self.dgrad_norm = None   # NormActOp producing x whose backward partial sums this conv's data gradient forms

    def fwd(self, ctx):
        if self.fused_into is not None:
            return
        if not self.batched:
            self.wp = ops.conv_pack(self.w.data, ops.conv_wino_pack_mode(self.wino_f, False) if self.wino_f >= 0 else 0,
                                    out=self.wp)
        # the consumer needs batch statistics unless it is a BatchNorm in eval mode (running statistics)
        stat = self.stat if (self.stat is not None and (self.stat_norm.per_sample or ctx.training)) else None
        ops.conv_fwd(self.x.t, self.wp, None if self.b is None else self.b.data, self.y.t, self.cin, self.cout,
                     self.ksize, stat=stat, wino=self.wino_f)

    def bwd(self, ctx):
        if self.fused_into is not None:
            return               # the NormActOp's backward writes w.grad / b.grad / its own input gradient
        if self.norm_bwd is not None:      # need_dx False, bias gradient exactly 0: the weight gradient is all there is
            n = self.norm_bwd
            ops.conv_wgrad_cin1_norm(self.x.t, n.y.grad(), self.y.t, n.per_sample, n.mean, n.rstd,
                                     None if n.gamma is None else n.gamma.data, None if n.beta is None else n.beta.data,
                                     n.sums, n.slope, self.w.grad)
            return
        dy = self.y.grad()
        side = getattr(ctx, "wgrad_stream", None) if self.need_dx else None
        if side is not None:
            _lib.wait_stream(side, torch.cuda.current_stream())     # dy is final
            with torch.cuda.stream(side):
                run_deferred(ctx)
                ops.conv_wgrad(self.x.t, dy, self.w.grad, self.ksize)
                if self.b is not None and self.bias_grad:
                    ops.channel_sum(dy, self.b.grad)
        else:
            ops.conv_wgrad(self.x.t, dy, self.w.grad, self.ksize)
            if self.b is not None and self.bias_grad:
                ops.channel_sum(dy, self.b.grad)
        # bias_grad False: the conv feeds a normalisation, its bias gradient is exactly 0
        # (sum over a normalisation group of dL/dx vanishes); the flat grad buffer keeps its zeros.
        if self.need_dx:
            if not self.batched:
                self.wpd = ops.conv_pack(self.w.data, ops.conv_wino_pack_mode(self.wino_b, True) if self.wino_b >= 0 else 1,
                                         out=self.wpd)
            if self.x.written:
                # the input has another consumer whose gradient is already there (residual blocks: UNETR's
                # UnetResBlock feeds its input to conv1 AND to the shortcut): data gradient into a scratch, then add
                if self._dx_tmp is None:
                    self._dx_tmp = torch.empty(self.x.shape, dtype=torch.float32, device="cuda")
                ops.conv_fwd(dy, self.wpd, None, self._dx_tmp, self.cout, self.cin, self.ksize, wino=self.wino_b)
                ops.add(self.x.grad(), self._dx_tmp, self.x.grad())
            elif self.dgrad_norm is not None:
                n = self.dgrad_norm
                ops.conv_dgrad_norm(dy, self.wpd, self.x.grad(), self.cout, self.cin, n.x.t, n.mean, n.slope, n.ext_part,
                                    self.wino_b)
                n.ext_ready = True
            else:
                ops.conv_fwd(dy, self.wpd, None, self.x.grad(), self.cout, self.cin, self.ksize, wino=self.wino_b)
            self.x.mark_written()


class NormActOp:
    """BatchNorm(train/eval), InstanceNorm or GroupNorm (``per_sample`` with ``cg`` channels per group and a per-channel
    affine), or no normalisation at all (``no_norm``), then (Leaky)ReLU, then inverted dropout."""

    def __init__(self, x, y, per_sample, gamma, beta, running, slope, drop_p, site, eps=1e-5, momentum=0.1, cg=1,
                 no_norm=False):
        self.x, self.y, self.per_sample = x, y, per_sample
        self.cg, self.no_norm = cg, no_norm
        self.gamma, self.beta, self.running = gamma, beta, running  # running = (mean, var, nbt) or None
        self.slope, self.drop_p, self.site, self.eps, self.momentum = slope, drop_p, site, eps, momentum
        self.salt = 0
        self.fused = None        # (partials, tiles): statistics come from the producing ConvOp's epilogue
        self.drop3d = False      # True: nn.Dropout3d semantics (whole feature maps)
        N, C = x.shape[0], x.shape[1]
        G = N * C // cg if per_sample else C
        self.mean = torch.zeros(G, dtype=torch.float32, device="cuda")     # no_norm: stays (0, 1)
        self.rstd = torch.ones(G, dtype=torch.float32, device="cuda")
        self._p = 0.0
        self._mask = None
        self.pool = None         # MaxPoolOp fed by this op's output whose backward runs inside this op's (Plan.maxpool)
        self.sums = None         # set: backward only reduces (into this [G, 2] buffer); the producing first-layer conv applies
        self.ext_part = None     # [N*C*tiles, 2] partial sums of the backward written by the consuming conv's data-gradient
        self.ext_tiles, self.ext_ready, self.ext_sums = 0, False, None     # launch (Plan._fuse_dgrad_norm)
        self.res = None          # Act added to the normalised value in front of the activation (Plan.norm_res_act)
        self.res_post = False    # True: added behind the activation instead (Plan.add's fusion of V-Net's x_up + skip)
        self.head = None         # 1x1x1 classifier ConvOp computed in this op's pass (Plan._fuse_head); head_w / head_b:
        self.head_w = self.head_b = None     # its parameters (their gradients are written by THIS op: dist.param_progress)

    def fwd(self, ctx):
        if self.no_norm:
            pass
        elif not self.per_sample and not ctx.training:
            ops.norm_stats_from_running(self.running[0], self.running[1], self.eps, self.mean, self.rstd)
        else:
            rm, rv, nbt = self.running if (self.running is not None and ctx.training) else (None, None, None)
            if self.fused is not None:       # the producing conv already left per-tile (sum, sumsq) partials
                N, C = self.x.shape[0], self.x.shape[1]
                S = self.x.shape[2] * self.x.shape[3] * self.x.shape[4]
                # a group of cg channels of one sample = cg * tiles consecutive partials
                ops.norm_stats_finalize(self.fused[0], N, C // self.cg, S * self.cg, self.fused[1] * self.cg,
                                        self.per_sample, self.eps, self.mean, self.rstd, rm, rv, nbt, self.momentum)
            elif self.cg > 1:
                ops.group_norm_stats(self.x.t, self.cg, self.eps, self.mean, self.rstd)
            else:
                ops.norm_stats(self.x.t, self.per_sample, self.eps, self.mean, self.rstd, rm, rv, nbt, self.momentum)
        self._p = self.drop_p if (ctx.training and ctx.dropout) else 0.0
        self._mask = ctx.drop_masks.get(self.site) if (ctx.drop_masks and self._p > 0) else None
        # bit 31 selects channel-wise dropout (nn.Dropout3d) in the kernel
        self.salt = ((ctx.rng_stream & 0x7FFF) << 16) | self.site | (0x80000000 if self.drop3d else 0)
        self._state = ctx.state
        if self._p > 0 and self._mask is None and ctx.state is None:
            raise RuntimeError("dropout is active but no device step state was supplied (Ctx.state)")
        if self.res is not None:
            ops.norm_res_act_fwd(self.x.t, self.res.t, self.y.t, self.per_sample, self.mean, self.rstd,
                                 None if self.gamma is None else self.gamma.data,
                                 None if self.beta is None else self.beta.data, self.slope, post=self.res_post)
            return
        if self.pool is not None and self.pool.fwd_fused:
            pl = self.pool
            ops.norm_act_fwd_pool(self.x.t, self.y.t, pl.y.t, pl.idx, self.per_sample, self.mean, self.rstd,
                                  None if self.gamma is None else self.gamma.data,
                                  None if self.beta is None else self.beta.data, self.slope, self._p, self.salt,
                                  self._state, self._mask, cg=self.cg)
            return
        if self.head is not None:
            h = self.head
            ops.norm_head_fwd(self.x.t, h.y.t, self.per_sample, self.mean, self.rstd,
                              None if self.gamma is None else self.gamma.data,
                              None if self.beta is None else self.beta.data, self.slope,
                              h.w.data.view(h.cout, h.cin), None if h.b is None else h.b.data, self._p, self.salt,
                              self._state, self._mask)
            return
        ops.norm_act_fwd(self.x.t, self.y.t, self.per_sample, self.mean, self.rstd,
                         None if self.gamma is None else self.gamma.data,
                         None if self.beta is None else self.beta.data, self.slope, self._p, self.salt,
                         self._state, self._mask, cg=self.cg)

    def bwd(self, ctx):
        assert not self.x.written
        if self.res is not None:
            r = self.res
            ops.norm_res_act_bwd(self.x.t, r.t, self.y.grad(), self.x.grad(), r.grad(), r.written, self.per_sample,
                                 self.mean, self.rstd, None if self.gamma is None else self.gamma.data,
                                 None if self.beta is None else self.beta.data, self.slope,
                                 None if self.gamma is None else self.gamma.grad,
                                 None if self.beta is None else self.beta.grad, post=self.res_post)
            r.mark_written()
            self.x.mark_written()
            return
        if self.ext_ready:       # the data gradient that produced y.grad() already formed the partial sums
            self.ext_ready = False
            to_sums = self.sums is not None      # first layer: its weight gradient applies the backward itself
            ops.norm_act_bwd_tiles(self.x.t, self.y.grad(), None if to_sums else self.x.grad(), self.mean, self.rstd,
                                   self.slope, self.ext_part, self.ext_tiles, self.sums if to_sums else self.ext_sums)
            if not to_sums:
                self.x.mark_written()
            return
        if self.sums is not None:
            ops.norm_act_bwd_sums(self.x.t, self.y.grad(), self.per_sample, self.mean, self.rstd,
                                  None if self.gamma is None else self.gamma.data,
                                  None if self.beta is None else self.beta.data, self.slope, self.sums,
                                  None if self.gamma is None else self.gamma.grad,
                                  None if self.beta is None else self.beta.grad)
            return
        if self.pool is not None:
            pl = self.pool
            ops.norm_act_bwd_pool(self.x.t, self.y.grad() if self.y.written else None, pl.y.grad(), pl.idx, self.x.grad(),
                                  self.per_sample, self.mean, self.rstd,
                                  None if self.gamma is None else self.gamma.data,
                                  None if self.beta is None else self.beta.data, self.slope, self._p, self.salt,
                                  self._state, self._mask, None if self.gamma is None else self.gamma.grad,
                                  None if self.beta is None else self.beta.grad, cg=self.cg)
            self.x.mark_written()
            return
        if self.head is not None:
            h = self.head
            ops.norm_head_bwd(self.x.t, h.y.grad(), self.x.grad(), self.per_sample, self.mean, self.rstd,
                              None if self.gamma is None else self.gamma.data,
                              None if self.beta is None else self.beta.data, self.slope,
                              h.w.data.view(h.cout, h.cin), h.w.grad.view(h.cout, h.cin),
                              h.b.grad if (h.b is not None and h.bias_grad) else None, self._p, self.salt, self._state,
                              self._mask, None if self.gamma is None else self.gamma.grad,
                              None if self.beta is None else self.beta.grad)
            self.x.mark_written()
            return
        ops.norm_act_bwd(self.x.t, self.y.grad(), self.x.grad(), self.per_sample, self.mean, self.rstd,
                         None if self.gamma is None else self.gamma.data,
                         None if self.beta is None else self.beta.data, self.slope, self._p, self.salt,
                         self._state, self._mask,
                         None if self.gamma is None else self.gamma.grad,
                         None if self.beta is None else self.beta.grad, cg=self.cg, no_norm=self.no_norm)
        self.x.mark_written()


class MaxPoolOp:
    def __init__(self, x, y):
        self.x, self.y = x, y
        self.idx = torch.empty(y.t.numel(), dtype=torch.uint8, device="cuda")
        self.fused_into = None   # NormActOp (producer of x) whose backward adds this pool's gradient on its load path
        self.fwd_fused = False   # ... and whose forward also writes this pool's output and argmax codes

    def fwd(self, ctx):
        if self.fwd_fused:
            return
        ops.maxpool2_fwd(self.x.t, self.y.t, self.idx)

    def bwd(self, ctx):
        if self.fused_into is not None:
            return
        ops.maxpool2_bwd(self.y.grad(), self.idx, self.x.grad(), accumulate=self.x.written)
        self.x.mark_written()


class UpsampleOp:
    def __init__(self, x, y, align_corners):
        self.x, self.y, self.align = x, y, align_corners

    def fwd(self, ctx):
        ops.upsample2_fwd(self.x.t, self.y.t, self.align)

    def bwd(self, ctx):
        ops.upsample2_bwd(self.y.grad(), self.x.grad(), self.align, accumulate=self.x.written)
        self.x.mark_written()


class DownConvOp:
    """nn.Conv3d(Cin, Cout, 2, stride=2) (reference vnet.py:73): on V-Net's large levels read in place from the fine volume
    (conv_k2s2.hip); otherwise space_to_depth + 1x1x1 MFMA conv.  The weight gradient always takes the space-to-depth
    view of x (formed in the backward when the forward ran in place: the teacher never needs it)."""

    def __init__(self, x, y, w, b, bias_grad=False, need_dx=True, in_shape=None):
        self.x, self.y, self.w, self.b, self.bias_grad, self.need_dx = x, y, w, b, bias_grad, need_dx
        N, Cin, D, H, W = in_shape if in_shape is not None else x.shape      # in_shape: x is the plan's (late-bound) input
        self.in_shape = (N, Cin, D, H, W)
        self.cin8, self.cout = 8 * Cin, w.data.shape[0]
        coarse = (D // 2, H // 2, W // 2)
        self.direct = ops.conv_k2s2_eligible(Cin, self.cout, coarse, False)
        self.direct_dx = ops.conv_k2s2_eligible(self.cout, Cin, coarse, True)
        self.direct_wg = ops.conv_k2s2_wgrad_eligible(Cin, self.cout, coarse)
        self._xs = None if self.direct else self._new_xs()
        self._xs_valid = False
        self.dxs = None
        self.wp = self.wpd = None
        self._wT = None

    def _new_xs(self):
        N, Cin, D, H, W = self.in_shape
        return torch.empty((N, self.cin8, D // 2, H // 2, W // 2), dtype=torch.float32, device="cuda")

    def fwd(self, ctx):
        if self.direct:
            self._xs_valid = False
            ops.conv_k2s2_down(self.x.t, self.w.data, None if self.b is None else self.b.data, self.y.t)
            return
        ops.space_to_depth2(self.x.t, self._xs, self.in_shape, True)
        self._xs_valid = True
        if ops.conv1x1_gemm_eligible(self._xs, self.y.t, self.cin8, self.cout):
            # few voxels, many channels (the 12^3 / 6^3 levels): a batched GEMM with the weights contraction-major
            from . import tops
            if self._wT is None:
                self._wT = torch.empty((self.cin8, self.cout), dtype=torch.float32, device="cuda")
            tops.transpose(self.w.data.view(self.cout, self.cin8), self._wT)
            ops.conv1x1_gemm(self._xs, self._wT, None if self.b is None else self.b.data, self.y.t)
            return
        self.wp = ops.conv_pack_raw(self.w.data, self.cout, self.cin8, 1, 0, out=self.wp)
        ops.conv_fwd(self._xs, self.wp, None if self.b is None else self.b.data, self.y.t, self.cin8, self.cout, (1, 1, 1))

    def bwd(self, ctx):
        dy = self.y.grad()
        if self.direct_wg and ops.conv_k2s2_wgrad_operands_ok(dy, self.x.t):
            ops.conv_k2s2_wgrad(dy, self.x.t, self.w.grad.view(-1))
        else:
            if not self._xs_valid:
                if self._xs is None:
                    self._xs = self._new_xs()
                ops.space_to_depth2(self.x.t, self._xs, self.in_shape, True)
            if ops.conv1x1_wgrad_eligible(dy, self._xs):
                ops.conv1x1_wgrad(dy, self._xs, self.w.grad.view(self.cout, self.cin8))
            else:
                ops.conv_wgrad(self._xs, dy, self.w.grad, (1, 1, 1))      # [Cout][8Cin] == [Cout][Cin][2][2][2] in memory
        # bias gradient exactly 0 when the conv feeds a normalisation (see ConvOp)
        if self.bias_grad and self.b is not None:
            ops.channel_sum(dy, self.b.grad)
        if not self.need_dx:
            return
        # the in-place kernels' operand requirements (pointer / batch-stride alignment of a channel-slice view) are checked
        # per call; a view that misses them takes the space-to-depth path instead of failing mid-step
        if self.direct_dx and ops.conv_k2s2_up_output_ok(self.x.grad()):       # dX = ConvTranspose3d(dy), parameter read as [K = Cout][M = 8 Cin]
            ops.conv_k2s2_up(dy, self.w.data, None, self.x.grad(), accumulate=self.x.written)
            self.x.mark_written()
            return
        if self.dxs is None:
            self.dxs = self._new_xs()
        if ops.conv1x1_gemm_eligible(dy, self.dxs, self.cout, self.cin8):
            # dxs[8 Cin][S] = W[Cout][8 Cin]^T . dy[Cout][S]: the parameter is already contraction-major for this product
            ops.conv1x1_gemm(dy, self.w.data.view(self.cout, self.cin8), None, self.dxs)
        else:
            self.wpd = ops.conv_pack_raw(self.w.data, self.cout, self.cin8, 1, 1, out=self.wpd)
            ops.conv_fwd(dy, self.wpd, None, self.dxs, self.cout, self.cin8, (1, 1, 1))
        ops.space_to_depth2(self.dxs, self.x.grad(), self.in_shape, False, accumulate=self.x.written)
        self.x.mark_written()


class UpConvOp:
    """nn.ConvTranspose3d(Cin, Cout, 2, stride=2) (reference vnet.py:100): on V-Net's large levels written in place into
    the fine volume (conv_k2s2.hip); otherwise 1x1x1 MFMA conv to 8*Cout channels (weight stored input-major
    [Cin][Cout*8]) + depth_to_space (+ bias)."""

    def __init__(self, x, y, w, b, bias_grad=False):
        self.x, self.y, self.w, self.b, self.bias_grad = x, y, w, b, bias_grad
        N, Cin, d, h, wd = x.shape
        self.cin, self.cout8 = Cin, 8 * w.data.shape[1]
        self.direct = ops.conv_k2s2_eligible(Cin, w.data.shape[1], (d, h, wd), True)
        self.direct_dx = ops.conv_k2s2_eligible(w.data.shape[1], Cin, (d, h, wd), False)
        self.direct_wg = ops.conv_k2s2_wgrad_eligible(w.data.shape[1], Cin, (d, h, wd))
        self.y8 = None if self.direct else self._new_y8()
        self.dy8 = None
        self.dw8 = None
        self.wp = self.wpd = None
        self._wT = None

    def _new_y8(self):
        N, _, d, h, wd = self.x.shape
        return torch.empty((N, self.cout8, d, h, wd), dtype=torch.float32, device="cuda")

    def fwd(self, ctx):
        if self.direct and ops.conv_k2s2_up_output_ok(self.y.t):
            ops.conv_k2s2_up(self.x.t, self.w.data, None if self.b is None else self.b.data, self.y.t)
            return
        if self.y8 is None:
            self.y8 = self._new_y8()
        if ops.conv1x1_gemm_eligible(self.x.t, self.y8, self.cin, self.cout8):
            # y8[8 Cout][S] = W[Cin][8 Cout]^T . x[Cin][S]: the parameter's own layout is contraction-major
            ops.conv1x1_gemm(self.x.t, self.w.data.view(self.cin, self.cout8), None, self.y8)
        else:
            self.wp = ops.conv_pack_raw(self.w.data, self.cout8, self.cin, 1, 2, out=self.wp)
            ops.conv_fwd(self.x.t, self.wp, None, self.y8, self.cin, self.cout8, (1, 1, 1))
        ops.space_to_depth2(self.y8, self.y.t, self.y.shape, False, bias=None if self.b is None else self.b.data)

    def bwd(self, ctx):
        from . import tops
        if self.bias_grad and self.b is not None:
            ops.channel_sum(self.y.grad(), self.b.grad)
        dyf = self.y.grad()
        if self.direct_wg and self.direct_dx and ops.conv_k2s2_wgrad_operands_ok(self.x.t, dyf):
            ops.conv_k2s2_wgrad(self.x.t, dyf, self.w.grad.view(-1))             # the parameter's own [Cin][8Cout] layout
        else:
            if self.dy8 is None:
                self.dy8 = self._new_y8()
                self.dw8 = torch.empty((self.cout8, self.cin), dtype=torch.float32, device="cuda")
            ops.space_to_depth2(dyf, self.dy8, self.y.shape, True)
            if ops.conv1x1_wgrad_eligible(self.x.t, self.dy8):
                # operands swapped: the result IS the parameter's input-major layout [Cin][8Cout]
                ops.conv1x1_wgrad(self.x.t, self.dy8, self.w.grad.view(self.cin, self.cout8))
            else:
                ops.conv_wgrad(self.x.t, self.dy8, self.dw8, (1, 1, 1))                  # [8Cout][Cin]
                tops.transpose(self.dw8, self.w.grad.view(self.cin, self.cout8))         # parameter is [Cin][8Cout]
        assert not self.x.written
        if self.direct_dx:       # dX = Conv3d(k2s2)(dy) with the parameter read as [M = Cin][K = 8 Cout]
            ops.conv_k2s2_down(self.y.grad(), self.w.data, None, self.x.grad())
            self.x.mark_written()
            return
        if ops.conv1x1_gemm_eligible(self.dy8, self.x.grad(), self.cout8, self.cin):
            # dx[Cin][S] = W[Cin][8 Cout] . dy8[8 Cout][S]: needs the parameter transposed
            if self._wT is None:
                self._wT = torch.empty((self.cout8, self.cin), dtype=torch.float32, device="cuda")
            tops.transpose(self.w.data.view(self.cin, self.cout8), self._wT)
            ops.conv1x1_gemm(self.dy8, self._wT, None, self.x.grad())
        else:
            self.wpd = ops.conv_pack_raw(self.w.data, self.cout8, self.cin, 1, 3, out=self.wpd)
            ops.conv_fwd(self.dy8, self.wpd, None, self.x.grad(), self.cout8, self.cin, (1, 1, 1))
        self.x.mark_written()


class UpConv2dOp:
    """nn.ConvTranspose2d(Cin, Cout, 2, stride=2) -- ``UpBlock(bilinear=False)`` of the 2-D UNet (reference
    networks/unet.py:76-78, :81-84): windows of a kernel-2 / stride-2 transposed convolution do not overlap, so it is a 1x1
    convolution to 4*Cout channels (the parameter [Cin][Cout][2][2] IS the input-major 1x1 weight [Cin][4 Cout]: pack modes
    2 / 3 of the V-Net op) + the 2-D pixel shuffle (+ bias).  Backward: bias sums, un-shuffle of dy, then the 1x1
    convolution's weight / data gradient on the MFMA kernels."""

    def __init__(self, x, y, w, b, bias_grad=True):
        self.x, self.y, self.w, self.b, self.bias_grad = x, y, w, b, bias_grad
        N, Cin, d, h, wd = x.shape
        assert d == 1
        self.cin, self.cout4 = Cin, 4 * w.data.shape[1]
        self.y4 = torch.empty((N, self.cout4, 1, h, wd), dtype=torch.float32, device="cuda")
        self.dy4 = self.dw4 = None
        self.wp = self.wpd = None

    def fwd(self, ctx):
        self.wp = ops.conv_pack_raw(self.w.data, self.cout4, self.cin, 1, 2, out=self.wp)
        ops.conv_fwd(self.x.t, self.wp, None, self.y4, self.cin, self.cout4, (1, 1, 1))
        ops.space_to_depth2d(self.y4, self.y.t, self.y.shape, False, bias=None if self.b is None else self.b.data)

    def bwd(self, ctx):
        from . import tops
        dyf = self.y.grad()
        if self.bias_grad and self.b is not None:
            ops.channel_sum(dyf, self.b.grad)
        if self.dy4 is None:
            self.dy4 = torch.empty_like(self.y4)
            self.dw4 = torch.empty((self.cout4, self.cin), dtype=torch.float32, device="cuda")
        ops.space_to_depth2d(dyf, self.dy4, self.y.shape, True)
        ops.conv_wgrad(self.x.t, self.dy4, self.dw4, (1, 1, 1))                      # [4Cout][Cin]
        tops.transpose(self.dw4, self.w.grad.view(self.cin, self.cout4))             # the parameter is [Cin][4Cout]
        assert not self.x.written
        self.wpd = ops.conv_pack_raw(self.w.data, self.cout4, self.cin, 1, 3, out=self.wpd)
        ops.conv_fwd(self.dy4, self.wpd, None, self.x.grad(), self.cout4, self.cin, (1, 1, 1))
        self.x.mark_written()


class AddOp:
    """out = a + b (the additive skips of V-Net, reference vnet.py:210-222)."""

    def __init__(self, a, b, out):
        self.a, self.b, self.out = a, b, out

    def fwd(self, ctx):
        ops.add(self.a.t, self.b.t, self.out.t)

    def bwd(self, ctx):
        dout = self.out.grad()
        for t in (self.a, self.b):
            if t.written:
                ops.add(t.grad(), dout, t.grad())
            else:
                ops.add(dout, None, t.grad())
            t.mark_written()


class Plan:
    """Layer list + buffers of one network for one input geometry."""

    def __init__(self, net, in_shape):
        self.net = net
        self.in_shape = tuple(in_shape)      # (N, C, D, H, W)
        self._tbatch = None
        self.ops = []
        self.acts = []
        self.inp = Act(tensor=torch.empty(0, device="cuda"))
        self.acts.append(self.inp)
        self._wgrad_stream = None
        self._salt = itertools.count(0)
        self.out = None
        self._packs = None
        self._head_checked = False
        self.generation = 0      # forwards run on this plan (its buffers hold the activations of the LAST one)
        self._progress = None    # dist.param_progress(self.ops, net.flat_grad), built on the first overlapped backward

    # ---- builders used by the networks ----
    def new(self, C, spatial, N=None):
        a = Act((self.in_shape[0] if N is None else N, C) + tuple(spatial))
        self.acts.append(a)
        return a

    def view(self, parent, c0, C):
        a = parent.slice(c0, C)
        self.acts.append(a)
        return a

    def conv(self, x, y, w, b, ksize, need_dx=True, bias_grad=True):
        self.ops.append(ConvOp(x, y, w, b, ksize, need_dx, bias_grad))
        return y

    def norm_act(self, x, y, per_sample, gamma=None, beta=None, running=None, slope=0.0, drop_p=0.0, drop3d=False,
                 cg=1, no_norm=False):
        op = NormActOp(x, y, per_sample, gamma, beta, running, slope, drop_p, next(self._salt), cg=cg, no_norm=no_norm)
        op.drop3d = drop3d
        # conv -> norm: let the conv epilogue produce the statistics (no separate pass over the conv output)
        prev = self.ops[-1] if self.ops else None
        if FUSE_CONV_STATS and not no_norm and type(prev) is ConvOp and prev.y is x:
            N, C, D, H, W = x.shape
            T = ops.conv_stat_tiles(N, prev.cin, prev.cout, D, H, W, prev.ksize, wino=prev.wino_f)
            if T > 0:
                part = torch.empty((C * N * T, 2), dtype=torch.float32, device="cuda")
                prev.stat = (part, T, C * T) if per_sample else (part, N * T, T)
                prev.stat_norm = op
                op.fused = (part, T)
        if (FUSE_FIRST and type(prev) is ConvOp and prev.y is x and not prev.need_dx and prev.cin == 1
                and prev.ksize in ((3, 3, 3), (1, 3, 3)) and drop_p == 0.0 and cg == 1 and not no_norm
                and not (per_sample and (gamma is not None or beta is not None)) and x.parent is None
                and ops.conv_wgrad_cin1_norm_eligible(x.shape[0], prev.cout, *x.shape[2:])):
            op.sums = torch.empty((x.shape[0] * x.shape[1] if per_sample else x.shape[1], 2), dtype=torch.float32,
                                  device="cuda")
            prev.norm_bwd = op
        self.ops.append(op)
        return y

    @staticmethod
    def can_norm_res_act(x):
        """Geometry the one-pass form serves (the float4 kernels): voxel count a multiple of 4."""
        return NORM_RES and (x.shape[2] * x.shape[3] * x.shape[4]) % 4 == 0

    def norm_res_act(self, x, res, y, per_sample, gamma=None, beta=None, running=None, slope=0.0):
        """y = act(norm(x) + res): the tail of a residual block in one pass (ops.norm_res_act_fwd); no dropout."""
        assert self.can_norm_res_act(x) and tuple(res.shape) == tuple(x.shape)
        assert not (per_sample and (gamma is not None or beta is not None))
        self.norm_act(x, y, per_sample, gamma, beta, running, slope)
        self.ops[-1].res = res
        return y

    def down_conv(self, x, y, w, b, bias_grad=False):
        self.ops.append(DownConvOp(x, y, w, b, bias_grad))
        return y

    def up_conv(self, x, y, w, b, bias_grad=False):
        self.ops.append(UpConvOp(x, y, w, b, bias_grad))
        return y

    def up_conv2d(self, x, y, w, b, bias_grad=True):
        self.ops.append(UpConv2dOp(x, y, w, b, bias_grad))
        return y

    def add(self, a, b, out=None, fuse=False):
        """out = a + b (``out`` None: a new buffer of ``a``'s shape, allocated only when an AddOp is really needed).
        ``fuse``: the caller guarantees that ``a`` -- the output of the norm/activation op just appended --
        has no other reader; that op then writes act(norm(x)) + b itself (one pass, forward and backward) and ``a`` is returned."""
        prev = self.ops[-1] if self.ops else None
        if (fuse and NORM_RES and type(prev) is NormActOp and prev.y is a and a.parent is None and prev.res is None
                and prev.pool is None and prev.sums is None and not prev.no_norm and prev.drop_p == 0.0 and prev.cg == 1
                and not (prev.per_sample and (prev.gamma is not None or prev.beta is not None))
                and self.can_norm_res_act(a)):
            prev.res, prev.res_post = b, True        # the sum lands in ``a``; ``out`` is not needed
            if out is not None and out in self.acts:
                self.acts.remove(out)
            return a
        if out is None:
            out = self.new(a.shape[1], a.shape[2:], N=a.shape[0])
        self.ops.append(AddOp(a, b, out))
        return out

    def maxpool(self, x, y):
        op = MaxPoolOp(x, y)
        if FUSE_POOL:
            # x = the output of a norm/act op (possibly a view into a decoder's concat buffer: the skip connection)
            prod = next((o for o in reversed(self.ops) if type(o) is NormActOp and o.y is x), None)
            N, C, D, H, W = x.shape
            if (prod is not None and prod.pool is None and prod.head is None and prod.sums is None and not prod.no_norm
                    and W % 4 == 0 and H % 2 == 0 and (D == 1 or D % 2 == 0) and D * H * W < 2 ** 31):
                prod.pool, op.fused_into = op, prod
                # forward too (2-D: config 2 10.47 -> 10.41 ms; measured neutral on the 3-D volumes, where the fused
                # kernel's four row pairs per thread cost what the saved pass gains: kept on the plain pair there)
                # when the two ops are adjacent and the rows split into 8-float pieces
                op.fwd_fused = FUSE_POOL_FWD and D == 1 and W % 8 == 0 and prod is self.ops[-1] and y.parent is None
        self.ops.append(op)
        return y

    def upsample(self, x, y, align_corners):
        self.ops.append(UpsampleOp(x, y, align_corners))
        return y

    # ---- execution ----
    def _pack(self, mode):
        """Re-layout the weights of every ConvOp for the MFMA kernels in ONE launch (mode 0 before the forward,
        mode 1 = flipped/transposed filters before the backward); weights change with every SGD update."""
        if self._packs is None:
            convs = [op for op in self.ops if type(op) is ConvOp]
            L = ops._l.load()
            fj, bj = [], []
            for op in convs:
                taps = op.w.data[0, 0].numel()
                mf = ops.conv_wino_pack_mode(op.wino_f, False) if op.wino_f >= 0 else 0
                mb = ops.conv_wino_pack_mode(op.wino_b, True) if op.wino_b >= 0 else 1
                op.wp = torch.empty(L.mis_conv_packed_floats(op.cout, op.cin, taps, mf), dtype=torch.float32,
                                    device="cuda")
                fj.append((op.w.data, op.wp, mf))
                if op.need_dx:
                    op.wpd = torch.empty(L.mis_conv_packed_floats(op.cout, op.cin, taps, mb), dtype=torch.float32,
                                         device="cuda")
                    bj.append((op.w.data, op.wpd, mb))
                op.batched = True
            self._packs = (ops.PackBatch(fj) if fj else None, ops.PackBatch(bj) if bj else None)
        if self._packs[mode] is not None:
            self._packs[mode].run()

    def _transpose_linear_weights(self):
        """Mixed plans (UNETR, SwinUNETR): W^T of every token-major Linear that needs a data gradient in ONE launch (as
        SwinPlan.transpose_weights; 136 launches of ~5 us per UNETR backward before)."""
        if self._tbatch is None:
            from . import tops
            from .swin_plan import LinearOp
            jobs = []
            for op in self.ops:
                if isinstance(op, LinearOp) and op.need_dx and op.w2.is_contiguous():
                    op.wT = torch.empty((op.w2.shape[1], op.w2.shape[0]), dtype=torch.float32, device="cuda")
                    op.wT_batched = True
                    jobs.append((op.w2, op.wT))
            self._tbatch = tops.TransposeBatch(jobs) if jobs else False
        if self._tbatch:
            self._tbatch.run()

    def _split_linear_weights(self, transposed):
        """Mixed plans: the token-major Linears' weights as pre-split GEMM operands (swin_plan.split_linear_weights)."""
        if self._tbatch is False:       # no Linear in this plan
            return False
        from .swin_plan import split_linear_weights
        return split_linear_weights(self, self.ops, transposed)

    def _fuse_dgrad_norm(self):
        """conv_a -> InstanceNorm -> ReLU -> conv_b (reference UnetConv3, utils.py:99-123): when the activation between
        the two convolutions has no other reader, conv_b's Winograd data gradient also forms the partial sums of the
        normalisation's backward (ops.conv_dgrad_norm)."""
        if not FUSE_DGRAD_NORM:
            return
        L = ops._l.load()
        for conv in self.ops:
            if type(conv) is not ConvOp or not conv.need_dx or conv.wino_b not in (0, 1) or conv.ksize != (3, 3, 3):
                continue
            if conv.fused_into is not None or conv.x.parent is not None:
                continue
            norm = next((o for o in self.ops if type(o) is NormActOp and o.y is conv.x), None)
            if norm is None or not norm.per_sample or norm.gamma is not None or norm.beta is not None or norm.cg != 1:
                continue
            if norm.no_norm or norm.drop_p > 0 or norm.pool is not None or norm.head is not None or norm.res is not None:
                continue
            N, C = norm.x.shape[0], norm.x.shape[1]
            if N * C > 384:
                continue
            readers = 0
            for op in self.ops:
                if op is norm:
                    continue
                for v in vars(op).values():
                    if isinstance(v, Act) and v._root() is norm.y:
                        readers += 1
            if readers != 1:
                continue
            _, _, D, H, W = norm.x.shape
            tiles = int(L.mis_conv3d_wino_stat_tiles(D, H, W, conv.wino_b))
            norm.ext_part = torch.zeros(N * C * tiles, 2, dtype=torch.float32, device="cuda")
            norm.ext_tiles = tiles
            norm.ext_sums = torch.zeros(N * C, 2, dtype=torch.float32, device="cuda")
            conv.dgrad_norm = norm

    def _fuse_head(self):
        """Last two ops = norm/act(/dropout) -> 1x1x1 classifier, the activation in between read by nobody else: one op."""
        self._head_checked = True
        if not FUSE_HEAD or len(self.ops) < 2:
            return
        norm, conv = self.ops[-2], self.ops[-1]
        if type(norm) is not NormActOp or type(conv) is not ConvOp or conv.ksize != (1, 1, 1) or conv.x is not norm.y:
            return
        if norm.y.parent is not None or conv.y is not self.out or not conv.need_dx or norm.res is not None:
            return
        for op in self.ops[:-1]:
            for v in vars(op).values():
                if isinstance(v, Act) and v is not norm.y and v._root() is norm.y:
                    return
                if v is norm.y and op is not norm:
                    return
        if not ops.norm_head_eligible(conv.cin, conv.cout, norm.per_sample, norm.gamma, norm.beta, norm.cg, norm.no_norm):
            return
        norm.head, norm.head_w, norm.head_b = conv, conv.w, conv.b
        conv.fused_into = norm

    def forward(self, x5, ctx):
        assert tuple(x5.shape) == self.in_shape, (tuple(x5.shape), self.in_shape)
        if not self._head_checked:
            self._fuse_head()
            self._fuse_dgrad_norm()
        self.generation += 1
        self.inp.t = x5
        self._pack(0)
        ctx.b3_fwd = self._split_linear_weights(False)
        for op in self.ops:
            op.fwd(ctx)
        return self.out.t

    def backward(self, dlogits5, ctx, on_progress=None):
        """``on_progress(lo)``: called after every op with the start of the finished suffix of the flat gradient
        buffer (data-parallel runs issue the finished gradient buckets while the backward continues, mis_hip.dist)."""
        for a in self.acts:
            a.reset()
        if dlogits5 is not None:
            self.out.g = dlogits5
        self._pack(1)
        if self._tbatch is None:        # first backward of a mixed plan (UNETR, SwinUNETR): pair its token ops
            from .swin_plan import pair_ln_residual
            pair_ln_residual(self.ops)
        self._transpose_linear_weights()
        ctx.b3_bwd = self._split_linear_weights(True)
        main = torch.cuda.current_stream()
        side = None
        if WGRAD_STREAM:
            if self._wgrad_stream is None:
                self._wgrad_stream = _lib.side_stream("wgrad")
            side = self._wgrad_stream
        ctx.wgrad_stream = side
        ctx.deferred = [] if side is not None else None      # see defer(): finishing launches queued for the side stream
        begin_finals(ctx, self)
        if on_progress is None:
            for op in reversed(self.ops):
                op.bwd(ctx)
            flush_deferred(ctx)
            ctx.deferred = ctx.finals = None
            if side is not None:
                _lib.wait_stream(main, side)      # every weight gradient is in the flat buffer before the optimizer reads it
            return
        if self._progress is None:
            from .dist import param_progress
            self._progress = param_progress(self.ops, self.net.flat_grad)
        # a reported gradient range is complete on (main, side): what was queued is finished before the callback issues anything
        report = progress_reporter(on_progress, main, side, before=lambda: flush_deferred(ctx))
        for i in range(len(self.ops) - 1, -1, -1):
            self.ops[i].bwd(ctx)
            if i == 0 or self._progress[i] != self._progress[i - 1] or i == len(self.ops) - 1:
                report(self._progress[i])
        flush_deferred(ctx)
        ctx.deferred = ctx.finals = None
        if side is not None:
            _lib.wait_stream(main, side)          # every weight gradient is in the flat buffer before the optimizer reads it

    def drop_sites(self):
        """Site ids (keys of ``net.drop_masks``) of the active dropout layers, in forward order."""
        return [op.site for op in self.ops if isinstance(op, NormActOp) and op.drop_p > 0]

    def drop_site_shape(self, site):
        """Activation shape an explicit ``net.drop_masks[site]`` tensor must have."""
        return next(op.y.shape for op in self.ops if isinstance(op, NormActOp) and op.site == site)


# Gradients nothing downstream reads (LayerNorm's dgamma / dbeta, the relative-position-bias-table gradient) are finished on
# the weight-gradient side stream: the data-gradient chain only runs the half that produces dx and the partials, into a
# workspace of the op's own; the finishing launches ride on the next side-stream section.  MIS_DEFER_FINALS=0: all on one stream.
DEFER = os.environ.get("MIS_DEFER_FINALS", "1") != "0"


def defer(ctx, fn):
    """Queue ``fn`` for the side stream (False: there is none in this pass and the caller runs the one-stream form)."""
    if not DEFER or getattr(ctx, "wgrad_stream", None) is None or getattr(ctx, "deferred", None) is None:
        return False
    ctx.deferred.append(fn)
    return True


# Finishing column sums of a backward pass (LayerNorm affine gradients, split-K partials of the Linear weight / bias gradients,
# relative-position-bias tables: ~130 launches of 5 .. 9 us per SwinUnet step) as ColsumJob records run in ONE launch per flush
# (tops.ColsumBatch / mis_colsum_batch) instead of one or two launches each.  MIS_BATCH_FINALS=0: the per-op launches.  A batch
# is also run when the queued partials exceed FINALS_FLUSH_BYTES, so that the sums overlap the rest of the backward instead of
# all landing behind its last kernel.
BATCH_FINALS = os.environ.get("MIS_BATCH_FINALS", "1") != "0"
FINALS_FLUSH_BYTES = int(os.environ.get("MIS_FINALS_FLUSH_MB", "32")) << 20


def begin_finals(ctx, holder):
    """Start a backward pass: ``ctx.finals`` collects ColsumJob records (None: batching is off -- no side stream, or switched
    off); the device job tables are cached on ``holder`` (the plan) by the identity of the jobs of a flush."""
    on = BATCH_FINALS and DEFER and getattr(ctx, "wgrad_stream", None) is not None
    ctx.finals = [] if on else None
    ctx.finals_bytes = 0
    if on and not hasattr(holder, "_final_batches"):
        holder._final_batches = {}
    ctx.final_batches = getattr(holder, "_final_batches", None)


def defer_final(ctx, *jobs):
    """Queue finishing sums for the next batch (False: batching is off and the caller launches its own finishing kernel)."""
    if getattr(ctx, "finals", None) is None:
        return False
    ctx.finals.extend(jobs)
    ctx.finals_bytes += sum(j.bytes for j in jobs)
    return True


def run_finals(ctx):
    """Called with the side stream current and ordered behind the producers of every queued partial."""
    pend = getattr(ctx, "finals", None)
    if pend:
        from . import tops
        key = tuple(id(j) for j in pend)
        batch = ctx.final_batches.get(key)
        if batch is None:
            batch = ctx.final_batches[key] = tops.ColsumBatch(pend)
        batch.run()
        del pend[:]
        ctx.finals_bytes = 0


def run_deferred(ctx):
    """Called with the side stream current and ordered behind everything enqueued on the compute stream so far."""
    pend = getattr(ctx, "deferred", None)
    if pend:
        for fn in pend:
            fn()
        del pend[:]
    if getattr(ctx, "finals_bytes", 0) >= FINALS_FLUSH_BYTES:
        run_finals(ctx)


def flush_deferred(ctx):
    side = getattr(ctx, "wgrad_stream", None)
    if side is not None and (getattr(ctx, "deferred", None) or getattr(ctx, "finals", None)):
        _lib.wait_stream(side, torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run_deferred(ctx)
            run_finals(ctx)


def progress_reporter(on_progress, main, side, before=None):
    """How a backward pass reports ``on_progress(lo)`` when weight gradients run on a side stream.

    The finished suffix includes gradients still in flight on ``side``.  The compute stream must NOT wait for them (that
    would serialise every weight gradient behind the data-gradient chain again, on multi-GPU runs only): instead ``side``
    waits for ``main`` (free: its later work is issued behind main's anyway) and the callback runs with ``side`` as the
    current stream, so whatever it enqueues -- the bucket all-reduce of mis_hip.dist.GradBucketer, which orders itself
    behind the current stream -- sees both streams' work.  A callback that is a bound method of an object with
    ``would_issue(lo)`` (the bucketer) is only called when it would do something: no stream dependency otherwise.
    ``before()`` runs ahead of every callback that IS made (the plans finish their queued parameter-gradient sums there: a
    range that is not reported yet may keep its finishing launches queued, so they batch up between two buckets)."""
    probe = getattr(getattr(on_progress, "__self__", None), "would_issue", None)
    if PROGRESS_SYNC_MAIN:          # A/B switch (scripts/ddp_overhead.py): the compute stream waits at every report
        probe = None

    def report(lo):
        if PROGRESS_SYNC_MAIN and side is not None:
            if before is not None:
                before()
            _lib.wait_stream(main, side)
            _lib.tape_call(on_progress, lo)
            return
        if probe is not None and not probe(lo):
            return
        if before is not None:
            before()
        if side is None:
            _lib.tape_call(on_progress, lo)        # (tape_call: the callback is part of a recorded step, lib.LaunchTape)
            return
        _lib.wait_stream(side, main)
        with torch.cuda.stream(side):
            _lib.tape_call(on_progress, lo)
    return report


class _PRef:
    """Parameter view + its gradient view inside the flat buffers."""

    def __init__(self, data, grad):
        self.data, self.grad = data, grad


class HipNet(nn.Module):
    """Base of the HIP-backed networks: flat parameter storage + plans + the nn.Module surface.

    Sub-classes call ``_declare`` for every parameter/buffer in the reference's state_dict order,
    then ``_materialize``; they implement ``_build(plan)`` describing the layer graph.
    """

    ndim_spatial = 2

    def __init__(self):
        super().__init__()
        self.net_id = next(_net_ids)
        self._specs = []       # (dotted name, shape, kind, init tensor)
        self._refs = {}
        self._plans = {}
        self.flat_param = None
        self.flat_grad = None
        self.step_state = None  # device MisStepState for Philox dropout (set by the step driver)
        self.drop_masks = None  # {site: mask} override for parity tests
        self.rng_stream = self.net_id  # Philox sub-stream; the step driver pins student=1 / teacher=2
        self.dropout_enabled = True  # False: p := 0 at every dropout site (fixture mode)
        self._offsets = {}      # parameter name -> (offset, numel, shape) in the flat buffers
        self._last = None

    # ---- declaration ----
    def _declare(self, name, init, kind="param"):
        self._specs.append((name, tuple(init.shape), kind, init))

    def _materialize(self):
        if not torch.cuda.is_available():
            raise RuntimeError("HIP-backed networks need an MI355X (gfx950) device; there is no CPU fallback. "
                               "Use the modules under oracle/ for CPU reference arithmetic in tests.")
        # every parameter starts on a 16-byte boundary of the flat buffer (float4 / MFMA staging loads);
        # the padding words stay zero in params, grads, momentum and EMA
        total = sum((math.prod(s) + 3) // 4 * 4 for _, s, k, _ in self._specs if k == "param")
        self.flat_param = torch.zeros(total, dtype=torch.float32, device="cuda")
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device="cuda")
        off = 0
        for name, shape, kind, init in self._specs:
            mod, leaf = self._container(name)
            if kind == "param":
                n = init.numel()
                view = self.flat_param[off:off + n].view(shape)
                view.copy_(init)
                p = nn.Parameter(view)
                mod.register_parameter(leaf, p)
                self._refs[name] = _PRef(p.data, self.flat_grad[off:off + n].view(shape))
                self._offsets[name] = (off, n, shape)
                off += (n + 3) // 4 * 4
            else:
                mod.register_buffer(leaf, init.clone().cuda())
        self._specs = [(n, s, k, None) for n, s, k, _ in self._specs]

    def _container(self, dotted):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        return mod, parts[-1]

    def P(self, name):
        return self._refs[name]

    def B(self, name):
        mod, leaf = self._container(name)
        return mod._buffers[leaf]

    # ---- plans ----
    def plan_for(self, shape5):
        key = tuple(shape5)
        plan = self._plans.get(key)
        if plan is None:
            self._check_alias()
            plan = self._new_plan(key)
            self._build(plan)
            self._plans[key] = plan
        return plan

    def _new_plan(self, key):
        return Plan(self, key)

    def _check_alias(self):
        first = next(iter(self.parameters()))
        if first.data_ptr() != self.flat_param.data_ptr():
            raise RuntimeError("parameters were moved/re-allocated (e.g. .to(), .half()); HIP nets keep all "
                               "parameters in one flat fp32 device buffer")

    def _as5(self, x):
        if self.ndim_spatial == 2:
            if x.dim() != 4:
                raise RuntimeError(f"expected [N,C,H,W], got {tuple(x.shape)}")
            return x.unsqueeze(2)
        if x.dim() != 5:
            raise RuntimeError(f"expected [N,C,D,H,W], got {tuple(x.shape)}")
        return x

    def _from5(self, y):
        return y.squeeze(2) if self.ndim_spatial == 2 else y

    def _ctx(self):
        return Ctx(self.training, self.step_state, self.drop_masks, self.dropout_enabled, self.rng_stream)

    def named_flat(self, flat):
        """Yield (parameter name, view into ``flat`` with that parameter's shape) in parameter order."""
        for name, (off, n, shape) in self._offsets.items():
            yield name, flat[off:off + n].view(shape)

    # raw (autograd-free) interface used by the fused training step
    def forward_raw(self, x, no_backward=False):
        """``no_backward``: nobody will differentiate this pass (the EMA teacher, validation): ops may skip what only a
        backward reads -- backward_raw() after such a pass fails loudly."""
        x5 = self._as5(x.contiguous())
        plan = self.plan_for(x5.shape)
        ctx = self._ctx()
        ctx.no_backward = bool(no_backward)
        out = plan.forward(x5, ctx)
        self._last = (plan, ctx)
        return out

    def backward_raw(self, dlogits5=None, on_progress=None):
        plan, ctx = self._last
        if getattr(ctx, "no_backward", False):
            raise RuntimeError("backward_raw() after forward_raw(no_backward=True): that pass did not keep its activations")
        if on_progress is None:
            plan.backward(dlogits5, ctx)
        else:
            plan.backward(dlogits5, ctx, on_progress)

    def logits_grad_buffer(self):
        return self._last[0].out.grad()

    # nn.Module surface: logits = model(x); loss.backward() works through _NetFn
    def forward(self, x):
        if x.dtype != torch.float32 or not x.is_cuda:
            raise RuntimeError("HIP-backed networks take fp32 device tensors (no CPU fallback)")
        params = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _NetFn.apply(self, x, *params)
        return self._from5(self.forward_raw(x)).clone()


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        out = net.forward_raw(x)
        ctx.net = net
        ctx.plan_ctx = net._last
        ctx.generation = net._last[0].generation
        return net._from5(out).clone()

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        plan = ctx.plan_ctx[0]
        if plan.generation != ctx.generation:
            # one static plan (activation buffers, dropout masks) per input shape: a later same-shape forward --
            # a pseudo-label pass under no_grad, a validation pass, a second grad-enabled call -- has overwritten
            # what this backward needs.  torch autograd would keep both graphs; here it must fail loudly.
            raise RuntimeError(
                "backward() of a HIP network whose activations were overwritten: forward was called again with the "
                f"same input shape {plan.in_shape} before this loss.backward() "
                f"(forward #{ctx.generation}, now #{plan.generation}).  Call backward before the next same-shape "
                "forward, or use a second network instance.")
        net._last = ctx.plan_ctx
        net.backward_raw(net._as5(dout.contiguous()))
        grads = tuple(net._refs[n].grad.clone() for n, _, k, _ in net._specs if k == "param")
        return (None, None) + grads

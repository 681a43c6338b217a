"""Static forward/backward executor for the token-major (B*L, C) layers of SwinUnet.

Same idea as ``mis_hip.plan`` (one pre-allocated op list, reverse walk for backward, gradients
straight into the flat buffer), over 2-D token tensors.  Concatenations along C
(``torch.cat([x, skip], -1)``, reference swin_transformer_unet_skip_expand_decoder_sys.py:767) are
column slices of one buffer; roll / window_partition / window_reverse never materialise (they are
token-index arithmetic inside the attention kernel).
"""
import itertools
import os

import torch

from . import lib as _lib
from . import tops
from .plan import Act, begin_finals, defer, defer_final, run_deferred, flush_deferred

# GEMM epilogue fusions (GELU into fc1 / fc2-dX, DropPath + residual add into proj / fc2); MIS_SWIN_FUSE=0 runs the
# separate element-wise passes (A/B timing, and the reference point of the fusion tests)
LN_HEAD = os.environ.get("MIS_LN_HEAD", "1") != "0"        # last LayerNorm + output head in one pass (LnHeadOp)
UNSHUFFLE = os.environ.get("MIS_LN_HEAD_UNSHUFFLE", "1") != "0"   # ... whose backward stores dx through FinalPatchExpand_X4's inverse shuffle
EXPAND_HEAD = os.environ.get("MIS_EXPAND_HEAD", "1") != "0"       # ... and its forward inside the expand GEMM's epilogue
FUSE = int(os.environ.get("MIS_SWIN_FUSE", "7"))      # bit 0: GELU forward, bit 1: GELU backward, bit 2: residual


class TAct:
    """Token activation ``[rows, C]`` (fp32), optionally a column slice of a wider buffer."""

    def __init__(self, rows=None, C=None, parent=None, c0=0):
        self.parent, self.c0 = parent, c0
        self._written = False
        self.g = None
        if parent is not None:
            self.t = parent.t[:, c0:c0 + C]
        else:
            self.t = torch.empty((rows, C), dtype=torch.float32, device="cuda")

    @property
    def rows(self):
        return self.t.shape[0]

    @property
    def C(self):
        return self.t.shape[1]

    def cols(self, c0, C):
        return TAct(parent=self, c0=c0, C=C)

    def grad(self):
        if self.g is None:
            if self.parent is not None:
                self.g = self.parent.grad()[:, self.c0:self.c0 + self.t.shape[1]]
            else:
                self.g = torch.empty_like(self.t)
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


class LinearOp:
    """nn.Linear: y = x W^T + b;  dW = dy^T x, db = colsum(dy), dx (+)= dy W.

    Optional epilogue fusions (set by the network when it builds a block, executed when ``FUSE``):
      ``gelu``     the GeluOp that consumes y: the GEMM writes y AND gelu(y); that op's forward pass is skipped
      ``res``      the ResidualOp that consumes y: the GEMM writes shortcut + DropPath(y) straight into the
                   residual's output (y itself is never written, only its gradient exists)
      ``dx_gelu``  the GeluOp that produced x: the dX GEMM multiplies by gelu'(pre-activation) and writes the gradient
                   of the GELU's input; that op's backward pass is skipped."""

    def __init__(self, x, y, w, b, need_dx=True):
        self.x, self.y, self.w, self.b, self.need_dx = x, y, w, b, need_dx
        self.w2 = w.data.view(w.data.shape[0], -1)
        self.gw2 = w.grad.view(w.grad.shape[0], -1)
        self.wT = None
        self.wT_batched = False      # True: the plan transposes every Linear weight in one launch (SwinPlan.backward)
        self.gelu = self.res = self.dx_gelu = None
        self.b3 = self.bT3 = None    # W / W^T cut into bf16 pieces once per pass (split_linear_weights)
        self._dw_ws, self._dw_jobs = None, {}      # own split-K workspace of the weight gradient + its finishing-sum jobs

    def _b3(self, ctx):
        return self.b3 if getattr(ctx, "b3_fwd", False) else None

    def _bT3(self, ctx):
        return self.bT3 if getattr(ctx, "b3_bwd", False) else None

    def fwd(self, ctx):
        bias = None if self.b is None else self.b.data
        if (FUSE & 1) and self.gelu is not None:
            pre = None if getattr(ctx, "no_backward", False) else self.y.t      # only the backward reads the pre-activation
            if tops.gemm_ex(self.x.t, self.w2, pre, tops.EP_GELU_FWD, bias=bias, C2=self.gelu.y.t, b3=self._b3(ctx)):
                self.gelu.skip_fwd = True
                return
        if (FUSE & 4) and self.res is not None:
            r = self.res
            scales = r.prepare(ctx)
            ln = r.ln_next if LN_EPILOGUE else None
            if ln is not None and tops.gemm_residual_ln(self.x.t, self._b3(ctx), r.out.t, r.a.t, ln.g.data, ln.b.data, ln.y.t,
                                                        ln.mean, ln.rstd, bias=bias, rowscale=scales, rows_per_scale=r.rps):
                r.skip_fwd = True        # the residual add AND the LayerNorm that reads it ran in this GEMM's epilogue
                ln.skip_fwd = True
                return
            if tops.gemm_ex(self.x.t, self.w2, r.out.t, tops.EP_RESIDUAL, bias=bias, E1=r.a.t, rowscale=scales,
                            rows_per_scale=r.rps, b3=self._b3(ctx)):
                r.skip_fwd = True
                return
        tops.gemm(self.x.t, self.w2, self.y.t, bias=bias, b3=self._b3(ctx))

    def _wgrad(self, dy, ctx=None):
        db = None if self.b is None else self.b.grad
        if ctx is not None and getattr(ctx, "finals", None) is not None and self.gw2.is_contiguous() and \
                self.gw2.shape[0] % 4 == 0 and self.gw2.shape[1] % 4 == 0:
            # split contraction: the k-slices' partials stay in a workspace of this layer's own and their sums join the pass's
            # batch of finishing sums (plan.defer_final) -- no gemm_reduce launch per Linear
            M, N = self.gw2.shape
            if self._dw_ws is None:
                ws = tops.gemm_dw_workspace(M, N, dy.shape[0])
                self._dw_ws = False if ws is None else ws
            ws = self._dw_ws if self._dw_ws is not False else None
            slices = tops.gemm_dw_parts(dy, self.x.t, self.gw2, db, ws)
            if slices:
                jobs = self._dw_jobs.get(slices)
                if jobs is None:
                    jobs = [tops.ColsumJob(ws, 0, M * N, slices, M * N, 2, self.gw2.view(-1))]
                    if db is not None:
                        jobs.append(tops.ColsumJob(ws, slices * M * N, M, slices, M, 2, db))
                    self._dw_jobs[slices] = jobs
                defer_final(ctx, *jobs)
            return
        if db is not None:
            tops.gemm_dw(dy, self.x.t, self.gw2, db)      # db rides on the dW GEMM's read of dy
        else:
            tops.gemm(dy, self.x.t, self.gw2, trans=True)

    def bwd(self, ctx):
        dy = self.y.grad()
        # dW / db need only (x, dy) and nothing downstream needs them: on the plan's side stream they run beside the
        # dX chain (mis_hip.plan.WGRAD_STREAM)
        side = getattr(ctx, "wgrad_stream", None) if self.need_dx else None
        if side is not None:
            _lib.wait_stream(side, torch.cuda.current_stream())
            with torch.cuda.stream(side):
                run_deferred(ctx)
                self._wgrad(dy, ctx)
        else:
            self._wgrad(dy)
        if self.need_dx:
            if not self.wT_batched:
                if self.wT is None:
                    self.wT = torch.empty((self.w2.shape[1], self.w2.shape[0]), dtype=torch.float32, device="cuda")
                tops.transpose(self.w2, self.wT)
            g = self.dx_gelu
            if (FUSE & 2) and g is not None and not g.x.written and \
                    tops.gemm_ex(dy, self.wT, g.x.grad(), tops.EP_GELU_BWD, E1=g.x.t, b3=self._bT3(ctx)):
                g.skip_bwd = True        # d(pre-activation) is written; d(gelu output) never exists
                g.x.mark_written()
                return
            tops.gemm(dy, self.wT, self.x.grad(), accumulate=self.x.written, b3=self._bT3(ctx))
            self.x.mark_written()


# LayerNorm backward + the backward of the residual add in front of it in one pass (MIS_SWIN_LNRES=0: two passes)
LNRES = os.environ.get("MIS_SWIN_LNRES", "1") != "0"
# 96-channel blocks: the LayerNorm that follows a residual add runs in the producing GEMM's epilogue (register-A kernels, round 6)
LN_EPILOGUE = os.environ.get("MIS_LN_EPILOGUE", "1") != "0"


def pair_ln_residual(ops_list):
    """``x = shortcut + drop_path(branch)`` directly followed by ``norm(x)``: every other reader of x comes later in the forward,
    i.e. has left its gradient in x.grad by the time the LayerNorm's backward runs, and nothing runs between that backward and
    the residual's -- the LayerNorm's pass can finish the residual's backward."""
    for i in range(len(ops_list) - 1):
        r, ln = ops_list[i], ops_list[i + 1]
        if isinstance(r, ResidualOp) and isinstance(ln, LayerNormOp) and ln.x is r.out:
            ln.res_bwd = r
            r.ln_next = ln           # forward: the producing GEMM's epilogue may run this LayerNorm (LinearOp.fwd)


def split_linear_weights(holder, ops_list, transposed):
    """Every Linear weight of the op list (``transposed``: its W^T, which the plan's transpose batch just wrote) cut into the
    bf16 piece planes the NT GEMMs consume, in one launch; the weights changed with the last SGD / EMA update, so once per
    forward and once per backward.  Returns True when the ops' ``b3`` / ``bT3`` are current (False: bf16x3 products are off)."""
    if not tops.split_active():
        return False
    key = "_split_bwd" if transposed else "_split_fwd"
    batch = getattr(holder, key, None)
    if batch is None:
        splits = []
        for op in ops_list:
            if not isinstance(op, LinearOp) or not op.w2.is_contiguous():
                continue
            n_out, k_in = op.w2.shape
            if k_in % 4 or n_out % 4 or 6 * n_out * (k_in + 31) >= 1 << 31:
                continue
            # planes in the element order of the kernel that will read them: the register-A kernel (many token rows, float4
            # epilogue) wants the natural order; the pixel-shuffle / LayerNorm-head stores of the expand layers stay staged
            # (the final expand with LayerNorm + head in its epilogue has a register-A form: LnHeadOp.eligible shapes, K <= 96)
            staged_only = isinstance(op, ExpandLinearOp) and not (EXPAND_HEAD and op.ln_head is not None and op.w2.shape[1] <= 96 and op.geo[3] == 96)
            if transposed:
                if op.need_dx and op.wT_batched:
                    op.bT3 = tops.SplitB(op.wT, rows=op.y.rows)
                    splits.append(op.bT3)
            else:
                op.b3 = tops.SplitB(op.w2, rows=None if staged_only else op.x.rows)
                splits.append(op.b3)
        batch = tops.SplitBatch(splits) if splits else False
        setattr(holder, key, batch)
    if batch:
        batch.run()
    return bool(batch)


class LayerNormOp:
    def __init__(self, x, y, g, b):
        self.x, self.y, self.g, self.b = x, y, g, b
        self.mean = torch.empty(x.rows, dtype=torch.float32, device="cuda")
        self.rstd = torch.empty(x.rows, dtype=torch.float32, device="cuda")
        self._ws = None
        self._job = None
        self.skip_fwd = False
        self.res_bwd = None      # the ResidualOp right in front of this op whose output is x (pair_ln_residual)

    def fwd(self, ctx):
        if self.skip_fwd:           # ran in the epilogue of the GEMM that produced x (LinearOp.fwd, LN_EPILOGUE)
            self.skip_fwd = False
            return
        tops.layernorm_fwd(self.x.t, self.y.t, self.g.data, self.b.data, self.mean, self.rstd)

    def bwd(self, ctx):
        r = self.res_bwd if LNRES else None
        if r is not None and not r.y.written and self.g.grad is not None and self.b.grad is not None:
            # the residual add that produced x: its backward rides on this pass (mis_layernorm_bwd_residual_parts); x's own
            # gradient buffer is read (what x's other readers left there) but not written again
            ok, scales = r.bwd_scales(ctx)
            if ok:
                if self._ws is None:
                    self._ws = tops.colreduce_workspace(self.x.rows, self.x.t.shape[1])
                if tops.layernorm_bwd_residual_parts(self.x.t, self.y.grad(), self.x.grad() if self.x.written else None,
                                                     r.a.grad(), r.y.grad(), self.g.data, self.mean, self.rstd, self._ws,
                                                     rowscale=scales, rows_per_scale=r.rps, accumulate_shortcut=r.a.written):
                    r.a.mark_written()
                    r.y.mark_written()
                    r.skip_bwd = True
                    if not self._defer_final(ctx):
                        self._final()
                    return
        if (self.g.grad is not None or self.b.grad is not None) and self._can_defer(ctx):
            if self._ws is None:
                self._ws = tops.colreduce_workspace(self.x.rows, self.x.t.shape[1])
            tops.layernorm_bwd_parts(self.x.t, self.y.grad(), self.x.grad(), self.g.data, self.mean, self.rstd, self._ws,
                                     accumulate_dx=self.x.written)
            self._defer_final(ctx)
        else:
            tops.layernorm_bwd(self.x.t, self.y.grad(), self.x.grad(), self.g.data, self.mean, self.rstd, self.g.grad,
                               self.b.grad, accumulate_dx=self.x.written)
        self.x.mark_written()

    def _final(self):
        tops.layernorm_bwd_final(self._ws, self.x.rows, self.x.t.shape[1], self.g.grad, self.b.grad)

    @staticmethod
    def _can_defer(ctx):
        from .plan import DEFER
        return DEFER and getattr(ctx, "wgrad_stream", None) is not None and getattr(ctx, "deferred", None) is not None

    def _defer_final(self, ctx):
        """The affine gradients from the partial rows in ``_ws``: a job of the pass's batch of finishing sums, else the deferred
        per-op launch; False: neither (the caller finishes on its own stream)."""
        if getattr(ctx, "finals", None) is not None and self.g.grad is not None and self.b.grad is not None:
            if self._job is None:
                C = self.x.t.shape[1]
                self._job = tops.ColsumJob(self._ws.view(torch.float32), 0, C, tops.colreduce_slabs(self.x.rows), C, True,
                                           self.g.grad, self.b.grad)
            return defer_final(ctx, self._job)
        return defer(ctx, self._final)


class GeluOp:
    def __init__(self, x, y):
        self.x, self.y = x, y
        self.skip_fwd = self.skip_bwd = False     # set for one pass by the LinearOp that fused this op's work

    def fwd(self, ctx):
        if self.skip_fwd:
            self.skip_fwd = False
            return
        tops.gelu(self.x.t, self.y.t)

    def bwd(self, ctx):
        if self.skip_bwd:
            self.skip_bwd = False
            return
        assert not self.x.written
        tops.gelu(self.x.t, self.x.grad(), dy=self.y.grad())
        self.x.mark_written()


class AttnOp:
    def __init__(self, qkv, out, table, B, H, W, nH, shift, window=7):
        self.qkv, self.out, self.table = qkv, out, table
        self.geo = (B, H, W, nH, shift)
        self.scale = 32 ** -0.5
        self.window = window
        self._ws = None
        self._job = None

    def fwd(self, ctx):
        tops.window_attention_fwd(self.qkv.t, self.out.t, self.table.data, *self.geo, self.scale, window=self.window)

    def bwd(self, ctx):
        assert not self.qkv.written
        if LayerNormOp._can_defer(ctx):
            if self._ws is None:
                self._ws = tops.window_attention_workspace(*self.geo[:4], window=self.window)
            tops.window_attention_bwd_parts(self.qkv.t, self.out.grad(), self.qkv.grad(), self.table.data, self._ws,
                                            *self.geo, self.scale, window=self.window)
            if getattr(ctx, "finals", None) is not None and self.table.grad.is_contiguous():
                if self._job is None:
                    shape = tops.window_attention_table_partials(*self.geo[:4], window=self.window)
                    self._job = False if shape is None else tops.ColsumJob(self._ws.view(torch.float32), 0, shape[1], shape[0],
                                                                           shape[1], False, self.table.grad.view(-1))
                if self._job:
                    defer_final(ctx, self._job)
                    self.qkv.mark_written()
                    return
            defer(ctx, self._dtable)
        else:
            tops.window_attention_bwd(self.qkv.t, self.out.grad(), self.qkv.grad(), self.table.data, self.table.grad,
                                      *self.geo, self.scale, window=self.window)
        self.qkv.mark_written()

    def _dtable(self):
        tops.window_attention_dtable(self._ws, self.table.grad, *self.geo[:4], window=self.window)


class ResidualOp:
    """out = a + DropPath(y)  (timm DropPath: per-sample Bernoulli(1-p) / (1-p), training only)."""

    def __init__(self, a, y, out, rows_per_sample, drop_p, site):
        self.a, self.y, self.out = a, y, out
        self.rps, self.drop_p, self.site = rows_per_sample, drop_p, site
        self._p, self._salt, self._state, self._scale = 0.0, 0, None, None
        self.plan = None            # set by SwinPlan.add: the per-forward DropPath scale table lives there
        self.skip_fwd = self.skip_bwd = False
        self.ln_next = None         # the LayerNormOp directly behind this op that reads ``out`` (pair_ln_residual)

    def prepare(self, ctx):
        """Fix this pass's DropPath parameters (also used by backward); returns the per-sample scale vector a fused
        GEMM epilogue multiplies the branch by (None: all ones)."""
        self._p = self.drop_p if (ctx.training and ctx.dropout) else 0.0
        self._scale = ctx.drop_masks.get(self.site) if (ctx.drop_masks and self._p > 0) else None
        self._salt = ((ctx.rng_stream & 0xFFFF) << 16) | self.site
        self._state = ctx.state
        if self._p > 0 and self._scale is None and ctx.state is None:
            raise RuntimeError("DropPath is active but no device step state was supplied")
        if self._p <= 0:
            return None
        return self._scale if self._scale is not None else self.plan.droppath_scales(ctx)[self.site]

    def fwd(self, ctx):
        if self.skip_fwd:           # the producing GEMM wrote shortcut + DropPath(branch) already
            self.skip_fwd = False
            return
        self.prepare(ctx)
        tops.residual_fwd(self.a.t, self.y.t, self.out.t, self.rps, self._p, self._salt, self._state, self._scale)

    def bwd_scales(self, ctx):
        """(usable, per-sample DropPath scales of this pass as a device vector or None = all ones) for a kernel that fuses this
        op's backward."""
        if self._p <= 0:
            return True, None
        if self._scale is not None:
            return True, self._scale
        if self.plan is None:
            return False, None
        return True, self.plan.droppath_scales(ctx)[self.site]

    def bwd(self, ctx):
        if self.skip_bwd:           # the LayerNormOp that reads ``out`` wrote a.grad and y.grad in its own backward pass
            self.skip_bwd = False
            return
        assert not self.y.written
        tops.residual_bwd(self.out.grad(), self.a.grad(), self.y.grad(), self.rps, self._p, self._salt, self._state,
                          self._scale, accumulate_shortcut=self.a.written)
        self.a.mark_written()
        self.y.mark_written()


class ExpandLinearOp(LinearOp):
    """PatchExpand / FinalPatchExpand_X4: Linear (no bias) + pixel shuffle.  Forward: one GEMM whose epilogue writes
    the shuffled layout (``mis_gemm_expand``) -- the token-major expanded tensor ``y`` is not materialised; shapes the
    fused form does not cover fall back to gemm + rearrange.  Backward: un-shuffle the gradient into ``y``'s layout,
    then the Linear backward (dW, dx) as usual."""

    def __init__(self, x, y, sh, w, geo):
        super().__init__(x, y, w, None)
        self.sh, self.geo = sh, geo            # geo = (B, H, W, c, P)
        self.unshuffled_by_consumer = False    # set for one backward by the LnHeadOp that consumes ``sh``
        self.ln_head = None                    # the LnHeadOp that consumes ``sh`` (set by the network): fused forward

    def fwd(self, ctx):
        B, H, W, c, P = self.geo
        h = self.ln_head if EXPAND_HEAD else None
        keep = None if getattr(ctx, "no_backward", False) else self.sh.t     # the shuffled tokens: only the backward reads them
        if h is not None and tops.gemm_expand_ln_head(self.x.t, self.w2, keep, B, H, W, P, c, h.g.data, h.b.data, h.w2,
                                                      h.mean, h.rstd, h.logits.t, b3=self._b3(ctx)):
            h.skip_fwd = True        # LayerNorm + output head ran in this GEMM's epilogue
            return
        if not tops.gemm_expand(self.x.t, self.w2, self.sh.t, B, H, W, P, c, b3=self._b3(ctx)):
            super().fwd(ctx)
            tops.token_rearrange(self.y.t, self.sh.t, B, H, W, c, P, 1)

    def bwd(self, ctx):
        B, H, W, c, P = self.geo
        if self.unshuffled_by_consumer:       # LnHeadOp stored its dx through the inverse shuffle: y.grad is complete
            self.unshuffled_by_consumer = False
        else:
            assert not self.y.written
            tops.token_rearrange(self.sh.grad(), self.y.grad(), B, H, W, c, P, 1, inverse=True)
            self.y.mark_written()
        super().bwd(ctx)


class RearrangeOp:
    def __init__(self, src, dst, B, H, W, C, P, mode):
        self.src, self.dst, self.args = src, dst, (B, H, W, C, P, mode)

    def fwd(self, ctx):
        tops.token_rearrange(self.src.t, self.dst.t, *self.args)

    def bwd(self, ctx):
        assert not self.src.written
        tops.token_rearrange(self.dst.grad(), self.src.grad(), *self.args, inverse=True)
        self.src.mark_written()


class Im2colOp:
    def __init__(self, plan, cols, in_chans):
        self.plan, self.cols, self.in_chans = plan, cols, in_chans

    def fwd(self, ctx):
        tops.patch_im2col(self.plan.inp_t, self.cols.t, self.in_chans)

    def bwd(self, ctx):
        pass


class HeadOp:
    def __init__(self, x, w, logits):
        self.x, self.w, self.logits = x, w, logits
        self.w2 = w.data.view(w.data.shape[0], -1)
        self.gw2 = w.grad.view(w.grad.shape[0], -1)

    def fwd(self, ctx):
        tops.head_fwd(self.x.t, self.w2, self.logits.t)

    def bwd(self, ctx):
        assert not self.x.written
        tops.head_bwd(self.x.t, self.w2, self.logits.grad(), self.x.grad(), self.gw2)
        self.x.mark_written()


class LnHeadOp:
    """LayerNorm(x) -> bias-free 1x1 output head in one pass (tops.ln_head_fwd): the normalised tensor -- the largest of the
    network, read by nobody else -- is never written; the backward forms dx and the three parameter gradients from one
    more read of x."""

    def __init__(self, x, g, b, w, logits):
        self.x, self.g, self.b, self.w, self.logits = x, g, b, w, logits
        self.w2 = w.data.view(w.data.shape[0], -1)
        self.gw2 = w.grad.view(w.grad.shape[0], -1)
        self.mean = torch.empty(x.rows, dtype=torch.float32, device="cuda")
        self.rstd = torch.empty(x.rows, dtype=torch.float32, device="cuda")
        self.expand = None
        self.skip_fwd = False

    @staticmethod
    def eligible(C, num_classes):
        return LN_HEAD and C % 4 == 0 and C <= 128 and 2 <= num_classes <= 4

    def fwd(self, ctx):
        if self.skip_fwd:         # the expand GEMM's epilogue did this op's forward (mis_gemm_expand_ln_head)
            self.skip_fwd = False
            return
        ok = tops.ln_head_fwd(self.x.t, self.g.data, self.b.data, self.w2, self.mean, self.rstd, self.logits.t)
        assert ok

    def bwd(self, ctx):
        e = self.expand           # the ExpandLinearOp whose pixel-shuffled output is this op's input (set by the plan), or None
        if UNSHUFFLE and e is not None and not self.x.written and not e.y.written:
            # dx goes straight into the gradient of the expand Linear's (un-shuffled) output: no rearrange pass
            B, H, W, c, P = e.geo
            tops.ln_head_bwd(self.x.t, self.g.data, self.b.data, self.w2, self.mean, self.rstd, self.logits.grad(),
                             e.y.grad(), self.g.grad, self.b.grad, self.gw2, unshuffle=(H, W, P))
            e.y.mark_written()
            e.unshuffled_by_consumer = True
            self.x.mark_written()
            return
        tops.ln_head_bwd(self.x.t, self.g.data, self.b.data, self.w2, self.mean, self.rstd, self.logits.grad(),
                         self.x.grad(), self.g.grad, self.b.grad, self.gw2, accumulate_dx=self.x.written)
        self.x.mark_written()


# ---------------------------------------------------------------- UNETR ops (token part of networks/unetr.py)
class Patch3dOp:
    """MONAI PatchEmbeddingBlock('perceptron') re-arrangement: volume [B,1,H,W,D] -> rows of 16^3 voxels per patch."""

    def __init__(self, plan, cols, P=16):
        self.plan, self.cols, self.P = plan, cols, P

    def fwd(self, ctx):
        tops.patch3d_im2col(self.plan.inp.t, self.cols.t, self.P)

    def bwd(self, ctx):
        pass                                   # the input volume needs no gradient


class PosAddOp:
    """out = x + position_embeddings[row % L];  d(pos) = sum over the batch, d(x) = d(out) (aliased, no copy)."""

    def __init__(self, x, pos, out, L):
        self.x, self.pos, self.out, self.L = x, pos, out, L

    def fwd(self, ctx):
        tops.add_rowcycle(self.x.t, self.pos.data.view(self.L, -1), self.out.t, self.L)

    def bwd(self, ctx):
        assert not self.x.written
        tops.sum_rowcycle(self.out.grad(), self.pos.grad.view(self.L, -1), self.L)
        self.x.g = self.out.grad()
        self.x.mark_written()


class FullAttnOp:
    """softmax(q k^T / sqrt(64)) v over all N tokens of a sample, per head (MONAI SABlock core)."""

    def __init__(self, qkv, out, B, N, nH):
        self.qkv, self.out, self.geo = qkv, out, (B, N, nH)
        self.scale = 64 ** -0.5
        self.stats = torch.empty(B * nH * N * 2, dtype=torch.float32, device="cuda")

    def fwd(self, ctx):
        tops.full_attention_fwd(self.qkv.t, self.out.t, self.stats, *self.geo, self.scale)

    def bwd(self, ctx):
        assert not self.qkv.written
        tops.full_attention_bwd(self.qkv.t, self.out.grad(), self.qkv.grad(), self.stats, *self.geo, self.scale)
        self.qkv.mark_written()


class TokToVolOp:
    """proj_feat (reference code/networks/unetr.py:183-186): tokens [B*L, C] -> volume [B, C, h, w, d] (a transpose per
    sample); backward transposes the volume gradient back into the (so far untouched) token gradient."""

    def __init__(self, tok, vol, B, L):
        self.tok, self.vol, self.B, self.L = tok, vol, B, L

    def fwd(self, ctx):
        C = self.tok.C
        for b in range(self.B):
            tops.transpose(self.tok.t[b * self.L:(b + 1) * self.L], self.vol.t[b].reshape(C, self.L))

    def bwd(self, ctx):
        assert not self.tok.written, "the conv branch must be the first consumer to run backward"
        C = self.tok.C
        g, gv = self.tok.grad(), self.vol.grad()
        for b in range(self.B):
            tops.transpose(gv[b].reshape(C, self.L), g[b * self.L:(b + 1) * self.L])
        self.tok.mark_written()


# ---------------------------------------------------------------- SwinUNETR ops (token part of networks/swinunetr.py)
class ConstRef:
    """Stand-in for a parameter reference with constant data (the affine-free F.layer_norm of SwinUNETR's proj_out: gamma
    = 1, beta = 0); its gradient buffer is private, so dist.param_progress does not see it."""

    def __init__(self, data):
        self.data, self.grad = data, torch.zeros_like(data)


class VolToTokOp:
    """volume [B, C, d, h, w] -> tokens [B*L, C] (a transpose per sample): the inverse of TokToVolOp."""

    def __init__(self, vol, tok, B, L):
        self.vol, self.tok, self.B, self.L = vol, tok, B, L

    def fwd(self, ctx):
        C = self.tok.C
        for b in range(self.B):
            tops.transpose(self.vol.t[b].reshape(C, self.L), self.tok.t[b * self.L:(b + 1) * self.L])

    def bwd(self, ctx):
        assert not self.vol.written
        C = self.tok.C
        g, gv = self.tok.grad(), self.vol.grad()
        for b in range(self.B):
            tops.transpose(g[b * self.L:(b + 1) * self.L], gv[b].reshape(C, self.L))
        self.vol.mark_written()


class Win3dOp:
    """MONAI SwinTransformerBlock.forward_part1's data movement: ``to_windows`` = F.pad + torch.roll + window_partition
    (tokens [B*L, C] -> windows [B*nW*n, C]); otherwise window_reverse + roll back + un-pad.  Each direction is the
    other's gradient (padding slots receive zero)."""

    def __init__(self, src, dst, B, dims, win, shift, to_windows):
        self.src, self.dst, self.args, self.to_windows = src, dst, (B, dims, src.C, win, shift), to_windows

    def fwd(self, ctx):
        tops.win3d_gather(self.src.t, self.dst.t, *self.args, inverse=not self.to_windows)

    def bwd(self, ctx):
        assert not self.src.written
        tops.win3d_gather(self.dst.grad(), self.src.grad(), *self.args, inverse=self.to_windows)
        self.src.mark_written()


class Win3dAttnOp:
    """MONAI WindowAttention core on 7^3 (or clipped) windows, heads of 16 channels (csrc/swin3d.hip)."""

    def __init__(self, qkv, out, table, region, BW, nW, n, nH):
        self.qkv, self.out, self.table, self.region = qkv, out, table, region
        self.geo = (BW, nW, n, nH)
        self.stats = torch.empty(BW * nH * n * 2, dtype=torch.float32, device="cuda")

    def fwd(self, ctx):
        tops.win3d_attn_fwd(self.qkv.t, self.out.t, self.stats, self.table.data, self.region, *self.geo)

    def bwd(self, ctx):
        assert not self.qkv.written
        tops.win3d_attn_bwd(self.qkv.t, self.out.t, self.out.grad(), self.qkv.grad(), self.stats, self.table.data,
                            self.region, self.table.grad, *self.geo)
        self.qkv.mark_written()


class Merge3dOp:
    """MONAI PatchMerging ("merging", v0.9 slot order) gather: tokens [B*L, C] -> [B*L/8, 8C]."""

    def __init__(self, src, dst, B, dims):
        self.src, self.dst, self.args = src, dst, (B, dims, src.C)

    def fwd(self, ctx):
        tops.merge3d(self.src.t, self.dst.t, *self.args)

    def bwd(self, ctx):
        assert not self.src.written      # the blocks' output has no other reader (proj_out reads the MERGED tokens)
        tops.merge3d(self.dst.grad(), self.src.grad(), *self.args, inverse=True)
        self.src.mark_written()


class SwinPlan:
    """Op list + buffers of one SwinUnet for one input geometry; same interface as ``plan.Plan``."""

    def __init__(self, net, in_shape):
        self.net = net
        self.in_shape = tuple(in_shape)      # (N, 1, 1, H, W)
        self.ops, self.acts = [], []
        self.inp_t = None
        self._site = itertools.count(0)
        self.out = None
        self.generation = 0      # forwards run on this plan (see mis_hip.plan.Plan / _NetFn.backward)
        self._progress = None
        self._wgrad_stream = None
        self._tbatch = None
        self._dp_gen, self._dp_const = -1, None
        self._paired = False

    def new(self, rows, C):
        a = TAct(rows, C)
        self.acts.append(a)
        return a

    def cols(self, parent, c0, C):
        a = parent.cols(c0, C)
        self.acts.append(a)
        return a

    def add(self, op):
        if isinstance(op, ResidualOp):
            op.plan = self
        self.ops.append(op)
        return op

    def droppath_scales(self, ctx):
        """[sites][B] DropPath scales of THIS forward pass (one launch per forward, cached by generation): the values
        ResidualOp derives on the fly from (state, salt, sample), as a table for the fused GEMM epilogues."""
        if self._dp_gen != self.generation:
            res = [op for op in self.ops if isinstance(op, ResidualOp)]
            n, B = len(res), self.in_shape[0]
            key = ctx.rng_stream
            # one (p, salt, table) set per Philox sub-stream: UA-MT cycles the teacher through several streams per step
            # on the same plan, and rebuilding the constants means two pageable host-to-device copies in the hot loop
            if self._dp_const is None:
                self._dp_const = {}
            if key not in self._dp_const:
                p = torch.tensor([op.drop_p for op in res], dtype=torch.float32).cuda()
                salt = torch.tensor([(((ctx.rng_stream & 0xFFFF) << 16) | op.site) for op in res],
                                    dtype=torch.int64).to(torch.int32).cuda()       # bit pattern of the unsigned salt
                assert [op.site for op in res] == list(range(n))
                self._dp_const[key] = (p, salt, torch.empty((n, B), dtype=torch.float32, device="cuda"))
            p, salt, table = self._dp_const[key]
            tops.droppath_table(table, p, salt, n, B, ctx.state)
            self._dp_gen = self.generation
            self._dp_table = table
        return self._dp_table

    def next_site(self):
        return next(self._site)

    def transpose_weights(self, ops_list=None):
        """W^T of every Linear that needs a data gradient, one launch (the weights changed with the last SGD update)."""
        if self._tbatch is None:
            jobs = []
            for op in (self.ops if ops_list is None else ops_list):
                if isinstance(op, LinearOp) and op.need_dx and op.w2.is_contiguous():
                    op.wT = torch.empty((op.w2.shape[1], op.w2.shape[0]), dtype=torch.float32, device="cuda")
                    op.wT_batched = True
                    jobs.append((op.w2, op.wT))
            self._tbatch = tops.TransposeBatch(jobs) if jobs else False
        if self._tbatch:
            self._tbatch.run()

    def forward(self, x5, ctx):
        assert tuple(x5.shape) == self.in_shape, (tuple(x5.shape), self.in_shape)
        if not self._paired:
            pair_ln_residual(self.ops)
            self._paired = True
        self.generation += 1
        self.inp_t = x5[:, :, 0]             # [N, 1 | 3, H, W]
        ctx.b3_fwd = split_linear_weights(self, self.ops, False)
        for op in self.ops:
            op.fwd(ctx)
        return self.out.t

    def backward(self, dlogits5, ctx, on_progress=None):
        for a in self.acts:
            a.reset()
        self.out.reset()
        if dlogits5 is not None:
            self.out.g = dlogits5
        if not self._paired:
            pair_ln_residual(self.ops)
            self._paired = True
        self.transpose_weights()
        ctx.b3_bwd = split_linear_weights(self, self.ops, True)
        from . import plan as _plan
        main, side = torch.cuda.current_stream(), None
        if _plan.WGRAD_STREAM:
            if self._wgrad_stream is None:
                self._wgrad_stream = _lib.side_stream("wgrad")
            side = self._wgrad_stream
        ctx.wgrad_stream = side
        ctx.deferred = [] if side is not None else None
        begin_finals(ctx, self)
        if on_progress is None:
            for op in reversed(self.ops):
                op.bwd(ctx)
            flush_deferred(ctx)
            ctx.deferred = ctx.finals = None
            if side is not None:
                _lib.wait_stream(main, side)
            return
        if self._progress is None:
            from .dist import param_progress
            self._progress = param_progress(self.ops, self.net.flat_grad)
        report = _plan.progress_reporter(on_progress, main, side, before=lambda: flush_deferred(ctx))
        for i in range(len(self.ops) - 1, -1, -1):      # see mis_hip.plan.Plan.backward
            self.ops[i].bwd(ctx)
            if i == 0 or self._progress[i] != self._progress[i - 1] or i == len(self.ops) - 1:
                report(self._progress[i])
        flush_deferred(ctx)
        ctx.deferred = ctx.finals = None
        if side is not None:
            _lib.wait_stream(main, side)

    def drop_sites(self):
        return [op.site for op in self.ops if isinstance(op, ResidualOp) and op.drop_p > 0]


def new_logits(N, C, H, W):
    return Act((N, C, 1, H, W))

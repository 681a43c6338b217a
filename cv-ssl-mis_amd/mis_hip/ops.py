"""Tensor-level wrappers over the C-ABI kernels (one Python function per entry point).

Activation tensors are 5-D ``[N, C, D, H, W]`` fp32 (2-D images use ``D == 1``) whose
channel/spatial dims are dense and whose batch stride is free (``>= C*D*H*W``), so
channel slices of a concatenated skip buffer are valid inputs and outputs.
Every function launches on torch's current stream and returns nothing (outputs are
caller-allocated), mirroring the C signatures in ``include/mis_hip.h``.
"""
import ctypes as _ctypes
import os as _os

import torch

from . import lib as _l

_scratch = {}


def _geom(t):
    """(N, C, D, H, W, S, batch_stride) of a 5-D activation view; validates density."""
    if t.dim() != 5 or t.dtype != torch.float32:
        raise RuntimeError(f"expected 5-D fp32 [N,C,D,H,W], got {tuple(t.shape)} {t.dtype}")
    _l.require_gpu(t)
    N, C, D, H, W = t.shape
    S = D * H * W
    st = t.stride()
    if C > 1 and st[1] != S or (D > 1 and st[2] != H * W) or (H > 1 and st[3] != W) or (W > 1 and st[4] != 1):
        raise RuntimeError(f"activation view must be dense in (C,D,H,W); strides {st} shape {tuple(t.shape)}")
    bs = st[0] if N > 1 else max(st[0], C * S)
    if bs < C * S:
        bs = C * S
    return N, C, D, H, W, S, bs


_scratch_retired = []


def scratch(nbytes, key="default"):
    """Grow-only device scratch buffer (bytes) for kernel workspaces, per device, key and stream."""
    # keyed by stream too: student and teacher forwards may run concurrently on two streams (step.py)
    k = (torch.cuda.current_device(), key, torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # a captured hipGraph may hold raw pointers into the buffer being outgrown (an eager validation forward
            # between replays can need more workspace than the training step did): never hand it back to the allocator
            _scratch_retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device="cuda")
        _scratch[k] = buf
    return buf


# ---------------------------------------------------------------- convolution
def conv_pack(weight, mode, out=None):
    """Repack ``weight [Cout,Cin,*k]`` for conv_fwd (mode 0) or the data gradient (mode 1)."""
    L = _l.load()
    Cout, Cin = weight.shape[0], weight.shape[1]
    taps = weight[0, 0].numel()
    n = L.mis_conv_packed_floats(Cout, Cin, taps, mode)
    if n < 0:
        _l.check(n, "mis_conv_packed_floats")
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=weight.device)
    assert out.numel() >= n and weight.is_contiguous()
    _l.check(L.mis_conv_pack_weights(_l.ptr(weight), _l.ptr(out), Cout, Cin, taps, mode, _l.stream_ptr()),
             "mis_conv_pack_weights")
    return out


class PackBatch:
    """All weight re-layouts of a network in one launch (``mis_conv_pack_batch``).  ``jobs``: list of
    (weight tensor [Cout,Cin,*k], packed buffer, mode); the device job table is built once."""

    def __init__(self, jobs):
        L = _l.load()
        nb = L.mis_conv_pack_job_bytes()
        host = (_ctypes.c_char * (nb * len(jobs)))()
        start = 0
        for i, (w, wp, mode) in enumerate(jobs):
            assert w.is_contiguous() and wp.is_contiguous()
            Cout, Cin = w.shape[0], w.shape[1]
            n = L.mis_conv_pack_job(_ctypes.byref(host, i * nb), _l.ptr(w), _l.ptr(wp), Cout, Cin, w[0, 0].numel(),
                                    mode, start)
            if n < 0:
                _l.check(n, "mis_conv_pack_job")
            assert wp.numel() >= n
            start += n
        self.n, self.total = len(jobs), start
        self.keep = jobs      # the table holds raw pointers into these tensors
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()

    def run(self):
        L = _l.load()
        _l.check(L.mis_conv_pack_batch(_l.ptr(self.table), self.n, self.total, _l.stream_ptr()), "mis_conv_pack_batch")


def conv_pack_raw(w, Cout, Cin, taps, mode, out=None):
    """Pack a weight buffer with explicit geometry (modes 2/3: 1x1 weight stored input-major [Cin][Cout])."""
    L = _l.load()
    n = L.mis_conv_packed_floats(Cout, Cin, taps, mode)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    assert out.numel() >= n and w.is_contiguous() and w.numel() == Cout * Cin * taps
    _l.check(L.mis_conv_pack_weights(_l.ptr(w), _l.ptr(out), Cout, Cin, taps, mode, _l.stream_ptr()),
             "mis_conv_pack_weights")
    return out


def space_to_depth2(src, dst, fine_shape, to_depth, bias=None, accumulate=False):
    """fine [N,C,D,H,W] <-> coarse [N,8C,D/2,H/2,W/2]; ``fine_shape`` = (N,C,D,H,W) of the fine tensor."""
    L = _l.load()
    N, C, D, H, W = fine_shape
    sbs, dbs = _geom(src)[6], _geom(dst)[6]
    _l.check(L.mis_space_to_depth2(_l.ptr(src), sbs, _l.ptr(dst), dbs, _l.ptr(bias), N, C, D, H, W, int(to_depth),
                                   int(accumulate), _l.stream_ptr()), "mis_space_to_depth2")


def space_to_depth2d(src, dst, fine_shape, to_depth, bias=None, accumulate=False):
    """2-D: fine [N,C,1,H,W] <-> coarse [N,4C,1,H/2,W/2]; ``fine_shape`` = (N,C,1,H,W) of the fine tensor."""
    L = _l.load()
    N, C, D, H, W = fine_shape
    assert D == 1
    sbs, dbs = _geom(src)[6], _geom(dst)[6]
    _l.check(L.mis_space_to_depth2d(_l.ptr(src), sbs, _l.ptr(dst), dbs, _l.ptr(bias), N, C, H, W, int(to_depth),
                                    int(accumulate), _l.stream_ptr()), "mis_space_to_depth2d")


def conv_k2s2_eligible(cin, cout, coarse, up):
    """(cin, cout) and the COARSE (Do, Ho, Wo) the in-place kernel-2 / stride-2 kernels serve (conv_k2s2.hip)."""
    return K2S2 and bool(_l.load().mis_conv_k2s2_eligible(int(cin), int(cout), *[int(v) for v in coarse], int(up)))


def conv_k2s2_wgrad_eligible(cf, cc, coarse):
    return K2S2 and bool(_l.load().mis_conv_k2s2_wgrad_eligible(int(cf), int(cc), *[int(v) for v in coarse]))


# When set to a set(), the ops below add a short tag of the path they dispatched (tests assert that the full-batch
# instantiations -- in-place kernel-2 / stride-2 kernels, Winograd weight-gradient variants -- are the ones that ran).
DISPATCH = None


def _tag(name):
    if DISPATCH is not None:
        DISPATCH.add(name)


def conv_k2s2_up_output_ok(y):
    """mis_conv_k2s2_up stores float2 pairs: the fine output (possibly a channel-slice view) must start on 8 bytes with an
    even batch stride (the kernel returns MIS_ERR_UNSUPPORTED otherwise)."""
    return y.data_ptr() % 8 == 0 and _geom(y)[6] % 2 == 0


def conv_k2s2_wgrad_operands_ok(coarse, fine):
    """mis_conv_k2s2_wgrad reads float4 rows: both operands on 16 bytes, batch strides multiples of 4 floats."""
    return (coarse.data_ptr() % 16 == 0 and fine.data_ptr() % 16 == 0 and _geom(coarse)[6] % 4 == 0
            and _geom(fine)[6] % 4 == 0)


def conv_k2s2_wgrad(coarse, fine, dw, accumulate=False):
    """dw[cc][cf*8 + tap] (+)= sum coarse[n][cc][v] * fine[n][cf][2v + tap] (mis_conv_k2s2_wgrad): the parameter gradient of
    Conv3d(k2s2) (coarse = dy, fine = x) and of ConvTranspose3d(k2s2) (coarse = x, fine = dy), in the parameter's layout."""
    L = _l.load()
    N, CC, Do, Ho, Wo, _, cbs = _geom(coarse)
    _, CF, _, _, _, _, fbs = _geom(fine)
    assert dw.is_contiguous() and dw.numel() == CC * CF * 8
    ws = scratch(L.mis_conv_k2s2_wgrad_workspace_bytes(CF, CC), "wgrad")
    _tag(f"k2s2_wgrad:{CF}x{CC}@{Do}")
    _l.check(L.mis_conv_k2s2_wgrad(_l.ptr(coarse), cbs, _l.ptr(fine), fbs, _l.ptr(dw), N, CF, CC, Do, Ho, Wo,
                                   int(accumulate), _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_conv_k2s2_wgrad")


def conv_k2s2_down(x, w, bias, y, accumulate=False):
    """y (coarse) (+)= bias + sum w[co][ci*8 + tap] x[ci][2v + tap]: Conv3d(k2s2) forward / ConvTranspose3d(k2s2) dX."""
    L = _l.load()
    N, Cin, _, _, _, _, xbs = _geom(x)
    _, Cout, Do, Ho, Wo, _, ybs = _geom(y)
    assert w.is_contiguous() and w.numel() == Cin * Cout * 8
    _tag(f"k2s2_down:{Cin}x{Cout}@{Do}")
    _l.check(L.mis_conv_k2s2_down(_l.ptr(x), xbs, _l.ptr(w), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, Do, Ho, Wo,
                                  int(accumulate), _l.stream_ptr()), "mis_conv_k2s2_down")


def conv_k2s2_up(x, w, bias, y, accumulate=False):
    """y (fine) (+)= bias + sum w[ci][co*8 + tap] x[ci][v]: ConvTranspose3d(k2s2) forward / Conv3d(k2s2) dX."""
    L = _l.load()
    N, Cin, Do, Ho, Wo, _, xbs = _geom(x)
    _, Cout, _, _, _, _, ybs = _geom(y)
    assert w.is_contiguous() and w.numel() == Cin * Cout * 8
    _tag(f"k2s2_up:{Cin}x{Cout}@{Do}")
    _l.check(L.mis_conv_k2s2_up(_l.ptr(x), xbs, _l.ptr(w), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, Do, Ho, Wo,
                                int(accumulate), _l.stream_ptr()), "mis_conv_k2s2_up")


def add(a, b, out):
    """out = a (+ b if b is not None) on [N,C,D,H,W] views."""
    L = _l.load()
    N, C, D, H, W, S, abs_ = _geom(a)
    bbs = _geom(b)[6] if b is not None else 0
    obs = _geom(out)[6]
    _l.check(L.mis_add(_l.ptr(a), abs_, _l.ptr(b), bbs, _l.ptr(out), obs, N, C, S, _l.stream_ptr()), "mis_add")


def _ksize(k):
    if len(k) == 2:
        return 1, k[0], k[1]
    return tuple(k)


K2S2 = _os.environ.get("MIS_K2S2", "1") != "0"      # kernel-2 / stride-2 (de)convolutions in place (conv_k2s2.hip)
WINO = int(_os.environ.get("MIS_WINO", "3"))      # bit 0: Winograd form of the 3x3x3 convolutions, bit 1: of the 3x3 ones
# diagnostics (numerics studies, tests/test_parity_gpu.py): the 3-D forward / data gradient resp. weight gradient alone on the
# direct kernels, and the smallest volume edge W the 3-D Winograd kernels are used for
WINO_FWD = _os.environ.get("MIS_WINO_FWD", "1") != "0"
WINO_WGRAD = _os.environ.get("MIS_WINO_WGRAD", "1") != "0"
WINO_MIN_W = int(_os.environ.get("MIS_WINO_MIN_W", "0"))


WINO2D = 10       # ids >= WINO2D: variant id - WINO2D of the 2-D kernels (conv_wino2d.hip); below: 3-D (conv_wino.hip)


# 1x1x1 convolutions of small volumes with many channels (the space-to-depth form of V-Net's deep kernel-2 / stride-2 layers):
# batched GEMM instead of the spatially tiled direct kernel (conv1x1_gemm.hip).  MIS_CONV1X1_GEMM=0 switches it off
CONV1X1_GEMM = _os.environ.get("MIS_CONV1X1_GEMM", "1") != "0"
CONV1X1_GEMM_MAX_S = int(_os.environ.get("MIS_CONV1X1_GEMM_MAX_S", "4096"))


def conv1x1_gemm_eligible(x, y, cin, cout):
    """x / y: 5-D NCDHW views with dense (C, D, H, W); few voxels, many channels, sizes the GEMM's float4 rows need."""
    if not CONV1X1_GEMM:
        return False
    try:
        _, _, _, _, _, S, xbs = _geom(x)
        _, _, _, _, _, _, ybs = _geom(y)
    except RuntimeError:
        return False
    return (S <= CONV1X1_GEMM_MAX_S and S % 4 == 0 and cout % 4 == 0 and cin >= 64 and xbs % 4 == 0 and ybs % 4 == 0 and
            x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0)


def conv1x1_gemm(x, wt, bias, y, accumulate=False):
    """y[n][co][s] (+)= bias[co] + sum_ci wt[ci][co] x[n][ci][s] (mis_conv1x1_gemm); ``wt`` = the weights [Cin][Cout]."""
    L = _l.load()
    N, Cin, D, H, W, S, xbs = _geom(x)
    _, Cout, _, _, _, _, ybs = _geom(y)
    assert wt.dim() == 2 and wt.shape == (Cin, Cout) and wt.stride(1) == 1, (wt.shape, Cin, Cout)
    nb = L.mis_conv1x1_gemm_workspace_bytes(N, Cin, Cout, S)
    if nb < 0:
        _l.check(nb, "mis_conv1x1_gemm_workspace_bytes")
    ws = scratch(nb, "conv1x1") if nb > 0 else None
    _tag(f"conv1x1_gemm:{Cin}x{Cout}@{W}")
    _l.check(L.mis_conv1x1_gemm(_l.ptr(x), xbs, _l.ptr(wt), wt.stride(0), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, S,
                                int(accumulate), _l.ptr(ws), nb, _l.stream_ptr()), "mis_conv1x1_gemm")


def conv1x1_wgrad_eligible(a, b):
    """a / b: the two 5-D NCDHW operands of the weight gradient (same N, D, H, W)."""
    if not CONV1X1_GEMM:
        return False
    try:
        _, _, _, _, _, S, abs_ = _geom(a)
        _, _, _, _, _, _, bbs = _geom(b)
    except RuntimeError:
        return False
    return (S <= CONV1X1_GEMM_MAX_S and S % 4 == 0 and abs_ % 4 == 0 and bbs % 4 == 0 and a.data_ptr() % 16 == 0 and
            b.data_ptr() % 16 == 0 and min(a.shape[1], b.shape[1]) >= 64)


def conv1x1_wgrad(a, b, dw, accumulate=False):
    """dw[M][Nc] (+)= sum_{image, voxel} a[.][m][s] * b[.][n][s] (mis_conv1x1_wgrad); dw: a 2-D view [a channels][b channels]."""
    L = _l.load()
    N, M, D, H, W, S, abs_ = _geom(a)
    _, Nc, _, _, _, _, bbs = _geom(b)
    assert dw.dim() == 2 and dw.shape == (M, Nc) and dw.stride(1) == 1, (dw.shape, M, Nc)
    nb = L.mis_conv1x1_wgrad_workspace_bytes(N, M, Nc, S)
    if nb < 0:
        _l.check(nb, "mis_conv1x1_wgrad_workspace_bytes")
    ws = scratch(nb, "conv1x1_wgrad")
    _tag(f"conv1x1_wgrad:{M}x{Nc}@{W}")
    _l.check(L.mis_conv1x1_wgrad(_l.ptr(a), abs_, _l.ptr(b), bbs, _l.ptr(dw), dw.stride(0), N, M, Nc, S, int(accumulate),
                                 _l.ptr(ws), nb, _l.stream_ptr()), "mis_conv1x1_wgrad")


def conv_wino_select(N, Cin, Cout, D, H, W, ksize):
    """Winograd variant serving this convolution (mis_conv3d_wino_select / mis_conv2d_wino_select), or -1: use the
    direct kernel."""
    if not WINO:
        return -1
    k = _ksize(ksize)
    if k == (3, 3, 3) and (WINO & 1):
        if not WINO_FWD or W < WINO_MIN_W:
            return -1
        return int(_l.load().mis_conv3d_wino_select(N, Cin, Cout, D, H, W))
    if k == (1, 3, 3) and D == 1 and (WINO & 2):
        v = int(_l.load().mis_conv2d_wino_select(N, Cin, Cout, H, W))
        return v + WINO2D if v >= 0 else -1
    return -1


def conv_wino_pack_mode(wino, dgrad):
    """Pack mode of the transformed filter for this variant (4 / 5: 3x3x3, 6 / 7: 3x3; forward / data gradient)."""
    return (6 if wino >= WINO2D else 4) + (1 if dgrad else 0)


def conv_stat_tiles(N, Cin, Cout, D, H, W, ksize, wino=-1):
    """Partial-statistics tiles per image of the fused conv+stats form for this geometry (0: not eligible)."""
    kd, kh, kw = _ksize(ksize)
    if wino >= WINO2D:
        t = _l.load().mis_conv2d_wino_stat_tiles(H, W, wino - WINO2D)
    elif wino >= 0:
        t = _l.load().mis_conv3d_wino_stat_tiles(D, H, W, wino)
    else:
        t = _l.load().mis_conv_fwd_stat_tiles(N, Cin, Cout, D, H, W, kd, kh, kw)
    if t < 0:
        _l.check(t, "mis_conv_fwd_stat_tiles")
    return int(t)


def norm_stats_finalize(part, N, C, S, tiles, per_sample, eps, mean, rstd, running_mean=None, running_var=None,
                        num_batches=None, momentum=0.1):
    L = _l.load()
    _l.check(L.mis_norm_stats_finalize(_l.ptr(part), N, C, S, tiles, int(per_sample), eps, _l.ptr(mean), _l.ptr(rstd),
                                       _l.ptr(running_mean), _l.ptr(running_var), _l.ptr(num_batches), momentum,
                                       _l.stream_ptr()), "mis_norm_stats_finalize")


def conv_fwd(x, wp, bias, y, Cin, Cout, ksize, stat=None, wino=-1):
    """y = conv(x) with packed weights ``wp``; stride 1, 'same' zero padding, k in {1,3}.
    ``stat = (buffer, stride_channel, stride_image)``: also emit the per-tile (sum, sumsq) of y (mis_conv_fwd_stats).
    ``wino >= 0``: ``wp`` is the Winograd-transformed filter (pack mode 4 / 5) and that variant of
    mis_conv3d_wino_fwd runs."""
    L = _l.load()
    N, Cx, D, H, W, S, xbs = _geom(x)
    Ny, Cy, Dy, Hy, Wy, _, ybs = _geom(y)
    assert Cx == Cin and Cy == Cout and (N, D, H, W) == (Ny, Dy, Hy, Wy)
    kd, kh, kw = _ksize(ksize)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if wino >= WINO2D:
        st = stat if stat is not None else (None, 0, 0)
        _l.check(L.mis_conv2d_wino_fwd(_l.ptr(x), xbs, _l.ptr(wp), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, H, W,
                                       _l.ptr(st[0]), st[1], st[2], wino - WINO2D, _l.stream_ptr()), "mis_conv2d_wino_fwd")
    elif wino >= 0:
        st = stat if stat is not None else (None, 0, 0)
        # few boxes (the 6^3 level, half batches at 12^3): the contraction is cut into slices, partials in scratch
        nb = L.mis_conv3d_wino_fwd_workspace_bytes(N, Cin, Cout, D, H, W, wino)
        if nb < 0:
            _l.check(nb, "mis_conv3d_wino_fwd_workspace_bytes")
        ws = scratch(nb, "wino_fwd") if nb > 0 else None
        if ws is not None:
            _tag(f"wino_fwd_split:v{wino}@{W}")
        _l.check(L.mis_conv3d_wino_fwd_ws(_l.ptr(x), xbs, _l.ptr(wp), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, D, H, W,
                                          _l.ptr(st[0]), st[1], st[2], wino, _l.ptr(ws), nb, _l.stream_ptr()),
                 "mis_conv3d_wino_fwd_ws")
    elif stat is not None:
        _l.check(L.mis_conv_fwd_stats(_l.ptr(x), xbs, _l.ptr(wp), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, D, H, W,
                                      kd, kh, kw, _l.ptr(stat[0]), stat[1], stat[2], _l.stream_ptr()),
                 "mis_conv_fwd_stats")
    else:
        _l.check(L.mis_conv_fwd(_l.ptr(x), xbs, _l.ptr(wp), _l.ptr(bias), _l.ptr(y), ybs, N, Cin, Cout, D, H, W,
                                kd, kh, kw, _l.stream_ptr()), "mis_conv_fwd")
    if prof is not None:
        e1.record()
        buf = _ctypes.create_string_buffer(128)
        if wino >= WINO2D:
            L.mis_conv2d_wino_kernel_name(wino - WINO2D, buf, 128)
        elif wino >= 0:
            L.mis_conv3d_wino_kernel_name(wino, buf, 128)
        else:
            L.mis_conv_fwd_kernel_name(N, Cin, Cout, D, H, W, kd, kh, kw, buf, 128)
        prof.append((buf.value.decode(), 2.0 * N * Cout * Cin * kd * kh * kw * S, e0, e1,
                     4.0 * (N * (Cin + Cout) * S + Cout * Cin * kd * kh * kw)))


# When set to a list, conv_fwd brackets every launch with HIP events on the launch stream and appends
# (kernel name, algorithmic FLOPs, start event, end event, algorithmic bytes = operands read once + result written once):
# bench.py's live roofline measurement.
PROFILE = None


def conv_dgrad_norm(dy, wpd, da, Cin, Cout, xn, mean, slope, part, wino):
    """The Winograd data gradient ``da = dgrad(dy)`` (``wpd``: pack mode 5) with the first stage of the backward of the
    InstanceNorm + (Leaky)ReLU that produced the conv's input from ``xn`` in its epilogue (mis_conv3d_wino_dgrad_norm):
    ``part [N*Cout*tiles, 2]`` receives (sum dz, sum dz * xn) per (n, channel, tile).  ``Cin`` / ``Cout``: channels of
    dy / da.  Returns the number of tiles per image."""
    L = _l.load()
    N, C, D, H, W, S, dbs = _geom(dy)
    _, _, _, _, _, _, abs_ = _geom(da)
    _, _, _, _, _, _, xbs = _geom(xn)
    assert C == Cin and da.shape[1] == Cout and xn.shape[1] == Cout
    tiles = int(L.mis_conv3d_wino_stat_tiles(D, H, W, wino))
    assert part.numel() >= N * Cout * tiles * 2
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _l.check(L.mis_conv3d_wino_dgrad_norm(_l.ptr(dy), dbs, _l.ptr(wpd), _l.ptr(da), abs_, N, Cin, Cout, D, H, W,
                                          _l.ptr(xn), xbs, _l.ptr(mean), float(slope), _l.ptr(part), tiles, Cout * tiles,
                                          wino, _l.stream_ptr()), "mis_conv3d_wino_dgrad_norm")
    if prof is not None:
        e1.record()
        buf = _ctypes.create_string_buffer(128)
        L.mis_conv3d_wino_kernel_name(wino, buf, 128)
        prof.append((buf.value.decode().replace(", false>", ", true>"), 2.0 * N * Cout * Cin * 27 * S, e0, e1,
                     4.0 * (N * (Cin + 2 * Cout) * S + Cout * Cin * 27)))     # + the norm input it reads
    return tiles


def norm_act_bwd_tiles(x, da, dx, mean, rstd, slope, part, tiles, sums):
    """Second stage + apply pass of the InstanceNorm + (Leaky)ReLU backward from conv_dgrad_norm's partials; ``dx`` may
    be None (only ``sums`` is wanted: the first layer's weight gradient forms dx itself)."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, dabs = _geom(da)
    dxbs = _geom(dx)[6] if dx is not None else 0
    _l.check(L.mis_norm_act_bwd_tiles(_l.ptr(x), xbs, _l.ptr(da), dabs, _l.ptr(dx), dxbs, N, C, S, _l.ptr(mean),
                                      _l.ptr(rstd), float(slope), _l.ptr(part), int(tiles), _l.ptr(sums),
                                      _l.stream_ptr()), "mis_norm_act_bwd_tiles")


def conv_wgrad(x, dy, dw, ksize, accumulate=False):
    """dw[Cout,Cin,*k] (+)= sum_n,p dy * shifted x."""
    L = _l.load()
    N, Cin, D, H, W, S, xbs = _geom(x)
    _, Cout, _, _, _, _, dbs = _geom(dy)
    kd, kh, kw = _ksize(ksize)
    assert dw.is_contiguous() and dw.numel() == Cout * Cin * kd * kh * kw
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    name = _conv_wgrad_launch(L, x, dy, dw, N, Cin, Cout, D, H, W, xbs, dbs, kd, kh, kw, accumulate)
    if prof is not None:       # bench.py's live roofline: the weight gradients count towards the executed step flops
        e1.record()
        prof.append((name, 2.0 * N * Cout * Cin * kd * kh * kw * S, e0, e1,
                     4.0 * (N * (Cin + Cout) * S + Cout * Cin * kd * kh * kw)))


def _conv_wgrad_launch(L, x, dy, dw, N, Cin, Cout, D, H, W, xbs, dbs, kd, kh, kw, accumulate):
    # 3x3x3 on large volumes: the Winograd F(2^3, 3^3) form (conv_wino_wgrad.hip), 3.375x fewer matrix-pipe flops
    wino = int(L.mis_conv3d_wino_wgrad_select(N, Cin, Cout, D, H, W)) if ((WINO & 1) and WINO_WGRAD and W >= WINO_MIN_W and
                                                                          (kd, kh, kw) == (3, 3, 3)) else -1
    if wino >= 0 and xbs % 4 == 0 and dbs % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0:
        nb = L.mis_conv3d_wino_wgrad_workspace_bytes(N, Cin, Cout, D, H, W, wino)
        if nb < 0:
            _l.check(nb, "mis_conv3d_wino_wgrad_workspace_bytes")
        ws = scratch(nb, "wgrad")
        _tag(f"wino_wgrad:v{wino}@{W}")
        _l.check(L.mis_conv3d_wino_wgrad(_l.ptr(x), xbs, _l.ptr(dy), dbs, _l.ptr(dw), _l.ptr(ws), ws.numel(), N, Cin,
                                         Cout, D, H, W, int(accumulate), wino, _l.stream_ptr()), "mis_conv3d_wino_wgrad")
        buf = _ctypes.create_string_buffer(96)
        _l.check(L.mis_conv3d_wino_wgrad_kernel_name(wino, buf, 96), "mis_conv3d_wino_wgrad_kernel_name")
        return buf.value.decode()
    wino2 = int(L.mis_conv2d_wino_wgrad_select(N, Cin, Cout, H, W)) if ((WINO & 2) and (kd, kh, kw) == (1, 3, 3) and D == 1) else -1
    if wino2 >= 0 and xbs % 4 == 0 and dbs % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0:
        nb = L.mis_conv2d_wino_wgrad_workspace_bytes(N, Cin, Cout, H, W, wino2)
        if nb < 0:
            _l.check(nb, "mis_conv2d_wino_wgrad_workspace_bytes")
        ws = scratch(nb, "wgrad")
        _l.check(L.mis_conv2d_wino_wgrad(_l.ptr(x), xbs, _l.ptr(dy), dbs, _l.ptr(dw), _l.ptr(ws), ws.numel(), N, Cin, Cout,
                                         H, W, int(accumulate), wino2, _l.stream_ptr()), "mis_conv2d_wino_wgrad")
        buf = _ctypes.create_string_buffer(96)
        _l.check(L.mis_conv2d_wino_wgrad_kernel_name(wino2, buf, 96), "mis_conv2d_wino_wgrad_kernel_name")
        return buf.value.decode()
    nb = L.mis_conv_wgrad_workspace_bytes(N, Cin, Cout, D, H, W, kd, kh, kw)
    if nb < 0:
        _l.check(nb, "mis_conv_wgrad_workspace_bytes")
    ws = scratch(nb, "wgrad")
    _tag(f"direct_wgrad:k{kd}{kh}{kw}@{W}")
    _l.check(L.mis_conv_wgrad(_l.ptr(x), xbs, _l.ptr(dy), dbs, _l.ptr(dw), _l.ptr(ws), ws.numel(), N, Cin, Cout,
                              D, H, W, kd, kh, kw, int(accumulate), _l.stream_ptr()), "mis_conv_wgrad")
    buf = _ctypes.create_string_buffer(96)
    _l.check(L.mis_conv_wgrad_kernel_name(N, Cin, Cout, D, H, W, kd, kh, kw, buf, 96), "mis_conv_wgrad_kernel_name")
    return buf.value.decode()


# ------------------------------------------------------- norm + act + dropout
def norm_stats(x, per_sample, eps, mean, rstd, running_mean=None, running_var=None, num_batches=None,
               momentum=0.1):
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    nb = L.mis_norm_workspace_bytes(N, C, S, int(per_sample))
    ws = scratch(nb, "norm")
    _l.check(L.mis_norm_stats(_l.ptr(x), xbs, N, C, S, int(per_sample), eps, _l.ptr(mean), _l.ptr(rstd),
                              _l.ptr(running_mean), _l.ptr(running_var), _l.ptr(num_batches), momentum,
                              _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_norm_stats")


def channel_sum(x, out, accumulate=False):
    """out[c] (+)= sum_{n,s} x[n,c,s]  (conv bias gradient)."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    nb = L.mis_norm_workspace_bytes(N, C, S, 0)
    ws = scratch(nb, "norm")
    _l.check(L.mis_channel_sum(_l.ptr(x), xbs, N, C, S, _l.ptr(out), int(accumulate), _l.ptr(ws), ws.numel(),
                               _l.stream_ptr()), "mis_channel_sum")


def norm_stats_from_running(running_mean, running_var, eps, mean, rstd):
    L = _l.load()
    _l.check(L.mis_norm_stats_from_running(_l.ptr(running_mean), _l.ptr(running_var), eps, _l.ptr(mean),
                                           _l.ptr(rstd), running_mean.numel(), _l.stream_ptr()),
             "mis_norm_stats_from_running")


def norm_act_fwd(x, y, per_sample, mean, rstd, gamma, beta, slope, drop_p=0.0, drop_salt=0, state=None,
                 drop_mask=None, cg=1):
    """``cg``: channels per statistics group of a per-sample norm (1 InstanceNorm, C/16 GroupNorm(16))."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, ybs = _geom(y)
    _l.check(L.mis_norm_act_fwd_g(_l.ptr(x), xbs, _l.ptr(y), ybs, N, C, S, int(per_sample), int(cg), _l.ptr(mean),
                                  _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta), slope, drop_p, drop_salt,
                                  _l.ptr(state), _l.ptr(drop_mask), _l.stream_ptr()), "mis_norm_act_fwd_g")


def norm_act_bwd(x, da, dx, per_sample, mean, rstd, gamma, beta, slope, drop_p=0.0, drop_salt=0, state=None,
                 drop_mask=None, dgamma=None, dbeta=None, accumulate_affine=False, cg=1, no_norm=False):
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, dabs = _geom(da)
    _, _, _, _, _, _, dxbs = _geom(dx)
    nb = L.mis_norm_workspace_bytes(N, C, S, int(per_sample))
    ws = scratch(nb, "norm")
    _l.check(L.mis_norm_act_bwd_g(_l.ptr(x), xbs, _l.ptr(da), dabs, _l.ptr(dx), dxbs, N, C, S, int(per_sample),
                                  int(cg), int(no_norm), _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta),
                                  slope, drop_p, drop_salt, _l.ptr(state), _l.ptr(drop_mask), _l.ptr(dgamma),
                                  _l.ptr(dbeta), int(accumulate_affine), _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_norm_act_bwd_g")


def norm_res_act_fwd(x, res, y, per_sample, mean, rstd, gamma, beta, slope, post=False):
    """y = act(norm(x) + res) in one pass (mis_norm_res_act_fwd); ``post``: y = act(norm(x)) + res."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    rbs, ybs = _geom(res)[6], _geom(y)[6]
    _l.check(L.mis_norm_res_act_fwd(_l.ptr(x), xbs, _l.ptr(res), rbs, _l.ptr(y), ybs, N, C, S, int(per_sample),
                                    _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta), slope, int(post),
                                    _l.stream_ptr()), "mis_norm_res_act_fwd")


def norm_res_act_bwd(x, res, dy, dx, dres, accumulate_dres, per_sample, mean, rstd, gamma, beta, slope, dgamma=None,
                     dbeta=None, accumulate_affine=False, post=False):
    """Backward of norm_res_act_fwd: dres (+)= dz, dx = norm backward of dz, dz = dy * act'(norm(x) + res)."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    rbs, dybs, dxbs, drbs = _geom(res)[6], _geom(dy)[6], _geom(dx)[6], _geom(dres)[6]
    ws = scratch(L.mis_norm_workspace_bytes(N, C, S, int(per_sample)), "norm")
    _l.check(L.mis_norm_res_act_bwd(_l.ptr(x), xbs, _l.ptr(res), rbs, _l.ptr(dy), dybs, _l.ptr(dx), dxbs, _l.ptr(dres),
                                    drbs, int(accumulate_dres), N, C, S, int(per_sample), _l.ptr(mean), _l.ptr(rstd),
                                    _l.ptr(gamma), _l.ptr(beta), slope, _l.ptr(dgamma), _l.ptr(dbeta),
                                    int(accumulate_affine), int(post), _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_norm_res_act_bwd")


def norm_act_fwd_pool(x, y, pooled, idx, per_sample, mean, rstd, gamma, beta, slope, drop_p=0.0, drop_salt=0, state=None,
                      drop_mask=None, cg=1):
    """norm_act_fwd and the 2x max-pool of its output (pooled, idx as maxpool2_fwd writes them) in one pass."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, ybs = _geom(y)
    _, _, _, _, _, _, pbs = _geom(pooled)
    _l.check(L.mis_norm_act_fwd_pool(_l.ptr(x), xbs, _l.ptr(y), ybs, _l.ptr(pooled), pbs, _l.ptr(idx), N, C, D, H, W,
                                     int(per_sample), int(cg), _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta),
                                     slope, drop_p, drop_salt, _l.ptr(state), _l.ptr(drop_mask), _l.stream_ptr()),
             "mis_norm_act_fwd_pool")


def norm_act_bwd_pool(x, da, dpool, idx, dx, per_sample, mean, rstd, gamma, beta, slope, drop_p=0.0, drop_salt=0,
                      state=None, drop_mask=None, dgamma=None, dbeta=None, accumulate_affine=False, cg=1):
    """norm_act_bwd of an activation that also feeds a 2x max-pool: incoming gradient = da (None: no other consumer) +
    the pool's backward of dpool through the argmax codes idx, added on the load path (no maxpool2_bwd pass)."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    dabs = _geom(da)[6] if da is not None else 0
    _, _, _, _, _, _, dpbs = _geom(dpool)
    _, _, _, _, _, _, dxbs = _geom(dx)
    ws = scratch(L.mis_norm_workspace_bytes(N, C, S, int(per_sample)), "norm")
    _l.check(L.mis_norm_act_bwd_pool(_l.ptr(x), xbs, _l.ptr(da), dabs, _l.ptr(dpool), dpbs, _l.ptr(idx), _l.ptr(dx), dxbs,
                                     N, C, D, H, W, int(per_sample), int(cg), _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma),
                                     _l.ptr(beta), slope, drop_p, drop_salt, _l.ptr(state), _l.ptr(drop_mask),
                                     _l.ptr(dgamma), _l.ptr(dbeta), int(accumulate_affine), _l.ptr(ws), ws.numel(),
                                     _l.stream_ptr()), "mis_norm_act_bwd_pool")


def norm_act_bwd_sums(x, da, per_sample, mean, rstd, gamma, beta, slope, sums, dgamma=None, dbeta=None,
                      accumulate_affine=False):
    """The reduction half of norm_act_bwd: sums[group] = (mean dz, mean dz*xhat) (+ dgamma / dbeta); no dropout."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, dabs = _geom(da)
    ws = scratch(L.mis_norm_workspace_bytes(N, C, S, int(per_sample)), "norm")
    _l.check(L.mis_norm_act_bwd_sums(_l.ptr(x), xbs, _l.ptr(da), dabs, N, C, S, int(per_sample), _l.ptr(mean),
                                     _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta), slope, _l.ptr(sums), _l.ptr(dgamma),
                                     _l.ptr(dbeta), int(accumulate_affine), _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_norm_act_bwd_sums")


def conv_wgrad_cin1_norm_eligible(N, Cout, D, H, W):
    return bool(_l.load().mis_conv_wgrad_cin1_norm_eligible(N, Cout, D, H, W))


def conv_wgrad_cin1_norm(x, da, y, per_sample, mean, rstd, gamma, beta, sums, slope, dw, accumulate=False):
    """dw of the first layer Conv3d(1 -> 16, 3) from the gradient at the activation after its norm + (Leaky)ReLU."""
    L = _l.load()
    N, Cin, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, dabs = _geom(da)
    _, Cout, _, _, _, _, ybs = _geom(y)
    nb = L.mis_conv_wgrad_workspace_bytes(N, Cin, Cout, D, H, W, 1 if D == 1 else 3, 3, 3)
    if nb < 0:
        _l.check(nb, "mis_conv_wgrad_workspace_bytes")
    ws = scratch(nb, "wgrad")
    _l.check(L.mis_conv_wgrad_cin1_norm(_l.ptr(x), xbs, _l.ptr(da), dabs, _l.ptr(y), ybs, N, D, H, W, int(per_sample),
                                        _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta), _l.ptr(sums), slope,
                                        _l.ptr(dw), _l.ptr(ws), ws.numel(), int(accumulate), _l.stream_ptr()),
             "mis_conv_wgrad_cin1_norm")


def norm_head_eligible(C, K, per_sample, gamma, beta, cg=1, no_norm=False):
    """norm + act + dropout + 1x1x1 classifier as one pass (mis_norm_head_*): C == 16, K in 2..4, one statistics group per
    channel, InstanceNorm only without affine."""
    return bool(_l.load().mis_norm_head_eligible(C, K)) and cg == 1 and not no_norm and not (
        per_sample and (gamma is not None or beta is not None))


def norm_head_fwd(x, logits, per_sample, mean, rstd, gamma, beta, slope, w, b, drop_p=0.0, drop_salt=0, state=None,
                  drop_mask=None):
    """logits = W . drop(act(norm(x))) + b without storing the activation (w: [K, C] view of the 1x1x1 conv weight)."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, K, _, _, _, _, lbs = _geom(logits)
    _l.check(L.mis_norm_head_fwd(_l.ptr(x), xbs, N, C, S, int(per_sample), _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma),
                                 _l.ptr(beta), slope, drop_p, drop_salt, _l.ptr(state), _l.ptr(drop_mask), _l.ptr(w),
                                 _l.ptr(b), K, _l.ptr(logits), lbs, _l.stream_ptr()), "mis_norm_head_fwd")


def norm_head_bwd(x, dlogits, dx, per_sample, mean, rstd, gamma, beta, slope, w, dw, db, drop_p=0.0, drop_salt=0,
                  state=None, drop_mask=None, dgamma=None, dbeta=None, accumulate_affine=False, accumulate_w=False):
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, K, _, _, _, _, dlbs = _geom(dlogits)
    _, _, _, _, _, _, dxbs = _geom(dx)
    nb = L.mis_norm_head_workspace_bytes(N, C, S, int(per_sample), K)
    ws = scratch(nb, "norm")
    _l.check(L.mis_norm_head_bwd(_l.ptr(x), xbs, _l.ptr(dlogits), dlbs, _l.ptr(dx), dxbs, N, C, S, int(per_sample),
                                 _l.ptr(mean), _l.ptr(rstd), _l.ptr(gamma), _l.ptr(beta), slope, drop_p, drop_salt,
                                 _l.ptr(state), _l.ptr(drop_mask), _l.ptr(w), K, _l.ptr(dgamma), _l.ptr(dbeta),
                                 int(accumulate_affine), _l.ptr(dw), _l.ptr(db), int(accumulate_w), _l.ptr(ws),
                                 ws.numel(), _l.stream_ptr()), "mis_norm_head_bwd")


def group_norm_stats(x, cg, eps, mean, rstd):
    """(mean, rstd) per (n, group of ``cg`` consecutive channels): the groups are contiguous in NCDHW, so this is the
    InstanceNorm statistics kernel on the [N, C/cg, cg*S] view of the same memory."""
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    G = C // cg
    nb = L.mis_norm_workspace_bytes(N, G, cg * S, 1)
    ws = scratch(nb, "norm")
    _l.check(L.mis_norm_stats(_l.ptr(x), xbs, N, G, cg * S, 1, eps, _l.ptr(mean), _l.ptr(rstd), None, None, None,
                              0.0, _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_norm_stats")


# ------------------------------------------------------------ pool / upsample
def maxpool2_fwd(x, y, idx):
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, ybs = _geom(y)
    _l.check(L.mis_maxpool2_fwd(_l.ptr(x), xbs, _l.ptr(y), ybs, _l.ptr(idx), N, C, D, H, W, _l.stream_ptr()),
             "mis_maxpool2_fwd")


def maxpool2_bwd(dy, idx, dx, accumulate=False):
    L = _l.load()
    N, C, D, H, W, S, dxbs = _geom(dx)
    _, _, _, _, _, _, dybs = _geom(dy)
    _l.check(L.mis_maxpool2_bwd(_l.ptr(dy), dybs, _l.ptr(idx), _l.ptr(dx), dxbs, N, C, D, H, W, int(accumulate),
                                _l.stream_ptr()), "mis_maxpool2_bwd")


def upsample2_fwd(x, y, align_corners):
    L = _l.load()
    N, C, D, H, W, S, xbs = _geom(x)
    _, _, _, _, _, _, ybs = _geom(y)
    _l.check(L.mis_upsample2_fwd(_l.ptr(x), xbs, _l.ptr(y), ybs, N, C, D, H, W, int(align_corners),
                                 _l.stream_ptr()), "mis_upsample2_fwd")


def upsample2_bwd(dy, dx, align_corners, accumulate=False):
    L = _l.load()
    N, C, D, H, W, S, dxbs = _geom(dx)
    _, _, _, _, _, _, dybs = _geom(dy)
    _l.check(L.mis_upsample2_bwd(_l.ptr(dy), dybs, _l.ptr(dx), dxbs, N, C, D, H, W, int(align_corners),
                                 int(accumulate), _l.stream_ptr()), "mis_upsample2_bwd")


# ------------------------------------------------------------------ loss tail
def loss_tail(student, teacher, label, labeled_bs, out, dlogits=None, cons_weight=0.0, state=None,
              loss_scale=1.0):
    """Fused softmax/CE/Dice/consistency.  ``out``: >= 5+C floats:
    [loss, loss_ce, loss_dice, consistency_loss, consistency_weight, class-wise dice...]."""
    L = _l.load()
    B, C, D, H, W, S, sbs = _geom(student)
    tbs = 0
    if teacher is not None:
        Bt, Ct, _, _, _, St, tbs = _geom(teacher)
        assert Bt == B - labeled_bs and Ct == C and St == S
    if label is not None:
        _l.require_gpu(label)
        assert label.is_contiguous() and label.dtype in (torch.uint8, torch.int64)
        assert label.numel() >= labeled_bs * S
    lb = 1 if (label is None or label.dtype == torch.uint8) else 8
    dbs = 0
    if dlogits is not None:
        dbs = _geom(dlogits)[6]
    nb = L.mis_loss_tail_workspace_bytes(B, C, S)
    ws = scratch(nb, "tail")
    _l.check(L.mis_loss_tail(_l.ptr(student), sbs, _l.ptr(teacher), tbs, _l.ptr(label), lb, B, labeled_bs, C, S,
                             cons_weight, _l.ptr(state), loss_scale, _l.ptr(out), _l.ptr(dlogits), dbs,
                             _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_loss_tail")


def softmax_mean_accumulate(logits, acc, repeats, scale, first):
    """acc[u] (+)= scale * sum_r softmax(logits[r*U + u]) over the channel dim (UA-MT MC-dropout mean)."""
    L = _l.load()
    Bl, C, D, H, W, S, lbs = _geom(logits)
    U, Ca, _, _, _, Sa, abs_ = _geom(acc)
    assert Bl == repeats * U and Ca == C and Sa == S
    _l.check(L.mis_softmax_mean_accumulate(_l.ptr(logits), lbs, _l.ptr(acc), abs_, U, repeats, C, S, scale,
                                           int(first), _l.stream_ptr()), "mis_softmax_mean_accumulate")


def uamt_tail(student, teacher, mean_probs, label, labeled_bs, out, max_iterations, dlogits=None, cons_weight=0.0,
              state=None, iter_num=0, loss_scale=1.0):
    """UA-MT loss tail.  ``out`` (>= 7+C floats): [loss, loss_ce, loss_dice, consistency_loss, consistency_weight,
    class-wise dice..., #unmasked voxels, threshold]."""
    L = _l.load()
    B, C, D, H, W, S, sbs = _geom(student)
    Bt, Ct, _, _, _, St, tbs = _geom(teacher)
    Bm, Cm, _, _, _, Sm, mbs = _geom(mean_probs)
    assert Bt == Bm == B - labeled_bs and Ct == Cm == C and St == Sm == S
    _l.require_gpu(label)
    assert label.is_contiguous() and label.dtype in (torch.uint8, torch.int64) and label.numel() >= labeled_bs * S
    lb = 1 if label.dtype == torch.uint8 else 8
    dbs = _geom(dlogits)[6] if dlogits is not None else 0
    nb = L.mis_uamt_tail_workspace_bytes(B, C, S)
    ws = scratch(nb, "tail")
    _l.check(L.mis_uamt_tail(_l.ptr(student), sbs, _l.ptr(teacher), tbs, _l.ptr(mean_probs), mbs, _l.ptr(label), lb,
                             B, labeled_bs, C, S, cons_weight, _l.ptr(state), int(iter_num), float(max_iterations),
                             loss_scale, _l.ptr(out), _l.ptr(dlogits), dbs, _l.ptr(ws), ws.numel(),
                             _l.stream_ptr()), "mis_uamt_tail")


def cross_teaching_tail(own, other, label, labeled_bs, out, dlogits=None, cons_weight=0.0, state=None,
                        pseudo_ce=False, teacher=None, mt_weight=0.0):
    """0.5*(CE+Dice) on the labeled half + w * Dice (``pseudo_ce``: cross-entropy, CPS) against the other network's
    arg-max pseudo labels.  ``out`` (>= 5 floats): [loss_m, loss_ce, loss_dice, pseudo_supervision, consistency_weight].
    ``teacher`` (EMA-teacher logits of the unlabeled half): + mt_weight * softmax-MSE consistency, out[5:7] =
    [consistency_loss, mt_weight] (reference code/train_cnn_meet_vit_2D.py:300-337)."""
    L = _l.load()
    B, C, D, H, W, S, sbs = _geom(own)
    Bo, Co, _, _, _, So, obs = _geom(other)
    assert (Bo, Co, So) == (B, C, S)
    if teacher is not None:
        Bt, Ct, _, _, _, St, tbs = _geom(teacher)
        assert (Bt, Ct, St) == (B - labeled_bs, C, S)
        _l.require_gpu(label)
        assert label.is_contiguous() and label.dtype in (torch.uint8, torch.int64) and label.numel() >= labeled_bs * S
        ws = scratch(L.mis_cross_teaching_tail_workspace_bytes(B, C, S), "tail")
        _l.check(L.mis_cross_pseudo_mt_tail(_l.ptr(own), sbs, _l.ptr(other), obs, _l.ptr(teacher), tbs, _l.ptr(label),
                                            1 if label.dtype == torch.uint8 else 8, B, labeled_bs, C, S, cons_weight,
                                            mt_weight, _l.ptr(state), int(bool(pseudo_ce)), _l.ptr(out),
                                            _l.ptr(dlogits), _geom(dlogits)[6] if dlogits is not None else 0,
                                            _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_cross_pseudo_mt_tail")
        return
    _l.require_gpu(label)
    assert label.is_contiguous() and label.dtype in (torch.uint8, torch.int64) and label.numel() >= labeled_bs * S
    lb = 1 if label.dtype == torch.uint8 else 8
    dbs = _geom(dlogits)[6] if dlogits is not None else 0
    ws = scratch(L.mis_cross_teaching_tail_workspace_bytes(B, C, S), "tail")
    _l.check(L.mis_cross_pseudo_tail(_l.ptr(own), sbs, _l.ptr(other), obs, _l.ptr(label), lb, B, labeled_bs, C,
                                     S, cons_weight, _l.ptr(state), int(bool(pseudo_ce)), _l.ptr(out),
                                     _l.ptr(dlogits), dbs, _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_cross_pseudo_tail")


# ------------------------------------------------------------ optimizer / rng
def sgd_ema_step(param, grad, momentum_buf, ema_param, lr=0.0, momentum=0.9, weight_decay=1e-4, ema_alpha=0.99,
                 grad_scale=1.0, state=None):
    L = _l.load()
    _l.require_gpu(param, grad, momentum_buf, ema_param)
    n = param.numel()
    assert grad.numel() == n and momentum_buf.numel() == n and (ema_param is None or ema_param.numel() == n)
    _l.check(L.mis_sgd_ema_step(_l.ptr(param), _l.ptr(grad), _l.ptr(momentum_buf), _l.ptr(ema_param), n, lr,
                                momentum, weight_decay, ema_alpha, grad_scale, _l.ptr(state), _l.stream_ptr()),
             "mis_sgd_ema_step")


def teacher_noise(x, y, state, sigma=0.1, clamp_abs=0.2, salt=0x7EAC4E5):
    L = _l.load()
    _l.require_gpu(x, y, state)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    _l.check(L.mis_teacher_noise(_l.ptr(x), _l.ptr(y), x.numel(), sigma, clamp_abs, salt, _l.ptr(state),
                                 _l.stream_ptr()), "mis_teacher_noise")


def new_step_state():
    return torch.zeros(_l.STEP_STATE_BYTES, dtype=torch.uint8, device="cuda")


def step_init(state, seed, iter_num, base_lr, max_iterations, ema_decay, consistency, rampup, ramp_div=150,
              cons_start_iter=0, lr_post_increment=False):
    L = _l.load()
    _l.check(L.mis_step_init(_l.ptr(state), seed, iter_num, base_lr, float(max_iterations), ema_decay, consistency,
                             rampup, ramp_div, cons_start_iter, int(lr_post_increment), _l.stream_ptr()),
             "mis_step_init")


def step_advance(state, base_lr, max_iterations, ema_decay, consistency, rampup, ramp_div=150, cons_start_iter=0,
                 lr_post_increment=False):
    L = _l.load()
    _l.check(L.mis_step_advance(_l.ptr(state), base_lr, float(max_iterations), ema_decay, consistency, rampup,
                                ramp_div, cons_start_iter, int(lr_post_increment), _l.stream_ptr()),
             "mis_step_advance")


def read_step_state(state):
    """Host copy of the device step state (one small D2H; not used inside the hot loop)."""
    import struct
    raw = bytes(state.cpu().numpy().tobytes())
    seed, offset, it, lr, alpha, w, gate = struct.unpack("<QQqffff", raw[:40])
    return dict(seed=seed, offset=offset, iter_num=it, lr=lr, ema_alpha=alpha, cons_weight=w, cons_gate=gate)


def argmax_channels(x, out):
    L = _l.load()
    B, C, D, H, W, S, xbs = _geom(x)
    assert out.dtype == torch.uint8 and out.numel() == B * S
    _l.check(L.mis_argmax_channels(_l.ptr(x), xbs, _l.ptr(out), B, C, S, _l.stream_ptr()), "mis_argmax_channels")

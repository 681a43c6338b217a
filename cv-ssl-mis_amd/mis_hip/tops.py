"""Tensor-level wrappers of the token-major C-ABI kernels (SwinUnet): GEMM, LayerNorm, GELU,
residual+DropPath, token re-arrangements, patch im2col, output head, window attention.

Token tensors are 2-D ``[rows, C]`` fp32 views with a dense last dim and a free row stride (``ld``),
so column slices of a wider buffer (the decoder's concat) are valid operands.
"""
import ctypes
import os

import torch

from . import lib as _l
from . import ops as _ops
from .ops import scratch

# B operands (weights) of the NT GEMMs pre-split into their bf16 pieces once per step (mis_gemm_nt_split); 0: split per tile
PRESPLIT = os.environ.get("MIS_GEMM_PRESPLIT", "1") != "0"


def _mat(t):
    """(rows, cols, ld) of a 2-D row-major view."""
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"expected 2-D fp32 row-major view, got {tuple(t.shape)} strides {t.stride()}")
    _l.require_gpu(t)
    return t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _nt_name(L, M, N, K, epilogue):
    """NT instantiation mis_gemm / mis_gemm_ex pick, as rocprofv3 names it."""
    buf = ctypes.create_string_buffer(96)
    _l.check(L.mis_gemm_nt_kernel_name(M, N, K, epilogue, buf, 96), "mis_gemm_nt_kernel_name")
    return buf.value.decode()


def _tn_name(L, A, lda, B, ldb, C, ldc, M, N, K):
    """TN kernel mis_gemm(trans) / mis_gemm_dw pick for these operands, as rocprofv3 names it."""
    buf = ctypes.create_string_buffer(96)
    _l.check(L.mis_gemm_tn_kernel_name(_l.ptr(A), lda, _l.ptr(B), ldb, _l.ptr(C), ldc, M, N, K, buf, 96),
             "mis_gemm_tn_kernel_name")
    return buf.value.decode()


class SplitB:
    """The B operand of an NT GEMM cut into its three bf16 piece planes (``mis_gemm_split_b``): ``src`` [N, K] dense fp32 (an
    nn.Linear weight or its transpose); ``t`` the planes.  ``refresh()`` after the weights changed; a ``SplitBatch`` refreshes
    every SplitB of a network in one launch."""

    def __init__(self, src, rows=None):
        """``rows``: the token rows M of the GEMM these planes serve -- decides the element order inside a K = 32 block
        (``natural``: the register-A kernel's, mis_gemm_nt_split_natural; None: the staged kernels' order)."""
        N, K, ldb = _mat(src)
        L = _l.load()
        nb = L.mis_gemm_split_bytes(N, K)
        if nb < 0:
            _l.check(nb, "mis_gemm_split_bytes")
        self.src, self.N, self.K, self.ldb = src, N, K, ldb
        self.natural = bool(rows) and bool(L.mis_gemm_nt_split_natural(int(rows), N, K))
        self.rows = rows
        self.t = torch.empty(nb, dtype=torch.uint8, device="cuda")

    def refresh(self):
        _l.check(_l.load().mis_gemm_split_b_layout(_l.ptr(self.src), self.ldb, self.N, self.K, _l.ptr(self.t), int(self.natural),
                                                   _l.stream_ptr()), "mis_gemm_split_b_layout")
        return self


class SplitBatch:
    """``refresh()`` of many SplitB in one launch (``mis_gemm_split_batch``); the device job table is built once."""

    def __init__(self, splits):
        L = _l.load()
        nb = L.mis_gemm_split_job_bytes()
        host = (ctypes.c_char * (nb * len(splits)))()
        first = 0
        for i, sb in enumerate(splits):
            n = L.mis_gemm_split_job_layout(ctypes.byref(host, i * nb), _l.ptr(sb.src), sb.ldb, sb.N, sb.K, _l.ptr(sb.t), first,
                                            int(sb.natural))
            if n < 0:
                _l.check(n, "mis_gemm_split_job_layout")
            first += n
        self.n, self.units, self.keep = len(splits), first, splits
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()

    def run(self):
        _l.check(_l.load().mis_gemm_split_batch(_l.ptr(self.table), self.n, self.units, _l.stream_ptr()), "mis_gemm_split_batch")


def split_active():
    """Pre-split B operands are used when the NT GEMMs run as bf16x3 products (``MIS_GEMM_PRESPLIT=0``: never)."""
    return PRESPLIT and bool(set_split_precision(-1) & 1)


def _nt_split(A, b3, C, bias=None, accumulate=False, epilogue=0, E1=None, C2=None, rowscale=None, rows_per_scale=1, ex=None):
    """mis_gemm_nt_split; False: outside what it covers (the caller runs the fp32-B entry point)."""
    L = _l.load()
    M, K, lda = _mat(A)
    N = b3.N
    assert K == b3.K, (A.shape, b3.N, b3.K)
    if ex is None and C is None:       # EP_GELU_FWD without the pre-activation (a forward nobody differentiates)
        assert epilogue == EP_GELU_FWD and C2 is not None
        ldc, eH, eW, eP, ec = N, 0, 0, 0, 0
    elif ex is None:
        Mc, Nc, ldc = _mat(C)
        assert (Mc, Nc) == (M, N), (C.shape, M, N)
        eH = eW = eP = ec = 0
    else:
        eH, eW, eP, ec = ex
        ldc = N
    lde1 = _mat(E1)[2] if E1 is not None else 0
    ldc2 = _mat(C2)[2] if C2 is not None else 0
    nb = L.mis_gemm_nt_split_workspace_bytes(M, N, K)
    ws = scratch(nb, "gemm") if nb > 0 else None
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if b3.natural and b3.rows != M:
        raise RuntimeError(f"natural-order planes were cut for {b3.rows} token rows, the GEMM has {M}")
    st = L.mis_gemm_nt_split_layout(_l.ptr(A), lda, _l.ptr(b3.t), _l.ptr(C), ldc, _l.ptr(bias), M, N, K, int(accumulate),
                                    int(epilogue), _l.ptr(E1), lde1, _l.ptr(C2), ldc2, _l.ptr(rowscale), int(rows_per_scale), eH, eW,
                                    eP, ec, _l.ptr(ws), ws.numel() if ws is not None else 0, int(b3.natural), _l.stream_ptr())
    if st == -2:
        if b3.natural:
            raise RuntimeError("natural-order planes, but the register-A kernel refuses this call (the staged kernels cannot read them)")
        return False
    _l.check(st, "mis_gemm_nt_split_layout")
    if prof is not None:
        e1.record()
        buf = ctypes.create_string_buffer(96)
        _l.check(L.mis_gemm_nt_split_layout_kernel_name(M, N, K, int(epilogue), int(b3.natural), buf, 96), "mis_gemm_nt_split_layout_kernel_name")
        prof.append((buf.value.decode(), 2.0 * M * N * K, e0, e1, 4.0 * (M * K + M * N) + 6.0 * N * K))
    return True


def gemm(A, B, C, bias=None, trans=False, accumulate=False, b3=None):
    """trans=False: C[M,N] (+)= A[M,K] @ B[N,K]^T (+ bias);  trans=True: C[M,N] (+)= A[K,M]^T @ B[K,N].
    ``b3``: B pre-split (SplitB, current): used when the NT GEMMs run as bf16x3 products."""
    if b3 is not None and not trans and split_active() and _nt_split(A, b3, C, bias=bias, accumulate=accumulate):
        return
    L = _l.load()
    if not trans:
        M, K, lda = _mat(A)
        N, K2, ldb = _mat(B)
    else:
        K, M, lda = _mat(A)
        K2, N, ldb = _mat(B)
    Mc, Nc, ldc = _mat(C)
    assert K == K2 and (Mc, Nc) == (M, N), (A.shape, B.shape, C.shape, trans)
    nb = L.mis_gemm_workspace_bytes(M, N, K, int(trans))
    ws = scratch(nb, "gemm") if nb > 0 else None
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _l.check(L.mis_gemm(_l.ptr(A), lda, _l.ptr(B), ldb, _l.ptr(C), ldc, _l.ptr(bias), M, N, K, int(trans),
                        int(accumulate), _l.ptr(ws), ws.numel() if ws is not None else 0, _l.stream_ptr()),
             "mis_gemm")
    if prof is not None:   # bench.py's live roofline measurement (same list as ops.conv_fwd)
        e1.record()
        # kernel instantiation mis_gemm picks (gemm.hip: tn_tile / nt_tile_n), as rocprofv3 names it
        if trans:
            name = _tn_name(L, A, lda, B, ldb, C, ldc, M, N, K)
        else:
            name = _nt_name(L, M, N, K, 0)
        prof.append((name, 2.0 * M * N * K, e0, e1, 4.0 * (M * K + N * K + M * N)))


def set_split_precision(mask):
    """Arithmetic of the nn.Linear GEMMs (mis_gemm_set_split_precision): bit 0 forward + dX, bit 1 dW as bf16x3 products
    (exact 3-way bf16 split of the fp32 operands, fp32 accumulation); 0 = fp32 MFMA.  Returns the previous mask; < 0 queries."""
    return int(_l.load().mis_gemm_set_split_precision(int(mask)))


def gemm_dw(dy, x, dW, db, accumulate=False):
    """dW[M,N] (+)= dy[K,M]^T @ x[K,N] and db[M] (+)= dy.sum(0) in one pass over dy (mis_gemm_dw)."""
    L = _l.load()
    K, M, lddy = _mat(dy)
    K2, N, ldx = _mat(x)
    Mc, Nc, ldw = _mat(dW)
    assert K == K2 and (Mc, Nc) == (M, N) and db.numel() == M and db.is_contiguous(), (dy.shape, x.shape, dW.shape)
    nb = L.mis_gemm_dw_workspace_bytes(M, N, K)
    ws = scratch(nb, "gemm") if nb > 0 else None
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _l.check(L.mis_gemm_dw(_l.ptr(dy), lddy, _l.ptr(x), ldx, _l.ptr(dW), ldw, _l.ptr(db), M, N, K, int(accumulate),
                           _l.ptr(ws), ws.numel() if ws is not None else 0, _l.stream_ptr()), "mis_gemm_dw")
    if prof is not None:
        e1.record()
        prof.append((_tn_name(L, dy, lddy, x, ldx, dW, ldw, M, N, K), 2.0 * M * N * K, e0, e1, 4.0 * (M * K + N * K + M * N)))


def gemm_dw_parts(dy, x, dW, db, ws, accumulate=False):
    """``gemm_dw`` without its finishing launch (mis_gemm_dw_parts): returns the number of k-slices whose partials now sit in the
    caller's workspace ``ws`` (fp32: ``slices`` matrices [M, N], then ``slices`` rows [M] when ``db`` is given) -- 0: the
    contraction was not split and dW / db are complete.  ``db`` may be None (bias-free Linear)."""
    L = _l.load()
    K, M, lddy = _mat(dy)
    K2, N, ldx = _mat(x)
    Mc, Nc, ldw = _mat(dW)
    assert K == K2 and (Mc, Nc) == (M, N) and (db is None or (db.numel() == M and db.is_contiguous())), (dy.shape, x.shape, dW.shape)
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    slices = ctypes.c_int(0)
    _l.check(L.mis_gemm_dw_parts(_l.ptr(dy), lddy, _l.ptr(x), ldx, _l.ptr(dW), ldw, _l.ptr(db), M, N, K, int(accumulate),
                                 _l.ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, ctypes.byref(slices),
                                 _l.stream_ptr()), "mis_gemm_dw_parts")
    if prof is not None:
        e1.record()
        prof.append((_tn_name(L, dy, lddy, x, ldx, dW, ldw, M, N, K), 2.0 * M * N * K, e0, e1, 4.0 * (M * K + N * K + M * N)))
    return slices.value


def gemm_dw_workspace(M, N, K):
    """A workspace of the caller's own for gemm_dw_parts (None: this shape is never split)."""
    nb = _l.load().mis_gemm_dw_workspace_bytes(M, N, K)
    if nb < 0:
        _l.check(nb, "mis_gemm_dw_workspace_bytes")
    return torch.empty(nb // 4, dtype=torch.float32, device="cuda") if nb > 0 else None


class ColsumJob:
    """One finishing column sum ``out_a[c] (+)= sum_s part[s][c]`` (``pairs``: part is float2, .x -> out_a, .y -> out_b) for a
    ``ColsumBatch`` (mis_colsum_job).  ``part`` is a tensor whose storage holds the partial rows, ``offset`` floats into it."""

    def __init__(self, part, offset, stride, slabs, C, pairs, out_a, out_b=None, accumulate=False):
        """``pairs``: False / 0 float partial rows, True / 1 float2 (``.x -> out_a, .y -> out_b``), 2 the k-slices of a split GEMM
        (fp32 sums in the order of the GEMM's own reduction kernel: bit-identical to ``gemm_dw``)."""
        self.part, self.offset, self.stride, self.slabs, self.C, self.pairs = part, int(offset), int(stride), int(slabs), int(C), int(pairs)
        self.out_a, self.out_b, self.accumulate = out_a, out_b, bool(accumulate)
        for o in (out_a, out_b):
            assert o is None or (o.is_contiguous() and o.numel() >= C and o.dtype == torch.float32)
        self.bytes = self.slabs * self.C * (8 if self.pairs == 1 else 4)

    def part_ptr(self):
        return self.part.data_ptr() + 4 * self.offset


class ColsumBatch:
    """Many ColsumJob in one launch (``mis_colsum_batch``); the device job table is built once and holds raw pointers."""

    def __init__(self, jobs):
        L = _l.load()
        nb = L.mis_colsum_job_bytes()
        host = (ctypes.c_char * (nb * len(jobs)))()
        first = 0
        for i, j in enumerate(jobs):
            n = L.mis_colsum_job(ctypes.byref(host, i * nb), ctypes.c_void_p(j.part_ptr()), j.stride, j.slabs, j.C, int(j.pairs),
                                 _l.ptr(j.out_a), _l.ptr(j.out_b), int(j.accumulate), first)
            if n < 0:
                _l.check(n, "mis_colsum_job")
            first += n
        self.n, self.blocks, self.keep = len(jobs), first, list(jobs)
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()

    def run(self):
        _l.check(_l.load().mis_colsum_batch(_l.ptr(self.table), self.n, self.blocks, _l.stream_ptr()), "mis_colsum_batch")


def colreduce_slabs(M):
    return int(_l.load().mis_colreduce_slabs(M))


def window_attention_table_partials(B, H, W, nH, window=7):
    """(rows, cols) of the bias-table partials ``window_attention_bwd_parts`` leaves at the start of its workspace, or None when
    the vector-pipe kernels (dS-block partials) are selected."""
    rows, cols = ctypes.c_longlong(0), ctypes.c_int(0)
    st = _l.load().mis_window_attention_table_partials(B, H, W, nH, window, ctypes.byref(rows), ctypes.byref(cols))
    if st == -2:
        return None
    _l.check(st, "mis_window_attention_table_partials")
    return rows.value, cols.value


EP_GELU_FWD, EP_GELU_BWD, EP_RESIDUAL = 1, 2, 3


def gemm_ex(A, B, C, epilogue, bias=None, E1=None, C2=None, rowscale=None, rows_per_scale=1, b3=None):
    """C = epilogue(A[M,K] @ B[N,K]^T + bias) (mis_gemm_ex): EP_GELU_FWD also writes C2 = gelu(.), EP_GELU_BWD multiplies
    by gelu'(E1), EP_RESIDUAL gives E1 + rowscale[row // rows_per_scale] * (.).  Returns False when the shape is
    outside the fused form (the caller then runs the un-fused ops)."""
    if b3 is not None and split_active() and _nt_split(A, b3, C, bias=bias, epilogue=epilogue, E1=E1, C2=C2, rowscale=rowscale,
                                                     rows_per_scale=rows_per_scale):
        return True
    L = _l.load()
    M, K, lda = _mat(A)
    N, K2, ldb = _mat(B)
    if C is None:          # EP_GELU_FWD only: the pre-activation is not kept (a forward nobody differentiates)
        assert epilogue == EP_GELU_FWD
        Mc, Nc, ldc = M, N, N
    else:
        Mc, Nc, ldc = _mat(C)
    assert K == K2 and (Mc, Nc) == (M, N)
    lde1 = _mat(E1)[2] if E1 is not None else 0
    ldc2 = _mat(C2)[2] if C2 is not None else 0
    nb = L.mis_gemm_workspace_bytes(M, N, K, 0)
    ws = scratch(nb, "gemm") if nb > 0 else None
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    st = L.mis_gemm_ex(_l.ptr(A), lda, _l.ptr(B), ldb, _l.ptr(C), ldc, _l.ptr(bias), M, N, K, int(epilogue),
                       _l.ptr(E1), lde1, _l.ptr(C2), ldc2, _l.ptr(rowscale), int(rows_per_scale), _l.ptr(ws),
                       ws.numel() if ws is not None else 0, _l.stream_ptr())
    if st == -2:
        return False
    _l.check(st, "mis_gemm_ex")
    if prof is not None:
        e1.record()
        # (split-K shapes run the plain instantiation + the reduce kernel; the label keeps the requested epilogue)
        prof.append((_nt_name(L, M, N, K, epilogue), 2.0 * M * N * K, e0, e1, 4.0 * (M * K + N * K + M * N)))
    return True


def gemm_residual_ln(A, b3, X, E1, gamma, beta, Y, mean, rstd, bias=None, rowscale=None, rows_per_scale=1, eps=1e-5):
    """X = E1 + rowscale[row // rows_per_scale] * (A @ W^T + bias) and Y = LayerNorm(X) * gamma + beta (mean / rstd kept) in one
    launch (mis_gemm_nt_residual_ln: the register-A kernels, 96 channels).  False: outside what it covers."""
    if b3 is None or not b3.natural or not split_active():
        return False
    L = _l.load()
    M, K, lda = _mat(A)
    Mx, N, ldx = _mat(X)
    _, _, lde1 = _mat(E1)
    _, _, ldy = _mat(Y)
    if N != 96 or b3.N != 96 or b3.K != K or Mx != M or b3.rows != M:
        return False
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    st = L.mis_gemm_nt_residual_ln(_l.ptr(A), lda, _l.ptr(b3.t), _l.ptr(bias), M, N, K, _l.ptr(E1), lde1, _l.ptr(rowscale),
                                   int(rows_per_scale), _l.ptr(X), ldx, _l.ptr(gamma), _l.ptr(beta), eps, _l.ptr(Y), ldy,
                                   _l.ptr(mean), _l.ptr(rstd), _l.stream_ptr())
    if st == -2:
        return False
    _l.check(st, "mis_gemm_nt_residual_ln")
    if prof is not None:
        e1.record()
        buf = ctypes.create_string_buffer(96)
        _l.check(L.mis_gemm_nt_split_layout_kernel_name(M, N, K, 5, 1, buf, 96), "mis_gemm_nt_split_layout_kernel_name")
        prof.append((buf.value.decode(), 2.0 * M * N * K, e0, e1, 4.0 * (M * K + 3 * M * N) + 6.0 * N * K))
    return True


def droppath_table(table, p_dev, salt_dev, nsites, B, state):
    L = _l.load()
    _l.check(L.mis_droppath_table(_l.ptr(table), _l.ptr(p_dev), _l.ptr(salt_dev), nsites, B, _l.ptr(state),
                                  _l.stream_ptr()), "mis_droppath_table")


def gemm_expand(x, w, out, B, H, W, P, c, b3=None):
    """out[(b, h*P+p1, w*P+p2)][c] = (x @ w^T) pixel-shuffled; returns False when the fused form does not cover the
    shape (the caller then runs gemm + token_rearrange)."""
    if b3 is not None and not b3.natural and split_active():      # (natural-order planes: only the register-A kernels read them)
        assert out.is_contiguous() and out.numel() == B * H * W * P * P * c
        if _nt_split(x, b3, out, ex=(H, W, P, c)):
            return True
    L = _l.load()
    M, K, lda = _mat(x)
    N, K2, ldb = _mat(w)
    assert K == K2 and M == B * H * W and N == P * P * c and out.is_contiguous() and out.numel() == M * N
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    st = L.mis_gemm_expand(_l.ptr(x), lda, _l.ptr(w), ldb, _l.ptr(out), B, H, W, K, P, c, _l.stream_ptr())
    if st == -2:        # MIS_ERR_UNSUPPORTED: shape outside the fused form
        return False
    _l.check(st, "mis_gemm_expand")
    if prof is not None:
        e1.record()
        prof.append((_nt_name(L, M, N, K, 0), 2.0 * M * N * K, e0, e1, 4.0 * (M * K + N * K + M * N)))
    return True


def gemm_expand_ln_head(x, w, out, B, H, W, P, c, gamma, beta, head_w, mean, rstd, logits5, eps=1e-5, b3=None):
    """FinalPatchExpand_X4's Linear + pixel shuffle + LayerNorm + output head in one launch (mis_gemm_expand_ln_head):
    ``out`` [B H P W P, c] (None: the expanded tokens are not kept), mean / rstd per expanded token, logits5 [B, NC, 1, H P, W P].
    ``b3``: the expand weight's planes in the natural order (SplitB(rows=M)): the persistent register-A kernel
    (mis_gemm_expand_ln_head_split).  False: outside the fused form."""
    L = _l.load()
    M, K, lda = _mat(x)
    N, K2, ldb = _mat(w)
    NC = logits5.shape[1]
    assert K == K2 and M == B * H * W and N == P * P * c and (out is None or (out.is_contiguous() and out.numel() == M * N))
    assert mean.numel() == M * P * P and rstd.numel() == M * P * P and tuple(head_w.shape) == (NC, c) and head_w.is_contiguous()
    prof = _ops.PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if b3 is not None and b3.natural and b3.rows == M and split_active():
        st = L.mis_gemm_expand_ln_head_split(_l.ptr(x), lda, _l.ptr(b3.t), _l.ptr(out), B, H, W, K, P, c, _l.ptr(gamma), _l.ptr(beta),
                                             _l.ptr(head_w), NC, eps, _l.ptr(mean), _l.ptr(rstd), _l.ptr(logits5),
                                             logits5.stride(0), _l.stream_ptr())
        if st == 0:
            if prof is not None:
                e1.record()
                prof.append(("gemm_nt_rega_res_kernel<4>", 2.0 * M * N * K, e0, e1,
                             4.0 * (M * K + (M * N if out is not None else 0)) + 6.0 * N * K))
            return True
        if st != -2:
            _l.check(st, "mis_gemm_expand_ln_head_split")
    st = L.mis_gemm_expand_ln_head(_l.ptr(x), lda, _l.ptr(w), ldb, _l.ptr(out), B, H, W, K, P, c, _l.ptr(gamma), _l.ptr(beta),
                                   _l.ptr(head_w), NC, eps, _l.ptr(mean), _l.ptr(rstd), _l.ptr(logits5), logits5.stride(0),
                                   _l.stream_ptr())
    if st == -2:
        return False
    _l.check(st, "mis_gemm_expand_ln_head")
    if prof is not None:
        e1.record()
        prec = 1 if (set_split_precision(-1) & 1) else 0
        prof.append((f"gemm_nt_kernel<64, 96, 4, {prec}, 4>", 2.0 * M * N * K, e0, e1, 4.0 * (M * K + N * K + (M * N if out is not None else 0))))
    return True


def transpose(src, dst):
    """dst[c][r] = src[r][c]."""
    L = _l.load()
    R, Cc, lds = _mat(src)
    _, _, ldd = _mat(dst)
    _l.check(L.mis_transpose(_l.ptr(src), lds, _l.ptr(dst), ldd, R, Cc, _l.stream_ptr()), "mis_transpose")


# ---------------------------------------------------------------- SwinUNETR encoder (csrc/swin3d.hip)
def win3d_gather(src, dst, B, dims, C, win, shift, inverse=False):
    """windows [B*nW, n, C] <- tokens [B*D*H*W, C] (zero padding to multiples of the window, cyclic shift); ``inverse``:
    tokens <- windows.  Each direction is the other's gradient."""
    L = _l.load()
    D, H, W = dims
    _l.check(L.mis_win3d_gather(_l.ptr(src), _l.ptr(dst), B, D, H, W, C, win[0], win[1], win[2], shift[0], shift[1],
                                shift[2], int(inverse), _l.stream_ptr()), "mis_win3d_gather")


def win3d_windows(B, dims, win):
    return int(_l.load().mis_win3d_windows(B, dims[0], dims[1], dims[2], win[0], win[1], win[2]))


def merge3d(src, dst, B, dims, C, inverse=False):
    """merged [B*D/2*H/2*W/2, 8C] <- tokens [B*D*H*W, C] in MONAI's v0.9 "merging" order; ``inverse``: token gradient."""
    L = _l.load()
    _l.check(L.mis_merge3d(_l.ptr(src), _l.ptr(dst), B, dims[0], dims[1], dims[2], C, int(inverse), _l.stream_ptr()),
             "mis_merge3d")


def win3d_attn_fwd(qkv, out, stats, table, region, BW, nW, n, nH):
    L = _l.load()
    _l.check(L.mis_win3d_attn_fwd(_l.ptr(qkv), qkv.stride(0), _l.ptr(out), out.stride(0), _l.ptr(stats), _l.ptr(table),
                                  _l.ptr(region), BW, nW, n, nH, _l.stream_ptr()), "mis_win3d_attn_fwd")


def win3d_attn_bwd(qkv, out, dout, dqkv, stats, table, region, dtable, BW, nW, n, nH, accumulate_table=False):
    L = _l.load()
    nb = L.mis_win3d_attn_workspace_bytes(BW, n, nH)
    ws = scratch(nb, "attn3d")
    _l.check(L.mis_win3d_attn_bwd(_l.ptr(qkv), qkv.stride(0), _l.ptr(out), _l.ptr(dout), dout.stride(0), _l.ptr(dqkv),
                                  dqkv.stride(0), _l.ptr(stats), _l.ptr(table), _l.ptr(region), _l.ptr(dtable),
                                  int(accumulate_table), BW, nW, n, nH, _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_win3d_attn_bwd")


class TransposeBatch:
    """All weight transposes of a network's backward in one launch (``mis_transpose_batch``).  ``jobs``: list of
    (src [R, C] dense, dst [C, R] dense); the device job table is built once (it holds raw pointers into the tensors)."""

    def __init__(self, jobs):
        import ctypes
        L = _l.load()
        nb = L.mis_transpose_job_bytes()
        host = (ctypes.c_char * (nb * len(jobs)))()
        first = 0
        for i, (src, dst) in enumerate(jobs):
            R, Cc = src.shape
            assert src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == (Cc, R)
            n = L.mis_transpose_job(ctypes.byref(host, i * nb), _l.ptr(src), _l.ptr(dst), R, Cc, first)
            if n < 0:
                _l.check(n, "mis_transpose_job")
            first += n
        self.n, self.tiles, self.keep = len(jobs), first, jobs
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()

    def run(self):
        _l.check(_l.load().mis_transpose_batch(_l.ptr(self.table), self.n, self.tiles, _l.stream_ptr()),
                 "mis_transpose_batch")


def layernorm_fwd(x, y, gamma, beta, mean, rstd, eps=1e-5):
    L = _l.load()
    M, C, ldx = _mat(x)
    _, _, ldy = _mat(y)
    _l.check(L.mis_layernorm_fwd(_l.ptr(x), ldx, _l.ptr(y), ldy, _l.ptr(gamma), _l.ptr(beta), _l.ptr(mean),
                                 _l.ptr(rstd), M, C, eps, _l.stream_ptr()), "mis_layernorm_fwd")


def layernorm_bwd(x, dy, dx, gamma, mean, rstd, dgamma, dbeta, accumulate_dx=False, accumulate_affine=False):
    L = _l.load()
    M, C, ldx = _mat(x)
    _, _, lddy = _mat(dy)
    _, _, lddx = _mat(dx)
    ws = scratch(L.mis_colreduce_workspace_bytes(M, C), "colreduce")
    _l.check(L.mis_layernorm_bwd(_l.ptr(x), ldx, _l.ptr(dy), lddy, _l.ptr(dx), lddx, _l.ptr(gamma), _l.ptr(mean),
                                 _l.ptr(rstd), _l.ptr(dgamma), _l.ptr(dbeta), M, C, int(accumulate_dx),
                                 int(accumulate_affine), _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_layernorm_bwd")


def layernorm_bwd_parts(x, dy, dx, gamma, mean, rstd, ws, accumulate_dx=False):
    """dx and the affine partials into the caller's workspace ``ws`` (``colreduce_workspace``); ``layernorm_bwd_final`` finishes."""
    L = _l.load()
    M, C, ldx = _mat(x)
    _, _, lddy = _mat(dy)
    _, _, lddx = _mat(dx)
    _l.check(L.mis_layernorm_bwd_parts(_l.ptr(x), ldx, _l.ptr(dy), lddy, _l.ptr(dx), lddx, _l.ptr(gamma), _l.ptr(mean),
                                       _l.ptr(rstd), M, C, int(accumulate_dx), _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_layernorm_bwd_parts")


def layernorm_bwd_residual_parts(x, dy, gin, d_shortcut, d_branch, gamma, mean, rstd, ws, rowscale=None, rows_per_scale=1,
                                 accumulate_shortcut=False):
    """LayerNorm backward + the backward of the residual add that produced its input (mis_layernorm_bwd_residual_parts);
    False: outside the fused form (C > 1536)."""
    L = _l.load()
    M, C, ldx = _mat(x)
    _, _, lddy = _mat(dy)
    ldg = _mat(gin)[2] if gin is not None else 0
    _, _, ldds = _mat(d_shortcut)
    _, _, lddb = _mat(d_branch)
    st = L.mis_layernorm_bwd_residual_parts(_l.ptr(x), ldx, _l.ptr(dy), lddy, _l.ptr(gin), ldg, _l.ptr(d_shortcut), ldds,
                                            int(accumulate_shortcut), _l.ptr(d_branch), lddb, _l.ptr(rowscale),
                                            int(rows_per_scale), _l.ptr(gamma), _l.ptr(mean), _l.ptr(rstd), M, C, _l.ptr(ws),
                                            ws.numel(), _l.stream_ptr())
    if st == -2:
        return False
    _l.check(st, "mis_layernorm_bwd_residual_parts")
    return True


def layernorm_bwd_final(ws, M, C, dgamma, dbeta, accumulate_affine=False):
    L = _l.load()
    _l.check(L.mis_layernorm_bwd_final(_l.ptr(ws), ws.numel(), M, C, _l.ptr(dgamma), _l.ptr(dbeta), int(accumulate_affine),
                                       _l.stream_ptr()), "mis_layernorm_bwd_final")


def colreduce_workspace(M, C):
    """A workspace of the caller's own for layernorm_bwd_parts / _final (the shared scratch is reused by the next call)."""
    return torch.empty(_l.load().mis_colreduce_workspace_bytes(M, C), dtype=torch.uint8, device="cuda")


def colsum(x, out, accumulate=False):
    L = _l.load()
    M, C, ldx = _mat(x)
    ws = scratch(L.mis_colreduce_workspace_bytes(M, C), "colreduce")
    _l.check(L.mis_colsum(_l.ptr(x), ldx, M, C, _l.ptr(out), int(accumulate), _l.ptr(ws), ws.numel(),
                          _l.stream_ptr()), "mis_colsum")


def gelu(x, out, dy=None):
    """dy None: out = gelu(x); else out = dy * gelu'(x).  Dense, same numel."""
    L = _l.load()
    assert x.is_contiguous() and out.is_contiguous() and (dy is None or dy.is_contiguous())
    _l.check(L.mis_gelu(_l.ptr(x), _l.ptr(dy), _l.ptr(out), x.numel(), int(dy is not None), _l.stream_ptr()),
             "mis_gelu")


def residual_fwd(a, y, out, rows_per_sample, drop_p=0.0, salt=0, state=None, scale_override=None):
    L = _l.load()
    M, C, lda = _mat(a)
    _, _, ldy = _mat(y)
    _, _, ldo = _mat(out)
    _l.check(L.mis_residual_droppath(_l.ptr(a), lda, _l.ptr(y), ldy, _l.ptr(out), ldo, None, 0, M, C,
                                     rows_per_sample, drop_p, salt, _l.ptr(state), _l.ptr(scale_override), 0,
                                     _l.stream_ptr()), "mis_residual_droppath")


def residual_bwd(dout, d_shortcut, d_branch, rows_per_sample, drop_p=0.0, salt=0, state=None,
                 scale_override=None, accumulate_shortcut=False):
    """d_shortcut (may be None) (+)= dout; d_branch = s_b * dout."""
    L = _l.load()
    mode = 2 if accumulate_shortcut else 1
    M, C, lda = _mat(dout)
    ldo = _mat(d_shortcut)[2] if d_shortcut is not None else 0
    _, _, ldo2 = _mat(d_branch)
    _l.check(L.mis_residual_droppath(_l.ptr(dout), lda, None, 0, _l.ptr(d_shortcut), ldo, _l.ptr(d_branch), ldo2,
                                     M, C, rows_per_sample, drop_p, salt, _l.ptr(state), _l.ptr(scale_override), mode,
                                     _l.stream_ptr()), "mis_residual_droppath")


def token_rearrange(src, dst, B, H, W, C, P, mode, inverse=False):
    L = _l.load()
    _, _, lds = _mat(src)
    _, _, ldd = _mat(dst)
    _l.check(L.mis_token_rearrange(_l.ptr(src), lds, _l.ptr(dst), ldd, B, H, W, C, P, mode, int(inverse),
                                   _l.stream_ptr()), "mis_token_rearrange")


def patch_im2col(x4, out, in_chans):
    """x4 [B,1|in_chans,H,W] (dense C,H,W) -> out [B*H/4*W/4, in_chans*16]; a single channel is repeated."""
    L = _l.load()
    B, Cs, H, W = x4.shape
    assert x4.stride(3) == 1 and x4.stride(2) == W and (Cs == 1 or x4.stride(1) == H * W)
    _l.check(L.mis_patch_im2col_c(_l.ptr(x4), x4.stride(0), _l.ptr(out), B, H, W, in_chans, Cs, _l.stream_ptr()),
             "mis_patch_im2col_c")


def head_fwd(x, w, logits5):
    """x [B*S, K] tokens, w [NC, K] -> logits [B, NC, 1, H, W]."""
    L = _l.load()
    M, K, ldx = _mat(x)
    B, NC = logits5.shape[0], logits5.shape[1]
    S = M // B
    _l.check(L.mis_head_fwd(_l.ptr(x), ldx, _l.ptr(w), _l.ptr(logits5), logits5.stride(0), B, S, K, NC,
                            _l.stream_ptr()), "mis_head_fwd")


def ln_head_fwd(x, gamma, beta, w, mean, rstd, logits5, eps=1e-5):
    """logits [B, NC, 1, H, W] = head(LayerNorm(x)) in one pass over x [B*S, C] (mis_ln_head_fwd); False when the shape
    is outside the fused form."""
    L = _l.load()
    M, C, ldx = _mat(x)
    B, NC = logits5.shape[0], logits5.shape[1]
    if C % 4 or C > 128 or not 2 <= NC <= 4 or M % B:
        return False
    _l.check(L.mis_ln_head_fwd(_l.ptr(x), ldx, _l.ptr(gamma), _l.ptr(beta), _l.ptr(w), _l.ptr(mean), _l.ptr(rstd),
                               _l.ptr(logits5), logits5.stride(0), B, M // B, C, NC, eps, _l.stream_ptr()), "mis_ln_head_fwd")
    return True


def ln_head_bwd(x, gamma, beta, w, mean, rstd, dlogits5, dx, dgamma, dbeta, dw, accumulate_dx=False, unshuffle=None):
    """``unshuffle = (H, W, P)``: ``x`` rows are the tokens of the pixel-shuffled (H P) x (W P) grid and ``dx`` is the gradient of
    the expand Linear's OUTPUT [B H W, P P C]: the rows are stored through the inverse shuffle (mis_ln_head_bwd_unshuffle)."""
    L = _l.load()
    M, C, ldx = _mat(x)
    _, _, lddx = _mat(dx)
    B, NC = dlogits5.shape[0], dlogits5.shape[1]
    ws = scratch(L.mis_ln_head_workspace_bytes(M, C, NC), "head")
    if unshuffle is not None:
        H, W, P = unshuffle
        assert M == B * H * P * W * P and dx.shape[0] == B * H * W and dx.shape[1] == P * P * C, (M, dx.shape, unshuffle)
        _l.check(L.mis_ln_head_bwd_unshuffle(_l.ptr(x), ldx, _l.ptr(gamma), _l.ptr(beta), _l.ptr(w), _l.ptr(mean), _l.ptr(rstd),
                                             _l.ptr(dlogits5), dlogits5.stride(0), _l.ptr(dx), lddx, int(accumulate_dx),
                                             _l.ptr(dgamma), _l.ptr(dbeta), _l.ptr(dw), 0, B, H, W, P, C, NC, _l.ptr(ws),
                                             ws.numel(), _l.stream_ptr()), "mis_ln_head_bwd_unshuffle")
        return
    _l.check(L.mis_ln_head_bwd(_l.ptr(x), ldx, _l.ptr(gamma), _l.ptr(beta), _l.ptr(w), _l.ptr(mean), _l.ptr(rstd),
                               _l.ptr(dlogits5), dlogits5.stride(0), _l.ptr(dx), lddx, int(accumulate_dx), _l.ptr(dgamma),
                               _l.ptr(dbeta), _l.ptr(dw), 0, B, M // B, C, NC, _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_ln_head_bwd")


def head_bwd(x, w, dlogits5, dx, dw, accumulate_dw=False):
    L = _l.load()
    M, K, ldx = _mat(x)
    _, _, lddx = _mat(dx)
    B, NC = dlogits5.shape[0], dlogits5.shape[1]
    S = M // B
    ws = scratch(L.mis_head_workspace_bytes(K, NC), "head")
    _l.check(L.mis_head_bwd(_l.ptr(x), ldx, _l.ptr(w), _l.ptr(dlogits5), dlogits5.stride(0), _l.ptr(dx), lddx,
                            _l.ptr(dw), int(accumulate_dw), B, S, K, NC, _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_head_bwd")


def window_attention_fwd(qkv, out, table, B, H, W, nH, shift, scale, window=7):
    L = _l.load()
    _, _, ldq = _mat(qkv)
    _, _, ldo = _mat(out)
    _l.check(L.mis_window_attention_fwd_ws(_l.ptr(qkv), ldq, _l.ptr(out), ldo, _l.ptr(table), B, H, W, nH, shift,
                                           scale, window, _l.stream_ptr()), "mis_window_attention_fwd_ws")


def window_attention_bwd(qkv, dout, dqkv, table, dtable, B, H, W, nH, shift, scale, accumulate_table=False, window=7):
    L = _l.load()
    _, _, ldq = _mat(qkv)
    _, _, ldo = _mat(dout)
    _, _, lddq = _mat(dqkv)
    nb = L.mis_window_attention_workspace_bytes_ws(B, H, W, nH, window)
    if nb < 0:
        _l.check(nb, "mis_window_attention_workspace_bytes_ws")
    ws = scratch(nb, "attn")
    _l.check(L.mis_window_attention_bwd_ws(_l.ptr(qkv), ldq, _l.ptr(dout), ldo, _l.ptr(dqkv), lddq, _l.ptr(table),
                                           _l.ptr(dtable), int(accumulate_table), B, H, W, nH, shift, scale, window,
                                           _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_window_attention_bwd_ws")


def window_attention_workspace(B, H, W, nH, window=7):
    L = _l.load()
    nb = L.mis_window_attention_workspace_bytes_ws(B, H, W, nH, window)
    if nb < 0:
        _l.check(nb, "mis_window_attention_workspace_bytes_ws")
    return torch.empty(nb, dtype=torch.uint8, device="cuda")


def window_attention_bwd_parts(qkv, dout, dqkv, table, ws, B, H, W, nH, shift, scale, window=7):
    L = _l.load()
    _, _, ldq = _mat(qkv)
    _, _, ldo = _mat(dout)
    _, _, lddq = _mat(dqkv)
    _l.check(L.mis_window_attention_bwd_parts_ws(_l.ptr(qkv), ldq, _l.ptr(dout), ldo, _l.ptr(dqkv), lddq, _l.ptr(table), B, H,
                                                 W, nH, shift, scale, window, _l.ptr(ws), ws.numel(), _l.stream_ptr()),
             "mis_window_attention_bwd_parts_ws")


def window_attention_dtable(ws, dtable, B, H, W, nH, accumulate_table=False, window=7):
    L = _l.load()
    _l.check(L.mis_window_attention_dtable_ws(_l.ptr(ws), ws.numel(), _l.ptr(dtable), int(accumulate_table), B, H, W, nH,
                                              window, _l.stream_ptr()), "mis_window_attention_dtable_ws")


# ---------------------------------------------------------------- UNETR (full attention, 3-D patch embedding)
def full_attention_fwd(qkv, out, stats, B, N, nH, scale):
    L = _l.load()
    _, _, ldq = _mat(qkv)
    _, _, ldo = _mat(out)
    _l.check(L.mis_full_attention_fwd(_l.ptr(qkv), ldq, _l.ptr(out), ldo, _l.ptr(stats), B, N, nH, scale,
                                      _l.stream_ptr()), "mis_full_attention_fwd")


def full_attention_bwd(qkv, dout, dqkv, stats, B, N, nH, scale):
    L = _l.load()
    _, _, ldq = _mat(qkv)
    _, _, ldo = _mat(dout)
    _, _, lddq = _mat(dqkv)
    ws = scratch(L.mis_full_attention_workspace_bytes(B, N, nH), "attn")
    _l.check(L.mis_full_attention_bwd(_l.ptr(qkv), ldq, _l.ptr(dout), ldo, _l.ptr(dqkv), lddq, _l.ptr(stats), B, N, nH,
                                      scale, _l.ptr(ws), ws.numel(), _l.stream_ptr()), "mis_full_attention_bwd")


def patch3d_im2col(x5, out, P):
    """x5 [B,1,H,W,D] -> out [B*(H/P)(W/P)(D/P), P^3] (einops 'b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)')."""
    L = _l.load()
    B, C, H, W, D = x5.shape
    assert C == 1 and x5.stride(4) == 1 and x5.stride(3) == D and x5.stride(2) == W * D
    _l.check(L.mis_patch3d_im2col(_l.ptr(x5), x5.stride(0), _l.ptr(out), B, H, W, D, P, _l.stream_ptr()),
             "mis_patch3d_im2col")


def add_rowcycle(x, pos, out, L_rows):
    L = _l.load()
    M, C, ldx = _mat(x)
    _, _, ldo = _mat(out)
    assert pos.is_contiguous() and pos.numel() == L_rows * C
    _l.check(L.mis_add_rowcycle(_l.ptr(x), ldx, _l.ptr(pos), _l.ptr(out), ldo, M, C, L_rows, _l.stream_ptr()),
             "mis_add_rowcycle")


def sum_rowcycle(dy, dpos, L_rows):
    L = _l.load()
    M, C, ld = _mat(dy)
    assert dpos.is_contiguous() and dpos.numel() == L_rows * C
    _l.check(L.mis_sum_rowcycle(_l.ptr(dy), ld, _l.ptr(dpos), M, C, L_rows, _l.stream_ptr()), "mis_sum_rowcycle")

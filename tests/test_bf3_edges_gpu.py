"""Edge semantics of the two arithmetic forms of the Linear GEMMs / window attention (mis_gemm_set_split_precision):
non-finite operands, denormals and a wide dynamic range inside one contraction.

What is asserted (and documented in DESIGN.md s.4 "bf16x3 edge cases"):

* NaN operands poison exactly the outputs whose contraction reads them -- both forms, same placement as torch fp32.
* +-inf operands: the fp32-MFMA form gives torch's result (inf with torch's sign, NaN where torch has NaN).  The bf16x3 form
  cuts x = h + m + l with m = x - h: for x = +-inf that is inf - inf = NaN, and even with m = l = 0 the piece products
  inf * (a zero low piece of the other operand) are NaN -- so an output that torch reports as +-inf is NaN here.  The
  PLACEMENT of the non-finite outputs is torch's; nothing finite is disturbed.
* denormal operands / results: both forms agree with float64 to an absolute 2^-126 per contraction element (a flushed
  piece or result is allowed), i.e. far below every tolerance of the path.
* 1e30 beside 1e-30 in one contraction: error <= 2e-6 * sum_k |a_k b_k| (the bound of an fp32 fmaf chain).
* operands below 2^-110 lose (at most) their lowest bf16 piece: relative 2^-16 of THAT operand, absolute < 2^-126.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t():
    from mis_hip import tops
    return tops


@pytest.fixture(params=[0, 7], ids=["fp32mfma", "bf16x3"])
def prec(request):
    tops = _t()
    prev = tops.set_split_precision(request.param)
    yield request.param
    tops.set_split_precision(prev)


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * 2 - 1


def _check_nonfinite(out, ref32, prec, what):
    """`out` (device result) against torch fp32 `ref32`: same non-finite placement; the fp32-MFMA form also the same kind/sign."""
    out = out.float().cpu()
    bad_o, bad_r = ~torch.isfinite(out), ~torch.isfinite(ref32)
    assert torch.equal(bad_o, bad_r), f"{what}: non-finite placement differs ({bad_o.sum().item()} vs {bad_r.sum().item()})"
    assert torch.isnan(out)[torch.isnan(ref32)].all(), f"{what}: a NaN of torch is not NaN here"
    if not prec & 1:
        inf_r = torch.isinf(ref32)
        assert torch.equal(out[inf_r], ref32[inf_r]), f"{what}: fp32-MFMA form must give torch's +-inf"
    fin = ~bad_r
    scale = ref32[fin].abs().max().item() if fin.any() else 1.0
    assert (out[fin] - ref32[fin]).abs().max().item() <= 1e-5 + 2e-4 * scale, f"{what}: finite outputs disturbed"


# shapes: general NT kernel (64 x 96 tiles), the short-contraction kernel (M large, K = 96), split-K (few tiles, long K)
@pytest.mark.parametrize("M,N,K", [(300, 192, 384), (50000, 288, 96), (1176, 768, 3072)])
def test_gemm_nt_nonfinite_operands(M, N, K, prec):
    tops = _t()
    A, B = _rand(M, K, seed=1), _rand(N, K, seed=2)
    A[3, 5] = float("nan")
    A[17, K - 1] = float("inf")
    A[M - 1, 0] = float("-inf")
    B[7, 11] = float("inf")              # column 7 of C: +-inf by the sign of A[:, 11] (NaN only where A is 0 / NaN / -inf + inf)
    B[N - 2, 3] = float("nan")
    ref = A @ B.t()
    C = torch.empty(M, N, device="cuda")
    tops.gemm(A.cuda(), B.cuda(), C)
    _check_nonfinite(C, ref, prec, "NT")
    # the pre-split weight path (B cut into planes once per pass) is the product path of the Linears
    b3 = tops.SplitB(B.cuda()).refresh()
    C2 = torch.empty(M, N, device="cuda")
    tops.gemm(A.cuda(), B.cuda(), C2, b3=b3)
    _check_nonfinite(C2, ref, prec, "NT pre-split")


@pytest.mark.parametrize("T,Cout,Cin", [(20000, 96, 288), (3137, 100, 36)])
def test_gemm_dw_nonfinite_operands(T, Cout, Cin, prec):
    tops = _t()
    X, dY = _rand(T, Cin, seed=3), _rand(T, Cout, seed=4)
    X[5, 2] = float("inf")
    X[T - 1, Cin - 1] = float("nan")
    dY[100, 1] = float("-inf")
    dY[7, Cout - 1] = float("nan")
    ref = dY.t() @ X
    refb = dY.sum(0)
    dW, db = torch.empty(Cout, Cin, device="cuda"), torch.empty(Cout, device="cuda")
    tops.gemm_dw(dY.cuda(), X.cuda(), dW, db)
    _check_nonfinite(dW, ref, prec, "TN dW")
    _check_nonfinite(db, refb, 0, "db")           # the bias gradient is a plain fp32 column sum in both forms


def test_gemm_denormals_and_wide_dynamic_range(prec):
    tops = _t()
    M, N, K = 288, 96, 96          # widths % 96 == 0: the dW form below runs the register-only TN kernel
    A, B = _rand(M, K, seed=5), _rand(N, K, seed=6)
    # row 0: 1e30 beside 1e-30 in one contraction; row 1: fp32 denormals; row 2: tiny normals whose lowest piece is denormal;
    # row 3: products that land in the denormal range
    A[0, ::2] *= 1e30
    A[0, 1::2] *= 1e-30
    A[1] = _rand(K, seed=7) * 1e-40
    A[2] = _rand(K, seed=8) * 1e-35
    A[3] = _rand(K, seed=9) * 1e-25
    B[5] *= 1e-15                                # C[3, 5] ~ 1e-40
    ref = A.double() @ B.double().t()
    mag = A.double().abs() @ B.double().abs().t()
    C = torch.empty(M, N, device="cuda")
    tops.gemm(A.cuda(), B.cuda(), C)
    err = (C.cpu().double() - ref).abs()
    tiny = K * 2.0 ** -126 * max(1.0, B.abs().max().item())
    assert torch.isfinite(C).all()
    assert (err <= 2e-6 * mag + tiny).all(), f"worst excess {(err - 2e-6 * mag - tiny).max().item():.3e}"
    # the wide-range row on its own: relative to its own magnitude
    assert (err[0] <= 2e-6 * mag[0]).all()
    # dW form: the same operands as contraction-major rows
    dW = torch.empty(N, M, device="cuda")
    At, Bt = A.t().contiguous(), B.t().contiguous()          # [K, M], [K, N]: dW[n][m] = sum_k Bt[k][n] At[k][m]
    tops.gemm(Bt.cuda(), At.cuda(), dW, trans=True)
    errw = (dW.cpu().double() - ref.t()).abs()
    assert (errw <= 2e-6 * mag.t() + tiny).all()


def test_window_attention_nonfinite_rows(prec):
    """An inf / NaN in q, k or v of one (window, head) poisons that unit's outputs only -- the same rows torch poisons."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_token_kernels_gpu import _ref_window_attention
    tops = _t()
    B, H, W, nH, shift, ws = 2, 14, 14, 3, 3, 7
    C = nH * 32
    qkv = _rand(B * H * W, 3 * C, seed=22) * 1.5
    table = _rand((2 * ws - 1) ** 2, nH, seed=23) * 0.5
    qkv[5, 3] = float("inf")                     # q of token 5, head 0
    qkv[40, C + 32 + 7] = float("nan")           # k of token 40, head 1
    qkv[300, 2 * C + 64 + 1] = float("-inf")     # v of token 300, head 2
    scale = 32 ** -0.5
    ref = _ref_window_attention(qkv, table, B, H, W, nH, shift, scale, ws)
    out = torch.empty(B * H * W, C, device="cuda")
    tops.window_attention_fwd(qkv.cuda(), out, table.cuda(), B, H, W, nH, shift, scale, window=ws)
    o = out.cpu()
    bad_o, bad_r = ~torch.isfinite(o), ~torch.isfinite(ref)
    assert torch.equal(bad_o, bad_r)
    fin = ~bad_r
    assert (o[fin] - ref[fin]).abs().max().item() <= 1e-5 + 1e-4 * ref[fin].abs().max().item()

"""Op-level parity of the token-major (SwinUnet) HIP kernels against stock torch CPU fp32/fp64 ops."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _t():
    from mis_hip import tops
    return tops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.fixture(params=[0, 7], ids=["fp32mfma", "bf16x3"])
def prec(request):
    """Arithmetic of the Linear GEMMs and the window attention (mis_gemm_set_split_precision): 0 = v_mfma_f32_16x16x4_f32, 7 =
    bf16x3 split products everywhere (the default).  Both must meet the same bounds against float64."""
    tops = _t()
    prev = tops.set_split_precision(request.param)
    yield request.param
    tops.set_split_precision(prev)


def _close(a, b, rtol=2e-4, atol=1e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * b.abs().max().item(), f"max err {err:.3e} vs scale {b.abs().max().item():.3e}"


@pytest.mark.parametrize("M,N,K", [(256, 128, 96), (3136, 288, 96), (200, 96, 48), (130, 1536, 96), (49, 768, 3072),
                                   (1000, 4, 96), (5000, 192, 384), (777, 576, 192), (100, 96, 100),
                                   # few tiles, long K: the NT form splits K too (ragged last slice for K = 1000)
                                   (1176, 768, 3072), (300, 200, 1000),
                                   # short contraction over many rows: gemm_nt_short_kernel (64-row tiles, ragged last)
                                   (50000, 288, 96), (33000, 384, 192),
                                   (16401, 576, 1536)])          # long K over many ragged row tiles
def test_gemm_nt_and_tn(M, N, K, prec):
    tops = _t()
    A, B, bias = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    ref = (A.double() @ B.double().t() + bias.double())
    Ad, Bd = A.cuda(), B.cuda()
    from mis_hip import lib
    name = ctypes.create_string_buffer(96)
    assert lib.load().mis_gemm_nt_kernel_name(M, N, K, 0, name, 96) == 0
    # the short-contraction kernel (K = 16 steps) serves K <= 192 in the fp32 form and K <= 96 in the bf16x3 form, whose K = 32
    # products of the general kernel win from K = 192 on (last template argument: 1 = bf16x3)
    assert name.value.decode().startswith("gemm_nt_short_kernel<") == (M >= 33000 and (not prec & 1 or K <= 96))
    if not name.value.decode().startswith("gemm_nt_short_kernel<"):
        assert name.value.decode().endswith(", %d, 4>" % (prec & 1))      # <BM, BN, EP, arithmetic, waves per workgroup>
    C = torch.empty(M, N, device="cuda")
    tops.gemm(Ad, Bd, C, bias=bias.cuda())
    _close(C, ref)
    tops.gemm(Ad, Bd, C, accumulate=True)           # C += A B^T
    _close(C, 2 * ref - bias.double())
    # strided operands / outputs (column slices of wider buffers)
    wide = torch.zeros(M, N + 32, device="cuda")
    tops.gemm(Ad, Bd, wide[:, 32:])
    _close(wide[:, 32:], ref - bias.double())
    assert wide[:, :32].abs().max().item() == 0
    # TN: dW[N,K] = dY[M,N]^T @ X[M,K]  (contraction over the M tokens, split-K path for small outputs)
    if N % 4 == 0:
        dY = _rand(M, N, seed=4)
        refw = dY.double().t() @ A.double()
        dW = torch.empty(N, K, device="cuda")
        tops.gemm(dY.cuda(), Ad, dW, trans=True)
        _close(dW, refw, rtol=3e-4)
        tops.gemm(dY.cuda(), Ad, dW, trans=True, accumulate=True)
        _close(dW, 2 * refw, rtol=3e-4)


def test_gemm_tn_split_k_large(prec):
    tops = _t()
    M, N, K = 20000, 96, 288     # M = tokens (contraction), small output -> many K slices
    X, dY = _rand(M, K, seed=5), _rand(M, N, seed=6)
    dW = torch.empty(N, K, device="cuda")
    tops.gemm(dY.cuda(), X.cuda(), dW, trans=True)
    _close(dW, dY.double().t() @ X.double(), rtol=3e-4)
    dW2 = torch.empty(N, K, device="cuda")
    tops.gemm(dY.cuda(), X.cuda(), dW2, trans=True)
    assert torch.equal(dW, dW2)        # deterministic


@pytest.mark.parametrize("T,Cout,Cin", [(20000, 96, 288), (150528, 288, 96), (2352, 768, 3072), (49, 1536, 768),
                                        (3137, 100, 36), (9408, 1536, 384)])
def test_gemm_dw_weight_and_bias_gradient_in_one_pass(T, Cout, Cin, prec):
    """mis_gemm_dw: dW = dy^T x and db = dy.sum(0) from one read of dy (nn.Linear backward), split-K and direct forms,
    ragged tiles, accumulate; bit-identical dW to the plain TN form and deterministic."""
    tops = _t()
    X, dY = _rand(T, Cin, seed=15).cuda(), (_rand(T, Cout, seed=16) + 0.25).cuda()
    dW, db = torch.full((Cout, Cin), float("nan"), device="cuda"), torch.full((Cout,), float("nan"), device="cuda")
    tops.gemm_dw(dY, X, dW, db)
    _close(dW, dY.double().t() @ X.double(), rtol=3e-4)
    refb = dY.double().sum(0)
    assert (db.double() - refb).abs().max().item() <= 2e-5 * max(1.0, refb.abs().max().item())
    plain = torch.empty_like(dW)
    tops.gemm(dY, X, plain, trans=True)
    assert torch.equal(dW, plain)
    dW2, db2 = dW.clone(), db.clone()
    tops.gemm_dw(dY, X, dW2, db2, accumulate=True)
    assert torch.allclose(dW2, 2 * dW, rtol=1e-6, atol=1e-6) and torch.allclose(db2, 2 * db, rtol=1e-6, atol=1e-6)
    dW3, db3 = torch.empty_like(dW), torch.empty_like(db)
    tops.gemm_dw(dY, X, dW3, db3)
    assert torch.equal(dW, dW3) and torch.equal(db, db3)


@pytest.mark.parametrize("T", [1001, 1003, 2054, 9419])
def test_register_only_tn_gemm_never_reads_past_its_row_range(T, prec):
    """gemm_tn_reg_kernel (widths % 96 == 0) keeps 4 row groups in flight and lets the buffer descriptor's range check
    zero the loads past a wave's last row -- the K tail inside a group of 4 rows and the ring's surplus groups when
    ngroups % 4 != 0.  The rows behind row T of the SAME allocations hold NaNs: a load that escapes the range poisons dW/db."""
    tops = _t()
    Cout, Cin, pad = 192, 96, 64
    Xf = torch.full((T + pad, Cin), float("nan"), device="cuda")
    Yf = torch.full((T + pad, Cout), float("nan"), device="cuda")
    Xf[:T], Yf[:T] = _rand(T, Cin, seed=21).cuda(), (_rand(T, Cout, seed=22) + 0.25).cuda()
    X, dY = Xf[:T], Yf[:T]
    dW, db = torch.empty(Cout, Cin, device="cuda"), torch.empty(Cout, device="cuda")
    tops.gemm_dw(dY, X, dW, db)
    assert torch.isfinite(dW).all() and torch.isfinite(db).all()
    _close(dW, dY.double().t() @ X.double(), rtol=3e-4)
    assert (db.double() - dY.double().sum(0)).abs().max().item() <= 2e-5 * float(dY.double().sum(0).abs().max())


@pytest.mark.parametrize("B,S,C,NC", [(2, 3136, 96, 4), (3, 1000, 96, 2), (1, 50, 128, 3), (2, 77, 36, 4)])
def test_layernorm_and_output_head_in_one_pass(B, S, C, NC):
    """mis_ln_head_{fwd,bwd}: nn.LayerNorm(C) + the bias-free 1x1 output convolution of SwinUnet's tail (reference
    swin_transformer_unet_skip_expand_decoder_sys.py:390-409, :671) against torch autograd in double; ragged slabs, C < 128
    (idle lanes), accumulate into dx, deterministic."""
    tops = _t()
    x = (_rand(B * S, C, seed=21, scale=2.0) + 0.3).double().requires_grad_(True)
    g = (1 + 0.2 * _rand(C, seed=22)).double().requires_grad_(True)
    b = (0.1 * _rand(C, seed=23)).double().requires_grad_(True)
    w = (_rand(NC, C, seed=24, scale=0.3)).double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (C,), g, b, 1e-5)
    logits = (y @ w.t()).view(B, S, NC).permute(0, 2, 1)                 # [B, NC, S]
    dl = _rand(B, NC, S, seed=25).double()
    logits.backward(dl)
    xd, gd, bd, wd = x.detach().float().cuda(), g.detach().float().cuda(), b.detach().float().cuda(), w.detach().float().cuda()
    mean, rstd = torch.empty(B * S, device="cuda"), torch.empty(B * S, device="cuda")
    out = torch.full((B, NC, 1, 1, S), float("nan"), device="cuda")
    assert tops.ln_head_fwd(xd, gd, bd, wd, mean, rstd, out)
    _close(out.view(B, NC, S), logits.detach())
    dx = torch.full((B * S, C), float("nan"), device="cuda")
    dg, db, dw = (torch.full((C,), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda"),
                  torch.full((NC, C), float("nan"), device="cuda"))
    dld = dl.float().cuda().view(B, NC, 1, 1, S)
    tops.ln_head_bwd(xd, gd, bd, wd, mean, rstd, dld, dx, dg, db, dw)
    _close(dx, x.grad, rtol=3e-4)
    _close(dg, g.grad, rtol=3e-4)
    _close(db, b.grad, rtol=3e-4)
    _close(dw, w.grad, rtol=3e-4)
    dx2, dg2, db2, dw2 = dx.clone(), torch.empty_like(dg), torch.empty_like(db), torch.empty_like(dw)
    tops.ln_head_bwd(xd, gd, bd, wd, mean, rstd, dld, dx2, dg2, db2, dw2, accumulate_dx=True)
    assert torch.allclose(dx2, 2 * dx, rtol=1e-6, atol=1e-7)
    assert torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(dw, dw2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,P,C,NC", [(2, 7, 7, 4, 96, 4), (1, 5, 3, 4, 96, 4), (3, 4, 6, 2, 48, 2)])
def test_output_head_backward_stores_through_the_inverse_pixel_shuffle(B, H, W, P, C, NC):
    """mis_ln_head_bwd_unshuffle: dx rows go straight into the expand Linear's output gradient [B H W, P P C]
    (FinalPatchExpand_X4's 'b h w (p1 p2 c) -> b (h p1) (w p2) c', reference
    swin_transformer_unet_skip_expand_decoder_sys.py:390-409) — bit-identical to mis_ln_head_bwd followed by the inverse
    mis_token_rearrange, parameter gradients untouched by the addressing, with and without accumulation."""
    tops = _t()
    M = B * H * P * W * P
    x = (_rand(M, C, seed=31, scale=2.0) + 0.3).cuda()
    g, b, w = (1 + 0.2 * _rand(C, seed=32)).cuda(), (0.1 * _rand(C, seed=33)).cuda(), _rand(NC, C, seed=34, scale=0.3).cuda()
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    out = torch.empty(B, NC, 1, H * P, W * P, device="cuda")
    assert tops.ln_head_fwd(x, g, b, w, mean, rstd, out)
    dl = _rand(B, NC, 1, H * P, W * P, seed=35).cuda()

    def params():
        return torch.full((C,), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda"), \
            torch.full((NC, C), float("nan"), device="cuda")
    dx = torch.full((M, C), float("nan"), device="cuda")
    dg, db, dw = params()
    tops.ln_head_bwd(x, g, b, w, mean, rstd, dl, dx, dg, db, dw)
    ref = torch.empty(B * H * W, P * P * C, device="cuda")
    tops.token_rearrange(dx, ref, B, H, W, C, P, 1, inverse=True)
    got = torch.full((B * H * W, P * P * C), float("nan"), device="cuda")
    dg2, db2, dw2 = params()
    tops.ln_head_bwd(x, g, b, w, mean, rstd, dl, got, dg2, db2, dw2, unshuffle=(H, W, P))
    assert torch.equal(got, ref)
    assert torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(dw, dw2)
    base = _rand(B * H * W, P * P * C, seed=36).cuda()
    acc = base.clone()
    tops.ln_head_bwd(x, g, b, w, mean, rstd, dl, acc, dg2, db2, dw2, accumulate_dx=True, unshuffle=(H, W, P))
    assert torch.equal(acc, base + ref)


@pytest.mark.parametrize("M,C", [(784, 96), (50, 1536), (3137, 384)])
def test_layernorm(M, C):
    tops = _t()
    x = (_rand(M, C, seed=7, scale=2.0) + 0.5).requires_grad_(True)
    g = (1 + 0.2 * _rand(C, seed=8)).requires_grad_(True)
    b = (0.1 * _rand(C, seed=9)).requires_grad_(True)
    y = F.layer_norm(x, (C,), g, b, 1e-5)
    dy = _rand(M, C, seed=10)
    y.backward(dy)
    xd, gd, bd = x.detach().cuda(), g.detach().cuda(), b.detach().cuda()
    yd = torch.empty(M, C, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    tops.layernorm_fwd(xd, yd, gd, bd, mean, rstd)
    _close(yd, y)
    dx = torch.ones(M, C, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    tops.layernorm_bwd(xd, dy.cuda(), dx, gd, mean, rstd, dg, db)
    _close(dx, x.grad, rtol=3e-4)
    _close(dg, g.grad, rtol=3e-4, atol=1e-4)
    _close(db, b.grad, rtol=3e-4, atol=1e-4)
    tops.layernorm_bwd(xd, dy.cuda(), dx, gd, mean, rstd, dg, db, accumulate_dx=True, accumulate_affine=True)
    _close(dx, 2 * x.grad, rtol=3e-4)
    _close(dg, 2 * g.grad, rtol=3e-4, atol=2e-4)
    # two halves (mis_layernorm_bwd_parts / _final), the second on another stream: bit-identical to the one-call form
    dx1, dg1, db1 = torch.empty(M, C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    tops.layernorm_bwd(xd, dy.cuda(), dx1, gd, mean, rstd, dg1, db1)
    ws_own = tops.colreduce_workspace(M, C)
    dx2, dg2, db2 = torch.full_like(dx1, float("nan")), torch.full_like(dg1, float("nan")), torch.full_like(db1, float("nan"))
    tops.layernorm_bwd_parts(xd, dy.cuda(), dx2, gd, mean, rstd, ws_own)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tops.layernorm_bwd_final(ws_own, M, C, dg2, db2)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    bsum = torch.empty(C, device="cuda")
    tops.colsum(dy.cuda(), bsum)
    _close(bsum, dy.double().sum(0), rtol=1e-5, atol=1e-4)


def test_gelu_and_residual():
    tops = _t()
    x = (_rand(64, 384, seed=11, scale=3.0)).requires_grad_(True)
    y = F.gelu(x)
    dy = _rand(64, 384, seed=12)
    y.backward(dy)
    out = torch.empty(64, 384, device="cuda")
    tops.gelu(x.detach().cuda(), out)
    _close(out, y, rtol=1e-5, atol=1e-6)
    tops.gelu(x.detach().cuda(), out, dy=dy.cuda())
    _close(out, x.grad, rtol=1e-5, atol=1e-6)
    # residual with injected per-sample DropPath scales, and the Philox path
    from mis_hip import ops
    a, br = _rand(4 * 49, 96, seed=13), _rand(4 * 49, 96, seed=14)
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0])
    o = torch.empty(4 * 49, 96, device="cuda")
    tops.residual_fwd(a.cuda(), br.cuda(), o, 49, drop_p=0.2, scale_override=sc.cuda())
    _close(o, a + br * sc.repeat_interleave(49)[:, None], rtol=1e-6, atol=1e-7)
    ds, db = torch.empty_like(o), torch.empty_like(o)
    tops.residual_bwd(o, ds, db, 49, drop_p=0.2, scale_override=sc.cuda())
    assert torch.equal(ds, o)
    _close(db, o.cpu() * sc.repeat_interleave(49)[:, None], rtol=1e-6, atol=1e-7)
    st = ops.new_step_state()
    ops.step_init(st, 3, 0, 0.01, 30000, 0.99, 0.1, 200.0)
    big_a, big_b = torch.zeros(4096 * 4, 96, device="cuda"), torch.ones(4096 * 4, 96, device="cuda")
    o2 = torch.empty_like(big_a)
    tops.residual_fwd(big_a, big_b, o2, 4, drop_p=0.25, salt=9, state=st)
    per_sample = o2.view(4096, 4 * 96)
    assert ((per_sample.min(1).values == per_sample.max(1).values)).all()      # one scale per sample
    kept = (per_sample[:, 0] != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.03
    assert torch.allclose(per_sample[per_sample[:, 0] != 0][:, 0], torch.tensor(1 / 0.75, device="cuda"))


def test_rearrange_and_im2col():
    tops = _t()
    from einops import rearrange
    B, H, W, C = 2, 8, 12, 16
    x = _rand(B, H, W, C, seed=15)
    # PatchMerging gather (reference order x0,x1,x2,x3 = (0,0),(1,0),(0,1),(1,1))
    ref = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    out = torch.empty(B * H * W // 4, 4 * C, device="cuda")
    tops.token_rearrange(x.view(-1, C).cuda(), out, B, H, W, C, 2, 0)
    assert torch.equal(out.cpu(), ref.reshape(-1, 4 * C))
    back = torch.empty(B * H * W, C, device="cuda")
    tops.token_rearrange(out, back, B, H, W, C, 2, 0, inverse=True)
    assert torch.equal(back.cpu(), x.view(-1, C))
    # PatchExpand shuffle, P = 2 and 4
    for P in (2, 4):
        xin = _rand(B, H, W, P * P * C, seed=16)
        ref = rearrange(xin, 'b h w (p1 p2 c)-> b (h p1) (w p2) c', p1=P, p2=P, c=C)
        out = torch.empty(B * H * P * W * P, C, device="cuda")
        tops.token_rearrange(xin.view(-1, P * P * C).cuda(), out, B, H, W, C, P, 1)
        assert torch.equal(out.cpu(), ref.reshape(-1, C))
        back = torch.empty(B * H * W, P * P * C, device="cuda")
        tops.token_rearrange(out, back, B, H, W, C, P, 1, inverse=True)
        assert torch.equal(back.cpu(), xin.view(-1, P * P * C))
    # PatchEmbed: conv k4 s4 on the 1->3 repeated image == im2col rows @ W^T
    img = _rand(2, 1, 16, 24, seed=17)
    w = _rand(96, 3, 4, 4, seed=18)
    ref = F.conv2d(img.repeat(1, 3, 1, 1), w, stride=4).flatten(2).transpose(1, 2).reshape(-1, 96)
    cols = torch.empty(2 * 4 * 6, 48, device="cuda")
    tops.patch_im2col(img.cuda(), cols, 3)
    y = torch.empty(2 * 4 * 6, 96, device="cuda")
    tops.gemm(cols, w.view(96, 48).cuda(), y)
    _close(y, ref)
    # 3-channel input (the other branch of vision_transformer.py:48-50): every channel from its own plane
    img3 = _rand(2, 3, 16, 24, seed=19)
    ref3 = F.conv2d(img3, w, stride=4).flatten(2).transpose(1, 2).reshape(-1, 96)
    tops.patch_im2col(img3.cuda(), cols, 3)
    tops.gemm(cols, w.view(96, 48).cuda(), y)
    _close(y, ref3)


def test_output_head():
    tops = _t()
    B, S, K, NC = 2, 14 * 14, 96, 4
    x = _rand(B * S, K, seed=19).requires_grad_(True)
    w = _rand(NC, K, seed=20).requires_grad_(True)
    y = (x @ w.t()).view(B, S, NC).permute(0, 2, 1)            # [B, NC, S]
    dy = _rand(B, NC, S, seed=21)
    y.backward(dy)
    lg = torch.empty(B, NC, 1, 14, 14, device="cuda")
    tops.head_fwd(x.detach().cuda(), w.detach().cuda(), lg)
    _close(lg.view(B, NC, S), y)
    dx, dw = torch.empty(B * S, K, device="cuda"), torch.empty(NC, K, device="cuda")
    tops.head_bwd(x.detach().cuda(), w.detach().cuda(), dy.view(B, NC, 1, 14, 14).cuda().contiguous(), dx, dw)
    _close(dx, x.grad)
    _close(dw, w.grad, rtol=3e-4)


def _ref_window_attention(qkv, table, B, H, W, nH, shift, scale, ws=7):
    """torch restatement of WindowAttention.forward + the roll/partition plumbing (natural token order in/out)."""
    C, n = nH * 32, ws * ws
    x = qkv.view(B, H, W, 3 * C)
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.view(B, H // ws, ws, W // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, n, 3, nH, 32).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * scale, xw[1], xw[2]
    attn = q @ k.transpose(-2, -1)
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + ws - 1
    idx = rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]
    attn = attn + table[idx.view(-1)].view(n, n, nH).permute(2, 0, 1).unsqueeze(0)
    if shift:
        img = torch.zeros(1, H, W, 1)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for ws_ in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, ws_, :] = cnt
                cnt += 1
        mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, n)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        am = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0).to(attn.dtype)
        nW = am.shape[0]
        attn = (attn.view(B, nW, nH, n, n) + am.unsqueeze(1).unsqueeze(0)).view(-1, nH, n, n)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(-1, ws, ws, C)
    o = o.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * H * W, C)


@pytest.mark.parametrize("B,H,W,nH,shift,ws", [(2, 14, 14, 3, 0, 7), (3, 14, 21, 2, 3, 7), (2, 7, 7, 6, 0, 7),
                                                (9, 28, 28, 3, 3, 7),
                                                # enough (sample, window, head) units that the backward sums dS over 4
                                                # samples per wave, with a ragged last chunk (18 = 4*4 + 2)
                                                (18, 56, 56, 3, 3, 7),
                                                # window 8 (64 tokens: IMG_SIZE 256 / WINDOW_SIZE 8), shift 4
                                                (2, 16, 16, 3, 0, 8), (3, 16, 24, 2, 4, 8), (2, 8, 8, 6, 0, 8),
                                                (5, 64, 64, 3, 4, 8)])
def test_window_attention(B, H, W, nH, shift, ws, prec):
    tops = _t()
    C = nH * 32
    qkv = _rand(B * H * W, 3 * C, seed=22, scale=1.5).double().requires_grad_(True)
    table = (_rand((2 * ws - 1) ** 2, nH, seed=23, scale=0.5)).double().requires_grad_(True)
    scale = 32 ** -0.5
    ref = _ref_window_attention(qkv, table, B, H, W, nH, shift, scale, ws)
    dout = _rand(B * H * W, C, seed=24).double()
    ref.backward(dout)
    qd, td = qkv.detach().float().cuda(), table.detach().float().cuda()
    out = torch.empty(B * H * W, C, device="cuda")
    tops.window_attention_fwd(qd, out, td, B, H, W, nH, shift, scale, window=ws)
    _close(out, ref, rtol=1e-4, atol=1e-5)
    dqkv = torch.empty(B * H * W, 3 * C, device="cuda")
    dt = torch.empty((2 * ws - 1) ** 2, nH, device="cuda")
    tops.window_attention_bwd(qd, dout.float().cuda(), dqkv, td, dt, B, H, W, nH, shift, scale, window=ws)
    _close(dqkv, qkv.grad, rtol=2e-4, atol=1e-5)
    _close(dt, table.grad, rtol=2e-4, atol=1e-4)
    # the same backward in two halves (mis_window_attention_bwd_parts_ws / _dtable_ws), the second on another stream with a
    # workspace of the caller's own: bit-identical
    ws_own = tops.window_attention_workspace(B, H, W, nH, window=ws)
    dqkv2, dt2 = torch.full_like(dqkv, float("nan")), torch.full_like(dt, float("nan"))
    tops.window_attention_bwd_parts(qd, dout.float().cuda(), dqkv2, td, ws_own, B, H, W, nH, shift, scale, window=ws)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tops.window_attention_dtable(ws_own, dt2, B, H, W, nH, window=ws)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(dqkv2, dqkv) and torch.equal(dt2, dt)


@pytest.mark.parametrize("B,H,K,P,c", [(2, 56, 96, 4, 96), (24, 28, 192, 2, 96), (3, 14, 384, 2, 192), (2, 7, 768, 2, 384)])
def test_gemm_expand_equals_gemm_plus_pixel_shuffle(B, H, K, P, c, prec):
    """mis_gemm_expand (PatchExpand / FinalPatchExpand_X4: Linear + 'b h w (p1 p2 c) -> b (h p1) (w p2) c') is
    bit-identical to mis_gemm followed by mis_token_rearrange; shapes mis_gemm would split over K are refused."""
    tops = _t()
    M, N = B * H * H, P * P * c
    x, w = _rand(M, K, seed=31).cuda(), _rand(N, K, seed=32).cuda()
    e = torch.empty(M, N, device="cuda")
    ref = torch.empty(M * P * P, c, device="cuda")
    tops.gemm(x, w, e)
    tops.token_rearrange(e, ref, B, H, H, c, P, 1)
    out = torch.full((M * P * P, c), float("nan"), device="cuda")
    fused = tops.gemm_expand(x, w, out, B, H, H, P, c)
    from mis_hip import lib as _l
    split = _l.load().mis_gemm_workspace_bytes(M, N, K, 0) > 0
    assert fused == (not split)
    if fused:
        assert torch.equal(out, ref)
    # reference semantics of the shuffle itself
    want = (x.double() @ w.double().t()).view(B, H, H, P, P, c).permute(0, 1, 3, 2, 4, 5).reshape(M * P * P, c)
    _close(ref, want)


@pytest.mark.parametrize("M,N,K", [(6272, 384, 96), (6272, 96, 384), (200, 96, 48), (98, 3072, 768), (98, 768, 3072),
                                   (33000, 384, 96), (65570, 96, 96),       # the short-contraction kernel's epilogues
                                   (20000, 768, 192), (16400, 192, 768)])   # many row tiles, ragged last one, K = 192 / 768
def test_gemm_fused_epilogues(M, N, K, prec):
    """mis_gemm_ex: GELU forward / backward and DropPath + residual add in the NT GEMM's epilogue (also through the
    split-K reduction for the deep-stage shapes) against torch in float64."""
    tops = _t()
    A, W = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=K ** -0.5)
    bias = _rand(N, seed=33)
    v = A.double() @ W.double().t() + bias.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    # 1: C = v, C2 = gelu(v)
    C, C2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    assert tops.gemm_ex(Ad, Wd, C, tops.EP_GELU_FWD, bias=bd, C2=C2)
    _close(C, v)
    _close(C2, F.gelu(v))
    # 2: C = (A @ W^T) * gelu'(E1)
    h = _rand(M, N, seed=34, scale=1.5).double().requires_grad_(True)
    F.gelu(h).backward(torch.ones(M, N, dtype=torch.float64))
    assert tops.gemm_ex(Ad, Wd, C, tops.EP_GELU_BWD, E1=h.detach().float().cuda())
    _close(C, (A.double() @ W.double().t()) * h.grad)
    # 3: C = E1 + rowscale[row // rps] * v, with E1 a column slice of a wider buffer (the decoder's concat)
    rps = M // 2 if M % 2 == 0 else M
    wide = _rand(M, N + 32, seed=35)
    sc = torch.tensor([0.0, 1.25] if rps < M else [1.25])
    assert tops.gemm_ex(Ad, Wd, C, tops.EP_RESIDUAL, bias=bd, E1=wide.cuda()[:, 16:16 + N], rowscale=sc.cuda(),
                        rows_per_scale=rps)
    _close(C, wide[:, 16:16 + N].double() + sc.repeat_interleave(rps)[:, None].double() * v)
    assert tops.gemm_ex(Ad, Wd, C, tops.EP_RESIDUAL, bias=bd, E1=wide.cuda()[:, 16:16 + N])
    _close(C, wide[:, 16:16 + N].double() + v)


def test_swin_step_fused_equals_unfused():
    """The epilogue fusions change where the element-wise work runs, not its arithmetic: a Mean-Teacher step of SwinUnet
    with DropPath active (device RNG) gives the same losses, gradients and weights with MIS_SWIN_FUSE (and the residual
    backward inside the LayerNorm backward, MIS_SWIN_LNRES) on and off."""
    from config import lite_config
    from mis_hip import swin_plan
    from mis_hip.step import MeanTeacherTrainer
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.swin import OracleSwinUnet
    sd0 = filler.fill_state_dict(OracleSwinUnet(4).new_state())
    vol = filler.image((4, 1, 224, 224), "volume").cuda()
    lab = filler.labels((4, 224, 224), 4, torch.uint8).cuda()
    res = []
    for fuse in (7, 0):
        swin_plan.FUSE = fuse
        swin_plan.LNRES = bool(fuse)
        try:
            m, e = SwinUnet(lite_config(), num_classes=4), SwinUnet(lite_config(), num_classes=4)
            m.load_state_dict(sd0); e.load_state_dict(sd0)
            tr = MeanTeacherTrainer(m, e, labeled_bs=2, num_classes=4, cons_start_iter=0, seed=11, iter_num=1500)
            for _ in range(2):
                tr.step(vol, lab)
            torch.cuda.synchronize()
            res.append((tr.losses(), m.flat_grad.clone(), m.flat_param.clone(), e.flat_param.clone()))
        finally:
            swin_plan.FUSE = 7
            swin_plan.LNRES = True
    (l0, g0, p0, t0), (l1, g1, p1, t1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-6, (k, l0[k], l1[k])
    gs = float(g1.abs().max())
    assert (g0 - g1).abs().max().item() <= 1e-5 * gs
    assert (p0 - p1).abs().max().item() <= 1e-6 and (t0 - t1).abs().max().item() <= 1e-6


@pytest.mark.parametrize("M,N,K", [(6272, 384, 96), (6272, 96, 384), (200, 96, 48), (98, 3072, 768), (98, 768, 3072),
                                   (33000, 384, 96), (65570, 96, 96), (20000, 768, 192), (16400, 192, 768), (777, 200, 104),
                                   (1176, 1536, 768), (9408, 1536, 384), (2352, 3072, 768)])
def test_gemm_with_presplit_weights_is_bit_identical(M, N, K):
    """mis_gemm_split_b + mis_gemm_nt_split: the B operand (an nn.Linear weight) cut into its bf16 pieces once instead of per
    tile and k-step.  Same pieces, same products, same order: bit-identical to the bf16x3 kernels that split B themselves, for the
    plain form (with bias, with accumulation), the three fused epilogues, ragged K (zero-padded planes), N off the tile
    widths (64 x 96 and 128 x 128 tiles) and the split-K shapes; the batched split writes the same planes as the single one."""
    import ctypes
    tops = _t()
    prev = tops.set_split_precision(7)
    try:
        A, W, bias = _rand(M, K, seed=41).cuda(), (_rand(N, K, seed=42) * K ** -0.5).cuda(), _rand(N, seed=43).cuda()
        b3 = tops.SplitB(W).refresh()
        b3b = tops.SplitB(W)
        other = tops.SplitB(_rand(40, 64, seed=44).cuda())
        tops.SplitBatch([other, b3b]).run()
        assert torch.equal(b3.t, b3b.t)
        ref, got = torch.empty(M, N, device="cuda"), torch.full((M, N), float("nan"), device="cuda")
        from mis_hip import lib as _l
        # the short-contraction kernel sums K in steps of 16 (v_mfma_f32_16x16x16_bf16), the pre-split form always in steps of
        # 32: same pieces and products, another association of the fp32 sums -- equal to rounding there, bit for bit elsewhere
        short = "short" in tops._nt_name(_l.load(), M, N, K, 0)
        # ... and the two forms choose their k slices separately (gemm.hip nt_choice / b3_choice)
        short = short or _l.load().mis_gemm_workspace_bytes(M, N, K, 0) != _l.load().mis_gemm_nt_split_workspace_bytes(M, N, K)

        def same(a, b):
            return torch.allclose(a, b, rtol=2e-6, atol=2e-6) if short else torch.equal(a, b)
        tops.gemm(A, W, ref, bias=bias)
        name = ctypes.create_string_buffer(96)
        covered = _l.load().mis_gemm_nt_split_kernel_name(M, N, K, 0, name, 96) == 0
        assert covered == (K > 96), name.value
        if not covered:  # short contraction: left to the short-contraction kernel, the fp32-B entry points serve it
            assert not tops._nt_split(A, b3, got, bias=bias)
            tops.gemm(A, W, got, bias=bias, b3=b3)
            assert torch.equal(ref, got)
            return
        assert tops._nt_split(A, b3, got, bias=bias)
        assert same(ref, got)
        want = A.double() @ W.double().t() + bias.double()
        _close(got, want)
        tops.gemm(A, W, ref, accumulate=True)
        assert tops._nt_split(A, b3, got, accumulate=True)
        assert same(ref, got)
        if N % 4 == 0:
            E1 = _rand(M, N, seed=45).cuda()
            sc = (1 + _rand(4, seed=46).abs()).cuda()
            rps = (M + 3) // 4
            for ep, kw in ((tops.EP_GELU_FWD, {}), (tops.EP_GELU_BWD, {"E1": E1}),
                           (tops.EP_RESIDUAL, {"E1": E1, "rowscale": sc, "rows_per_scale": rps})):
                r2, g2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
                if ep == tops.EP_GELU_FWD:
                    kw = {"C2": r2}
                    kwg = {"C2": g2}
                else:
                    kwg = kw
                ok = tops.gemm_ex(A, W, ref, ep, bias=bias, **kw)
                okg = tops._nt_split(A, b3, got, bias=bias, epilogue=ep, **kwg)
                assert ok == okg
                if ok:
                    assert same(ref, got)
                    if ep == tops.EP_GELU_FWD:
                        assert same(r2, g2)
    finally:
        tops.set_split_precision(prev)


@pytest.mark.parametrize("B,H,K,P,c", [(2, 56, 96, 4, 96), (24, 28, 192, 2, 96), (3, 14, 384, 2, 192)])
def test_gemm_expand_with_presplit_weights_is_bit_identical(B, H, K, P, c):
    tops = _t()
    prev = tops.set_split_precision(7)
    try:
        M, N = B * H * H, P * P * c
        x, w = _rand(M, K, seed=31).cuda(), _rand(N, K, seed=32).cuda()
        ref = torch.empty(M * P * P, c, device="cuda")
        got = torch.full((M * P * P, c), float("nan"), device="cuda")
        assert tops.gemm_expand(x, w, ref, B, H, H, P, c)
        assert tops.gemm_expand(x, w, got, B, H, H, P, c, b3=tops.SplitB(w).refresh())
        assert torch.equal(ref, got)
        if K > 96:
            got.fill_(float("nan"))
            assert tops._nt_split(x, tops.SplitB(w).refresh(), got, ex=(H, H, P, c))
            assert torch.equal(ref, got)
    finally:
        tops.set_split_precision(prev)


def test_swin_step_with_presplit_weights_equals_split_per_tile():
    """A Mean-Teacher step of SwinUnet with the Linear weights pre-split once per pass (mis_gemm_split_batch +
    mis_gemm_nt_split) against the kernels that split B per tile: the same products; tile shapes and k slices may differ, so
    equal to the rounding of the fp32 sums."""
    from config import lite_config
    from mis_hip import tops
    from mis_hip.step import MeanTeacherTrainer
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.swin import OracleSwinUnet
    sd0 = filler.fill_state_dict(OracleSwinUnet(4).new_state())
    vol = filler.image((4, 1, 224, 224), "volume").cuda()
    lab = filler.labels((4, 224, 224), 4, torch.uint8).cuda()
    res = []
    prev = tops.set_split_precision(7)
    try:
        for on in (True, False):
            tops.PRESPLIT = on
            m, e = SwinUnet(lite_config(), num_classes=4), SwinUnet(lite_config(), num_classes=4)
            m.load_state_dict(sd0); e.load_state_dict(sd0)
            tr = MeanTeacherTrainer(m, e, labeled_bs=2, num_classes=4, cons_start_iter=0, seed=11, iter_num=1500)
            for _ in range(2):
                tr.step(vol, lab)
            torch.cuda.synchronize()
            used = any(getattr(op, "b3", None) is not None for op in m._last[0].ops)
            assert used == on
            res.append((tr.losses(), m.flat_grad.clone(), m.flat_param.clone(), e.flat_param.clone()))
    finally:
        tops.PRESPLIT = True
        tops.set_split_precision(prev)
    (l0, g0, p0, t0), (l1, g1, p1, t1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 2e-6 * max(1.0, abs(l1[k])), (k, l0[k], l1[k])
    gs = float(g1.abs().max())
    assert (g0 - g1).abs().max().item() <= 2e-5 * gs
    assert (p0 - p1).abs().max().item() <= 1e-6 and (t0 - t1).abs().max().item() <= 1e-6


@pytest.mark.parametrize("M,C,B", [(784, 96, 4), (3137, 384, 1), (50, 1536, 2), (6272, 192, 8)])
def test_layernorm_backward_with_the_residual_backward_in_one_pass(M, C, B):
    """mis_layernorm_bwd_residual_parts against mis_layernorm_bwd (accumulating into the input's gradient) followed by
    mis_residual_droppath's backward: bit-identical, with and without a gradient already in the input, with and without
    DropPath scales, accumulating into the shortcut's gradient or not."""
    tops = _t()
    x = (_rand(M, C, seed=51, scale=2.0) + 0.5).cuda()
    g, b = (1 + 0.2 * _rand(C, seed=52)).cuda(), (0.1 * _rand(C, seed=53)).cuda()
    dy = _rand(M, C, seed=54).cuda()
    y = torch.empty(M, C, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    tops.layernorm_fwd(x, y, g, b, mean, rstd)
    rps = (M + B - 1) // B
    for has_gin in (False, True):
        for scales in (None, (torch.tensor([0.0, 1.25, 1.0, 2.5, 0.0, 1.0, 1.25, 1.25][:B])).cuda()):
            for acc in (False, True):
                gin = _rand(M, C, seed=55).cuda() if has_gin else None
                base = _rand(M, C, seed=56).cuda()
                # two passes
                dxin = gin.clone() if has_gin else torch.empty(M, C, device="cuda")
                dg1, db1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
                tops.layernorm_bwd(x, dy, dxin, g, mean, rstd, dg1, db1, accumulate_dx=has_gin)
                ds1, dbr1 = base.clone(), torch.empty(M, C, device="cuda")
                tops.residual_bwd(dxin, ds1, dbr1, rps, drop_p=0.5 if scales is not None else 0.0, scale_override=scales,
                                  accumulate_shortcut=acc)
                # one pass
                ws = tops.colreduce_workspace(M, C)
                ds2, dbr2 = base.clone(), torch.full((M, C), float("nan"), device="cuda")
                assert tops.layernorm_bwd_residual_parts(x, dy, gin, ds2, dbr2, g, mean, rstd, ws, rowscale=scales,
                                                         rows_per_scale=rps, accumulate_shortcut=acc)
                dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
                tops.layernorm_bwd_final(ws, M, C, dg2, db2)
                assert torch.equal(ds1, ds2) and torch.equal(dbr1, dbr2), (has_gin, scales is not None, acc)
                assert torch.equal(dg1, dg2) and torch.equal(db1, db2)


@pytest.mark.parametrize("B,H,W,NC,keep", [(2, 56, 56, 4, True), (1, 20, 12, 2, True), (3, 14, 14, 3, False)])
@pytest.mark.parametrize("prec2", [0, 7], ids=["fp32mfma", "bf16x3"])
def test_final_expand_layernorm_and_head_in_the_gemm_epilogue(B, H, W, NC, keep, prec2):
    """mis_gemm_expand_ln_head against mis_gemm_expand + mis_ln_head_fwd: shuffled tokens, mean / rstd and logits bit for bit
    (same arithmetic term for term); ``out`` NULL keeps only what the loss needs."""
    tops = _t()
    prev = tops.set_split_precision(prec2)
    try:
        P, c, K = 4, 96, 96
        M = B * H * W
        x, w = _rand(M, K, seed=61).cuda(), (_rand(P * P * c, K, seed=62) * 0.1).cuda()
        g, b = (1 + 0.2 * _rand(c, seed=63)).cuda(), (0.1 * _rand(c, seed=64)).cuda()
        hw = _rand(NC, c, seed=65, scale=0.3).cuda()
        T = M * P * P
        sh0 = torch.empty(T, c, device="cuda")
        assert tops.gemm_expand(x, w, sh0, B, H, W, P, c)
        m0, r0 = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
        lg0 = torch.empty(B, NC, 1, H * P, W * P, device="cuda")
        assert tops.ln_head_fwd(sh0, g, b, hw, m0, r0, lg0)
        sh1 = torch.full((T, c), float("nan"), device="cuda") if keep else None
        m1, r1 = torch.full((T,), float("nan"), device="cuda"), torch.full((T,), float("nan"), device="cuda")
        lg1 = torch.full((B, NC, 1, H * P, W * P), float("nan"), device="cuda")
        assert tops.gemm_expand_ln_head(x, w, sh1, B, H, W, P, c, g, b, hw, m1, r1, lg1)
        if keep:
            assert torch.equal(sh0, sh1)
        assert torch.equal(m0, m1) and torch.equal(r0, r1)
        assert torch.equal(lg0, lg1)
    finally:
        tops.set_split_precision(prev)


def test_colsum_batch_jobs_are_independent_column_sums():
    """mis_colsum_batch: jobs of both partial formats (float / float2), strides wider than C, ragged widths, accumulate, in one
    launch; every job's result is the fixed-order double sum of its own partial rows whatever else is in the batch."""
    tops = _t()
    g = torch.Generator().manual_seed(5)
    specs = [(7, 96, 96, 0, False), (300, 100, 128, 1, False), (1, 4, 4, 0, True), (1882, 96, 96, 1, True),
             (85, 27648, 27648, 2, False), (33, 507, 507, 0, False), (3, 288, 288, 2, True)]
    jobs, want, outs = [], [], []
    for slabs, C, stride, pairs, acc in specs:
        part = (torch.rand(slabs, stride, 2 if pairs == 1 else 1, generator=g) - 0.5).cuda()
        oa = torch.full((C,), 0.25, device="cuda")
        ob = torch.full((C,), -0.5, device="cuda") if pairs == 1 else None
        jobs.append(tops.ColsumJob(part, 0, stride, slabs, C, pairs, oa, ob, accumulate=acc))
        s = part.double().sum(0)[:C]
        want.append((s[:, 0] + (0.25 if acc else 0.0), s[:, 1] + (-0.5 if acc else 0.0) if pairs == 1 else None))
        outs.append((oa, ob))
    tops.ColsumBatch(jobs).run()
    for (oa, ob), (wa, wb) in zip(outs, want):
        assert (oa.double() - wa).abs().max().item() <= 4e-6 * max(1.0, wa.abs().max().item())
        if ob is not None:
            assert (ob.double() - wb).abs().max().item() <= 1e-6 * max(1.0, wb.abs().max().item())
    # the same jobs in another order / another batch: bit-identical results
    first = [o[0].clone() for o in outs]
    for (oa, ob), (_, _, _, _, acc) in zip(outs, specs):
        oa.fill_(0.25)
        if ob is not None:
            ob.fill_(-0.5)
    tops.ColsumBatch(jobs[::-1]).run()
    tops_again = [o[0] for o in outs]
    assert all(torch.equal(a, b) for a, b in zip(first, tops_again))


@pytest.mark.parametrize("T,Cout,Cin,bias", [(20000, 96, 288, True), (150528, 288, 96, True), (2352, 768, 3072, True),
                                             (49, 1536, 768, True), (3137, 100, 36, True), (9408, 384, 1536, False)])
def test_gemm_dw_parts_plus_batched_sums_equal_gemm_dw(T, Cout, Cin, bias, prec):
    """mis_gemm_dw_parts leaves the k-slices' partials in the caller's workspace; two mis_colsum_batch jobs finish dW / db
    (the token plans' batched form of nn.Linear's parameter gradients).  Bit-identical to mis_gemm_dw."""
    tops = _t()
    X, dY = _rand(T, Cin, seed=15).cuda(), (_rand(T, Cout, seed=16) + 0.25).cuda()
    dW = torch.full((Cout, Cin), float("nan"), device="cuda")
    db = torch.full((Cout,), float("nan"), device="cuda") if bias else None
    ws = tops.gemm_dw_workspace(Cout, Cin, T)
    slices = tops.gemm_dw_parts(dY, X, dW, db, ws)
    if slices:
        assert torch.isnan(dW).all()                          # untouched until the sums run
        jobs = [tops.ColsumJob(ws, 0, Cout * Cin, slices, Cout * Cin, 2, dW.view(-1))]
        if bias:
            jobs.append(tops.ColsumJob(ws, slices * Cout * Cin, Cout, slices, Cout, 2, db))
        tops.ColsumBatch(jobs).run()
    else:
        assert ws is None or T < 4096
    _close(dW, dY.double().t() @ X.double(), rtol=3e-4)
    ref = torch.empty_like(dW)
    if bias:
        refb = torch.empty_like(db)
        tops.gemm_dw(dY, X, ref, refb)
        assert torch.equal(db, refb)
    else:
        tops.gemm(dY, X, ref, trans=True)
    assert torch.equal(dW, ref)           # pairs = 2: the GEMM's own summation order


def test_window_attention_table_gradient_as_a_batched_sum(prec):
    """The MFMA backward reduces each (sample, window, head) unit's dS to the (2 ws - 1)^2 table entries in the kernel and
    leaves [units / nH][entries][nH] partial rows: their column sum -- as a mis_colsum_batch job -- is the table gradient the
    one-call form returns."""
    tops = _t()
    for B, H, W, nH, shift, ws in ((3, 14, 21, 2, 3, 7), (5, 16, 24, 3, 4, 8)):
        C = nH * 32
        qkv = (_rand(B * H * W, 3 * C, seed=22, scale=1.5)).cuda()
        table = (_rand((2 * ws - 1) ** 2, nH, seed=23, scale=0.5)).cuda()
        dout = _rand(B * H * W, C, seed=24).cuda()
        scale = 32 ** -0.5
        dqkv, dt = torch.empty(B * H * W, 3 * C, device="cuda"), torch.empty_like(table)
        tops.window_attention_bwd(qkv, dout, dqkv, table, dt, B, H, W, nH, shift, scale, window=ws)
        own = tops.window_attention_workspace(B, H, W, nH, window=ws)
        dqkv2, dt2 = torch.empty_like(dqkv), torch.full_like(dt, float("nan"))
        tops.window_attention_bwd_parts(qkv, dout, dqkv2, table, own, B, H, W, nH, shift, scale, window=ws)
        rows, cols = tops.window_attention_table_partials(B, H, W, nH, window=ws)
        assert (rows, cols) == (B * (H // ws) * (W // ws), (2 * ws - 1) ** 2 * nH)
        tops.ColsumBatch([tops.ColsumJob(own.view(torch.float32), 0, cols, rows, cols, False, dt2.view(-1))]).run()
        assert torch.equal(dqkv, dqkv2) and torch.equal(dt, dt2)


def test_swin_step_with_batched_finishing_sums_equals_per_op_launches():
    """plan.BATCH_FINALS: the LayerNorm affine, Linear weight / bias and bias-table gradients of a SwinUnet Mean-Teacher step
    finished by a few mis_colsum_batch launches instead of ~130 per-op launches -- the same step bit for bit."""
    from config import lite_config
    from mis_hip import plan
    from mis_hip.step import MeanTeacherTrainer
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.swin import OracleSwinUnet
    sd0 = filler.fill_state_dict(OracleSwinUnet(4).new_state())
    vol = filler.image((4, 1, 224, 224), "volume").cuda()
    lab = filler.labels((4, 224, 224), 4, torch.uint8).cuda()
    res = []
    for batch, flush_mb in ((True, 32), (False, 32), (True, 1)):
        plan.BATCH_FINALS, plan.FINALS_FLUSH_BYTES = batch, flush_mb << 20
        try:
            m, e = SwinUnet(lite_config(), num_classes=4), SwinUnet(lite_config(), num_classes=4)
            m.load_state_dict(sd0); e.load_state_dict(sd0)
            tr = MeanTeacherTrainer(m, e, labeled_bs=2, num_classes=4, cons_start_iter=0, seed=11, iter_num=1500)
            for _ in range(2):
                tr.step(vol, lab)
            torch.cuda.synchronize()
            res.append((tr.losses(), m.flat_grad.clone(), m.flat_param.clone(), e.flat_param.clone()))
        finally:
            plan.BATCH_FINALS, plan.FINALS_FLUSH_BYTES = True, 32 << 20
    (l0, g0, p0, t0), (l1, g1, p1, t1), (l2, g2, p2, t2) = res
    assert l0 == l1
    # the batch kernel sums every job in its per-op launch's own order: not a bit changes, however the jobs are grouped
    assert torch.equal(g0, g1) and torch.equal(p0, p1) and torch.equal(t0, t1)
    assert torch.equal(g0, g2) and torch.equal(p0, p2) and torch.equal(t0, t2)


@pytest.mark.parametrize("M,N,K", [(150528, 288, 96), (150528, 96, 384), (37632, 576, 192), (37632, 768, 192), (165570, 96, 96),
                                   (70001, 192, 100), (9408 * 12, 384, 1152)])
def test_register_a_nt_gemm(M, N, K):
    """gemm_nt_rega_kernel (round 6): the A operand goes from HBM straight into the v_mfma_f32_16x16x32_bf16 operand registers,
    the pre-split weight planes (NATURAL element order, SplitB(rows=M)) through LDS.  Plain / bias / accumulate, the three fused
    epilogues, a ragged last row tile, K not a multiple of 32, strided C and E1 -- against float64; agreement with the staged
    kernel to fp32 rounding (other element order inside a K = 32 block: not bit for bit); deterministic."""
    tops = _t()
    prev = tops.set_split_precision(7)
    try:
        A, W = _rand(M, K, seed=41), _rand(N, K, seed=42, scale=K ** -0.5)
        bias = _rand(N, seed=43)
        Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
        b3 = tops.SplitB(Wd, rows=M).refresh()
        assert b3.natural, "the register-A kernel serves this shape"
        idx = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])         # first and last (ragged) row tiles
        ref = A[idx].double() @ W.double().t()
        C = torch.full((M, N), float("nan"), device="cuda")
        tops.gemm(Ad, Wd, C, bias=bd, b3=b3)
        _close(C[idx.cuda()], ref + bias.double())
        assert torch.isfinite(C).all()
        C2 = torch.empty_like(C)
        tops.gemm(Ad, Wd, C2, bias=bd, b3=b3)
        assert torch.equal(C, C2)
        staged = torch.empty_like(C)
        tops.gemm(Ad, Wd, staged, bias=bd, b3=tops.SplitB(Wd).refresh())
        assert (C - staged).abs().max().item() <= 2e-6 * max(1.0, staged.abs().max().item())
        tops.gemm(Ad, Wd, C, accumulate=True, b3=b3)
        _close(C[idx.cuda()], 2 * ref + bias.double())
        wide = torch.zeros(M, N + 32, device="cuda")
        tops.gemm(Ad, Wd, wide[:, 32:], b3=b3)
        _close(wide[idx.cuda(), 32:], ref)
        assert wide[:, :32].abs().max().item() == 0
        del wide, staged, C2
        v = ref + bias.double()
        Cg = torch.empty(M, N, device="cuda")
        assert tops.gemm_ex(Ad, Wd, C, tops.EP_GELU_FWD, bias=bd, C2=Cg, b3=b3)
        _close(C[idx.cuda()], v)
        _close(Cg[idx.cuda()], F.gelu(v))
        assert tops.gemm_ex(Ad, Wd, None, tops.EP_GELU_FWD, bias=bd, C2=Cg, b3=b3)          # a forward nobody differentiates
        _close(Cg[idx.cuda()], F.gelu(v))
        h = _rand(M, N, seed=44, scale=1.5)
        hd = h[idx].double().requires_grad_(True)
        F.gelu(hd).backward(torch.ones(idx.numel(), N, dtype=torch.float64))
        assert tops.gemm_ex(Ad, Wd, C, tops.EP_GELU_BWD, E1=h.cuda(), b3=b3)
        _close(C[idx.cuda()], ref * hd.grad)
        rps = M // 2 if M % 2 == 0 else M
        wide = _rand(M, N + 32, seed=45)
        sc = torch.tensor([0.0, 1.25] if rps < M else [1.25])
        assert tops.gemm_ex(Ad, Wd, C, tops.EP_RESIDUAL, bias=bd, E1=wide.cuda()[:, 16:16 + N], rowscale=sc.cuda(),
                            rows_per_scale=rps, b3=b3)
        _close(C[idx.cuda()], wide[idx, 16:16 + N].double() + sc.repeat_interleave(rps)[idx, None].double() * v)
    finally:
        tops.set_split_precision(prev)


@pytest.mark.parametrize("M,K", [(150528, 96), (150528, 384), (75264, 96), (65570, 288)])
def test_residual_and_layernorm_in_the_register_a_epilogue(M, K):
    """mis_gemm_nt_residual_ln: X = E1 + rowscale * (A W^T + b) and Y = LayerNorm(X) g + be, mean / rstd, in one launch (the
    persistent resident-panel kernel for K = 96, the streamed one above) against float64; ragged last slab; strided E1 / X."""
    tops = _t()
    prev = tops.set_split_precision(7)
    try:
        N = 96
        A, W, bias = _rand(M, K, seed=51), _rand(N, K, seed=52, scale=K ** -0.5), _rand(N, seed=53)
        gam, bet = _rand(N, seed=54) + 1.5, _rand(N, seed=55)
        wide = _rand(M, N + 32, seed=56)
        rps = M // 2 if M % 2 == 0 else M
        sc = torch.tensor([0.5, 1.25] if rps < M else [1.25])
        b3 = tops.SplitB(W.cuda(), rows=M).refresh()
        if not b3.natural:
            pytest.skip("the register-A kernels do not serve this shape")
        Xw = torch.zeros(M, N + 16, device="cuda")
        Y = torch.empty(M, N, device="cuda")
        mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        assert tops.gemm_residual_ln(A.cuda(), b3, Xw[:, 16:], wide.cuda()[:, 16:16 + N], gam.cuda(), bet.cuda(), Y, mean, rstd,
                                     bias=bias.cuda(), rowscale=sc.cuda(), rows_per_scale=rps)
        idx = torch.cat([torch.arange(0, 200), torch.arange(M // 2 - 100, M // 2 + 100), torch.arange(M - 200, M)])
        v = A[idx].double() @ W.double().t() + bias.double()
        x = wide[idx, 16:16 + N].double() + sc.repeat_interleave(rps)[idx, None].double() * v
        _close(Xw[idx.cuda(), 16:], x)
        assert Xw[:, :16].abs().max().item() == 0
        mu = x.mean(1)
        rs = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
        assert (mean[idx.cuda()].cpu().double() - mu).abs().max().item() <= 1e-5 * max(1.0, mu.abs().max().item())
        assert ((rstd[idx.cuda()].cpu().double() - rs) / rs).abs().max().item() <= 1e-4
        _close(Y[idx.cuda()], (x - mu[:, None]) * rs[:, None] * gam.double() + bet.double(), rtol=3e-4, atol=2e-5)
        # the separate LayerNorm on the same X: the same normalisation to fp32 rounding
        Y2, m2, r2 = torch.empty_like(Y), torch.empty_like(mean), torch.empty_like(rstd)
        tops.layernorm_fwd(Xw[:, 16:], Y2, gam.cuda(), bet.cuda(), m2, r2)
        assert (Y - Y2).abs().max().item() <= 2e-5 * max(1.0, Y2.abs().max().item())
    finally:
        tops.set_split_precision(prev)


@pytest.mark.parametrize("B,H,W,NC,keep", [(24, 56, 56, 4, True), (12, 56, 56, 3, False), (11, 60, 52, 2, True)])
def test_final_expand_layernorm_and_head_on_the_register_a_kernel(B, H, W, NC, keep):
    """mis_gemm_expand_ln_head_split: the persistent resident-panel register-A kernel with LayerNorm + output head on the
    accumulators (a lane holds 24 of a shuffled token's 96 values; cross-group sums) against mis_gemm_expand + mis_ln_head_fwd:
    shuffled tokens, mean / rstd, logits to fp32 rounding (other element order inside a K = 32 block); ragged last slab."""
    tops = _t()
    prev = tops.set_split_precision(7)
    try:
        P, c, K = 4, 96, 96
        M = B * H * W
        x, w = _rand(M, K, seed=61).cuda(), (_rand(P * P * c, K, seed=62) * 0.1).cuda()
        g, b = (1 + 0.2 * _rand(c, seed=63)).cuda(), (0.1 * _rand(c, seed=64)).cuda()
        hw = _rand(NC, c, seed=65, scale=0.3).cuda()
        b3 = tops.SplitB(w, rows=M).refresh()
        assert b3.natural
        T = M * P * P
        sh0 = torch.empty(T, c, device="cuda")
        assert tops.gemm_expand(x, w, sh0, B, H, W, P, c)
        m0, r0 = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
        lg0 = torch.empty(B, NC, 1, H * P, W * P, device="cuda")
        assert tops.ln_head_fwd(sh0, g, b, hw, m0, r0, lg0)
        sh1 = torch.full((T, c), float("nan"), device="cuda") if keep else None
        m1, r1 = torch.full((T,), float("nan"), device="cuda"), torch.full((T,), float("nan"), device="cuda")
        lg1 = torch.full((B, NC, 1, H * P, W * P), float("nan"), device="cuda")
        from mis_hip import ops
        ops.PROFILE = []
        try:
            assert tops.gemm_expand_ln_head(x, w, sh1, B, H, W, P, c, g, b, hw, m1, r1, lg1, b3=b3)
            assert ops.PROFILE[-1][0] == "gemm_nt_rega_res_kernel<4>"
        finally:
            ops.PROFILE = None
        if keep:
            assert (sh0 - sh1).abs().max().item() <= 2e-6 * max(1.0, sh0.abs().max().item())
        assert (m0 - m1).abs().max().item() <= 1e-6 and ((r0 - r1) / r0).abs().max().item() <= 1e-4
        assert (lg0 - lg1).abs().max().item() <= 1e-4 * max(1.0, lg0.abs().max().item())
        lg2 = torch.full_like(lg1, float("nan"))
        assert tops.gemm_expand_ln_head(x, w, sh1, B, H, W, P, c, g, b, hw, m1, r1, lg2, b3=b3)
        assert torch.equal(lg1, lg2)
    finally:
        tops.set_split_precision(prev)

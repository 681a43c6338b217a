import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo/cv-ssl-mis_amd")
from mis_hip import lib as _l, ops
L = _l.load()
torch.manual_seed(0)
for (N, Cin, Cout, S, var) in [(4,128,128,8,2),(2,128,128,8,2),(8,256,256,6,2),(4,128,128,12,3)]:
    for kind in ("randn", "sparse"):
        x = torch.randn(N, Cin, S, S, S, dtype=torch.float64)
        if kind == "sparse":   # half the channels zero, the others doubled + a relu'd distribution with a large mean
            m = (torch.rand(1, Cin, 1, 1, 1) > 0.5).double() * 2
            x = (x.abs() + 1.0) * m
        w = torch.randn(Cout, Cin, 3, 3, 3, dtype=torch.float64) * (Cin * 27) ** -0.5
        b = torch.randn(Cout, dtype=torch.float64) * 0.1
        ref = F.conv3d(x, w, b, padding=1)
        xd, wd, bd = x.float().cuda(), w.float().cuda(), b.float().cuda()
        wt = ops.conv_pack(wd, 4)
        Sv = S ** 3
        T = L.mis_conv3d_wino_stat_tiles(S, S, S, var)
        res = {}
        for name in ("unsplit", "split"):
            y = torch.empty(N, Cout, S, S, S, device="cuda")
            st = torch.zeros(Cout * N * T, 2, device="cuda")
            if name == "unsplit":
                _l.check(L.mis_conv3d_wino_fwd(_l.ptr(xd), Cin * Sv, _l.ptr(wt), _l.ptr(bd), _l.ptr(y), Cout * Sv, N, Cin, Cout, S, S, S,
                                               _l.ptr(st), T, Cout * T, var, _l.stream_ptr()), "fwd")
            else:
                nb = L.mis_conv3d_wino_fwd_workspace_bytes(N, Cin, Cout, S, S, S, var)
                ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
                _l.check(L.mis_conv3d_wino_fwd_ws(_l.ptr(xd), Cin * Sv, _l.ptr(wt), _l.ptr(bd), _l.ptr(y), Cout * Sv, N, Cin, Cout, S, S, S,
                                                  _l.ptr(st), T, Cout * T, var, _l.ptr(ws), nb, _l.stream_ptr()), "fwd_ws")
            e = (y.cpu().double() - ref)
            s = st.view(N, Cout, T, 2).sum(2).cpu().double()      # stat layout here: [n][c][tile]
            s1r, s2r = ref.sum((2, 3, 4)), (ref * ref).sum((2, 3, 4))
            res[name] = (e.abs().max().item(), e.pow(2).mean().sqrt().item(), ((s[..., 0] - s1r).abs() / s2r.sqrt()).max().item(),
                         ((s[..., 1] - s2r).abs() / s2r).max().item())
        # the direct kernel
        yd = torch.empty(N, Cout, S, S, S, device="cuda")
        ops.conv_fwd(xd, ops.conv_pack(wd, 0), bd, yd, Cin, Cout, (3, 3, 3))
        ed = (yd.cpu().double() - ref)
        ks = L.mis_conv3d_wino_fwd_splits(N, Cin, Cout, S, S, S, var)
        print(f"N{N} {Cin}->{Cout} {S}^3 {kind:6s} ks={ks} |ref|max {ref.abs().max():.2f}: " +
              "  ".join(f"{k}: max {v[0]:.2e} rms {v[1]:.2e} s1 {v[2]:.1e} s2 {v[3]:.1e}" for k, v in res.items()) +
              f"  direct: max {ed.abs().max():.2e} rms {ed.pow(2).mean().sqrt():.2e}")
print("data gradient (pack mode 5):")
for (N, Cin, Cout, S, var) in [(4,128,128,8,2),(2,128,128,8,2),(8,256,256,6,2)]:
    dy = torch.randn(N, Cout, S, S, S, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, 3, dtype=torch.float64) * (Cin * 27) ** -0.5
    ref = F.conv3d(dy, w.transpose(0, 1).flip(2, 3, 4), padding=1)
    dyd, wd = dy.float().cuda(), w.float().cuda()
    wt = ops.conv_pack(wd, 5)
    Sv = S ** 3
    out = {}
    for name in ("unsplit", "split"):
        dx = torch.empty(N, Cin, S, S, S, device="cuda")
        if name == "unsplit":
            _l.check(L.mis_conv3d_wino_fwd(_l.ptr(dyd), Cout * Sv, _l.ptr(wt), None, _l.ptr(dx), Cin * Sv, N, Cout, Cin, S, S, S,
                                           None, 0, 0, var, _l.stream_ptr()), "fwd")
        else:
            nb = L.mis_conv3d_wino_fwd_workspace_bytes(N, Cout, Cin, S, S, S, var)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
            _l.check(L.mis_conv3d_wino_fwd_ws(_l.ptr(dyd), Cout * Sv, _l.ptr(wt), None, _l.ptr(dx), Cin * Sv, N, Cout, Cin, S, S, S,
                                              None, 0, 0, var, _l.ptr(ws), nb, _l.stream_ptr()), "fwd_ws")
        e = dx.cpu().double() - ref
        out[name] = (e.abs().max().item(), e.pow(2).mean().sqrt().item())
    print(f"N{N} {Cout}->{Cin} {S}^3: " + "  ".join(f"{k}: max {v[0]:.2e} rms {v[1]:.2e}" for k, v in out.items()))

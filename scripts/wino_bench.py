"""Development harness: Winograd F(2^3, 3^3) conv (mis_conv3d_wino_fwd) vs the direct kernel and torch, on the GPU.
    python scripts/wino_bench.py            correctness at small shapes + timing at the config-3 layer shapes"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import lib as _l, ops  # noqa: E402

L = _l.load()
c_p, c_i, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong


def wino(x, w, bias, variant, stat=False):
    N, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    wt = ops.conv_pack(w, 4)
    y = torch.empty(N, Cout, D, H, W, device="cuda")
    S = D * H * W
    tiles = L.mis_conv3d_wino_stat_tiles(D, H, W, variant)
    st = torch.zeros(Cout, N, tiles, 2, device="cuda") if stat else None

    def run():
        _l.check(L.mis_conv3d_wino_fwd(_l.ptr(x), Cin * S, _l.ptr(wt), _l.ptr(bias), _l.ptr(y), Cout * S, N, Cin, Cout,
                                       D, H, W, _l.ptr(st) if stat else None, N * tiles if stat else 0, tiles if stat else 0,
                                       variant, _l.stream_ptr()), "wino")
    run()
    return y, run


def direct(x, w, bias):
    N, Cin, D, H, W = x.shape
    Cout = w.shape[0]
    wp = ops.conv_pack(w, 0)
    y = torch.empty(N, Cout, D, H, W, device="cuda")

    def run():
        ops.conv_fwd(x, wp, bias, y, Cin, Cout, (3, 3, 3))
    run()
    return y, run


def timeit(fn, n=30):
    for _ in range(40):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def wgrad_main():
    """Winograd weight gradient vs the direct kernel and torch."""
    torch.manual_seed(0)

    def wino_wg(x, dy, var):
        N, Cin, D, H, W = x.shape
        Cout = dy.shape[1]
        S = D * H * W
        nb = L.mis_conv3d_wino_wgrad_workspace_bytes(N, Cin, Cout, D, H, W, var)
        ws = torch.empty(nb // 4, device="cuda")
        dw = torch.zeros(Cout, Cin, 3, 3, 3, device="cuda")

        def run():
            _l.check(L.mis_conv3d_wino_wgrad(_l.ptr(x), Cin * S, _l.ptr(dy), Cout * S, _l.ptr(dw), _l.ptr(ws), nb, N, Cin,
                                             Cout, D, H, W, 0, var, _l.stream_ptr()), "wino wgrad")
        run()
        return dw, run

    def direct_wg(x, dy):
        dw = torch.zeros(dy.shape[1], x.shape[1], 3, 3, 3, device="cuda")

        def run():
            keep, ops.WINO = ops.WINO, 0
            ops.conv_wgrad(x, dy, dw, (3, 3, 3))
            ops.WINO = keep
        run()
        return dw, run

    for (N, Cin, Cout, D, H, W, var) in [(1, 16, 16, 2, 4, 32, 0), (2, 16, 16, 4, 8, 64, 0), (1, 24, 40, 6, 4, 32, 0),
                                         (2, 16, 32, 4, 8, 16, 1), (1, 48, 16, 8, 4, 48, 1), (3, 16, 16, 6, 12, 96, 0),
                                         (2, 32, 16, 8, 4, 8, 2), (1, 64, 64, 8, 12, 24, 2)]:
        x = torch.randn(N, Cin, D, H, W, device="cuda")
        dy = torch.randn(N, Cout, D, H, W, device="cuda")
        xr = x.double().requires_grad_(False)
        w = torch.zeros(Cout, Cin, 3, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv3d(xr, w, padding=1).backward(dy.double())
        ref = w.grad
        dww, _ = wino_wg(x, dy, var)
        dwd, _ = direct_wg(x, dy)
        sc = ref.abs().max().item()
        print(f"wgrad N{N} {Cin}->{Cout} {D}x{H}x{W} v{var}: wino err {(dww.double() - ref).abs().max().item() / sc:.3e}  "
              f"direct err {(dwd.double() - ref).abs().max().item() / sc:.3e} (rel. to max {sc:.1f})", flush=True)
    if "flat" in sys.argv[1:]:      # the flat form (variant 5) against the direct kernel on the deep levels of unet_3D / V-Net
        for (N, Cin, Cout, S) in [(8, 64, 128, 12), (8, 128, 128, 12), (8, 384, 128, 12), (8, 128, 256, 6), (8, 256, 256, 6),
                                  (4, 128, 256, 6), (8, 256, 128, 12)]:
            x = torch.randn(N, Cin, S, S, S, device="cuda")
            dy = torch.randn(N, Cout, S, S, S, device="cuda")
            fl = 2.0 * N * Cout * Cin * 27 * S ** 3
            da, ra = wino_wg(x, dy, 5)
            db, rb = direct_wg(x, dy)
            err = (da - db).abs().max().item() / db.abs().max().item()
            ta, tb = timeit(ra), timeit(rb)
            print(f"wgrad N{N} {Cin}->{Cout} {S}^3: flat {ta:8.1f} us ({fl / ta / 1e6:6.1f} TF eq)   direct {tb:8.1f} us ({fl / tb / 1e6:6.1f} TF)"
                  f"  x{tb / ta:.2f}  rel diff {err:.2e}", flush=True)
        return
    if "ab" in sys.argv[1:]:        # box kernels (0 / 1 / 2) against the z-ring kernels (3 / 4 / 6) on the config-3 / V-Net layers
        cases = [(8, 16, 16, 96, 0, 3), (8, 48, 16, 96, 0, 3), (8, 32, 32, 48, 1, 4), (8, 96, 32, 48, 1, 4), (8, 16, 32, 48, 1, 4),
                 (8, 32, 32, 48, 0, 4), (8, 64, 64, 24, 2, 6), (8, 192, 64, 24, 2, 6), (8, 32, 64, 24, 2, 6), (4, 64, 64, 24, 2, 6)]
        if "24" in sys.argv[1:]:
            cases = [c for c in cases if c[3] == 24]
        for (N, Cin, Cout, S, va, vb) in cases:
            x = torch.randn(N, Cin, S, S, S, device="cuda")
            dy = torch.randn(N, Cout, S, S, S, device="cuda")
            fl = 2.0 * N * Cout * Cin * 27 * S ** 3
            try:
                da, ra = wino_wg(x, dy, va)
            except RuntimeError as e:
                print("skip", N, Cin, Cout, S, va, e)
                continue
            db, rb = wino_wg(x, dy, vb)
            err = (da - db).abs().max().item() / da.abs().max().item()
            ta, tb = timeit(ra), timeit(rb)
            print(f"wgrad N{N} {Cin}->{Cout} {S}^3: v{va} {ta:8.1f} us ({fl / ta / 1e6:6.1f} TF eq, {fl / 3.375 / ta / 157.3e6:.3f} of the pipe)"
                  f"   v{vb} {tb:8.1f} us ({fl / tb / 1e6:6.1f} TF eq, {fl / 3.375 / tb / 157.3e6:.3f})  x{ta / tb:.3f}  rel diff {err:.2e}",
                  flush=True)
        return
    for (N, Cin, Cout, S, var) in [(8, 16, 16, 96, 0), (8, 48, 16, 96, 0), (8, 32, 32, 48, 1), (8, 96, 32, 48, 1),
                                   (8, 16, 32, 48, 1), (8, 64, 64, 24, 2), (8, 192, 64, 24, 2), (8, 32, 64, 24, 2)]:
        x = torch.randn(N, Cin, S, S, S, device="cuda")
        dy = torch.randn(N, Cout, S, S, S, device="cuda")
        fl = 2.0 * N * Cout * Cin * 27 * S ** 3
        dww, rw = wino_wg(x, dy, var)
        dwd, rd = direct_wg(x, dy)
        err = (dww - dwd).abs().max().item() / dwd.abs().max().item()
        tw, td = timeit(rw), timeit(rd)
        print(f"wgrad N{N} {Cin}->{Cout} {S}^3 v{var}: wino {tw:8.1f} us ({fl / tw / 1e6:6.1f} TF eq)  direct {td:8.1f} us "
              f"({fl / td / 1e6:6.1f} TF)  speedup {td / tw:.2f}  rel diff {err:.2e}", flush=True)


def main():
    torch.manual_seed(0)
    if os.environ.get("MIS_WINO_DBG"):
        for (N, Cin, Cout, S, var) in [(8, 16, 16, 96, 0), (8, 48, 16, 96, 0)]:
            x = torch.randn(N, Cin, S, S, S, device="cuda")
            w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
            b = torch.randn(Cout, device="cuda")
            yw, rw = wino(x, w, b, var)
            yw2, rw2 = wino(x, w, b, var, stat=True)
            line = f"dbg {os.environ['MIS_WINO_DBG']} N{N} {Cin}->{Cout} {S}^3: {timeit(rw):8.1f} us   with stats {timeit(rw2):8.1f} us"
            if Cin == Cout:      # the data gradient with the norm backward's partial sums in its epilogue
                xn = torch.randn(N, Cout, S, S, S, device="cuda")
                mean = xn.mean(dim=(2, 3, 4)).flatten().contiguous()
                T = L.mis_conv3d_wino_stat_tiles(S, S, S, var)
                part = torch.zeros(N * Cout * T, 2, device="cuda")
                da = torch.empty_like(xn)
                wpd = ops.conv_pack(w, 5)
                rn = lambda: ops.conv_dgrad_norm(x, wpd, da, Cin, Cout, xn, mean, 0.0, part, var)
                rn()
                line += f"   dgrad+norm partials {timeit(rn):8.1f} us"
            print(line, flush=True)
        return
    for (N, Cin, Cout, D, H, W, var) in [(1, 16, 16, 4, 4, 32, 0), (2, 16, 16, 8, 12, 64, 0), (1, 24, 32, 6, 10, 20, 0),
                                         (2, 16, 32, 8, 8, 16, 1), (1, 32, 32, 6, 6, 36, 1), (3, 48, 16, 10, 6, 40, 0),
                                         (1, 32, 32, 8, 8, 8, 2), (2, 64, 48, 8, 16, 24, 2), (1, 40, 16, 6, 10, 12 + 2, 2)]:
        x = torch.randn(N, Cin, D, H, W, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.1
        b = torch.randn(Cout, device="cuda")
        ref = torch.nn.functional.conv3d(x.double(), w.double(), b.double(), padding=1)
        yw, _ = wino(x, w, b, var)
        yd, _ = direct(x, w, b)
        ew = (yw.double() - ref).abs().max().item()
        ed = (yd.double() - ref).abs().max().item()
        print(f"shape N{N} {Cin}->{Cout} {D}x{H}x{W} v{var}: wino err {ew:.3e}  direct err {ed:.3e}  ref max {ref.abs().max().item():.2f}",
              flush=True)
    for (N, Cin, Cout, S, var) in [(8, 16, 16, 96, 0), (8, 48, 16, 96, 0), (8, 16, 48, 96, 0), (8, 32, 32, 48, 1),
                                   (8, 32, 32, 48, 0), (8, 96, 32, 48, 1), (8, 64, 64, 24, 2), (8, 192, 64, 24, 2), (4, 64, 64, 24, 2), (8, 64, 192, 24, 2)]:
        x = torch.randn(N, Cin, S, S, S, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
        b = torch.randn(Cout, device="cuda")
        fl = 2.0 * N * Cout * Cin * 27 * S ** 3
        try:
            yw, rw = wino(x, w, b, var)
        except RuntimeError as e:
            print("skip", N, Cin, Cout, S, var, e)
            continue
        yd, rd = direct(x, w, b)
        err = (yw - yd).abs().max().item()
        tw, td = timeit(rw), timeit(rd)
        print(f"N{N} {Cin}->{Cout} {S}^3 v{var}: wino {tw:8.1f} us ({fl / tw / 1e6:6.1f} TF eq)  direct {td:8.1f} us "
              f"({fl / td / 1e6:6.1f} TF)  speedup {td / tw:.2f}  maxdiff {err:.2e}", flush=True)


if __name__ == "__main__":
    wgrad_main() if "wgrad" in sys.argv[1:] else main()

import sys, os, torch
sys.path.insert(0, "/root/repo/cv-ssl-mis_amd")
from mis_hip import ops
import torch.nn.functional as F
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, Cin, Cout, H, W) in [(1, 16, 16, 16, 16), (2, 16, 32, 32, 48), (3, 32, 16, 16, 32), (2, 64, 64, 32, 32), (1, 8, 16, 48, 16)]:
    x = torch.randn(N, Cin, 1, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1; b = torch.randn(Cout, device="cuda")
    ref = F.conv2d(x[:, :, 0].double(), w.double(), b.double(), padding=1)
    v = ops.conv_wino_select(N, Cin, Cout, 1, H, W, (3, 3))
    y = torch.empty(N, Cout, 1, H, W, device="cuda")
    ops.conv_fwd(x, ops.conv_pack(w, 6), b, y, Cin, Cout, (3, 3), wino=v)
    yd = torch.empty_like(y); ops.conv_fwd(x, ops.conv_pack(w, 0), b, yd, Cin, Cout, (3, 3))
    print(f"N{N} {Cin}->{Cout} {H}x{W} v{v}: wino err {(y[:, :, 0].double() - ref).abs().max().item():.3e} direct err {(yd[:, :, 0].double() - ref).abs().max().item():.3e} ref {ref.abs().max().item():.1f}", flush=True)
for (N, Cin, Cout, S) in [(48, 16, 16, 256), (48, 32, 16, 256), (48, 32, 32, 128), (48, 64, 64, 64), (48, 128, 128, 32), (48, 256, 256, 16), (24, 16, 16, 256), (48, 256, 128, 32)]:
    x = torch.randn(N, Cin, 1, S, S, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    v = ops.conv_wino_select(N, Cin, Cout, 1, S, S, (3, 3))
    y = torch.empty(N, Cout, 1, S, S, device="cuda"); yd = torch.empty_like(y)
    wt, wp = ops.conv_pack(w, 6), ops.conv_pack(w, 0)
    fw = lambda: ops.conv_fwd(x, wt, b, y, Cin, Cout, (3, 3), wino=v)
    fd = lambda: ops.conv_fwd(x, wp, b, yd, Cin, Cout, (3, 3))
    fw(); fd()
    fl = 2.0 * N * Cout * Cin * 9 * S * S
    tw, td = timeit(fw), timeit(fd)
    print(f"N{N} {Cin}->{Cout} {S}^2 v{v}: wino {tw:7.1f} us ({fl/tw/1e6:6.1f} TF eq) direct {td:7.1f} us ({fl/td/1e6:6.1f} TF) speedup {td/tw:.2f} maxdiff {(y-yd).abs().max().item():.2e}", flush=True)
print("---- weight gradient ----")
for (N, Cin, Cout, H, W) in [(1, 16, 16, 8, 16), (2, 16, 32, 32, 48), (3, 24, 40, 16, 32), (2, 64, 64, 32, 32), (1, 8, 16, 48, 16)]:
    x = torch.randn(N, Cin, 1, H, W, device="cuda"); dy = torch.randn(N, Cout, 1, H, W, device="cuda")
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x[:, :, 0].cpu().double(), w, padding=1).backward(dy[:, :, 0].cpu().double())
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda"); ops.conv_wgrad(x, dy, dw, (3, 3))
    keep, ops.WINO = ops.WINO, 0
    dwd = torch.empty_like(dw); ops.conv_wgrad(x, dy, dwd, (3, 3)); ops.WINO = keep
    sc = w.grad.abs().max().item()
    print(f"wgrad N{N} {Cin}->{Cout} {H}x{W}: wino err {(dw.cpu().double() - w.grad).abs().max().item()/sc:.3e} direct err {(dwd.cpu().double() - w.grad).abs().max().item()/sc:.3e}", flush=True)
for (N, Cin, Cout, S) in [(48, 16, 16, 256), (48, 32, 16, 256), (48, 32, 32, 128), (48, 64, 64, 64), (48, 128, 128, 32), (48, 256, 256, 16), (48, 256, 128, 32)]:
    x = torch.randn(N, Cin, 1, S, S, device="cuda"); dy = torch.randn(N, Cout, 1, S, S, device="cuda")
    dw = torch.empty(Cout, Cin, 3, 3, device="cuda"); dwd = torch.empty_like(dw)
    fw = lambda: ops.conv_wgrad(x, dy, dw, (3, 3))
    def fd():
        keep, ops.WINO = ops.WINO, 0
        ops.conv_wgrad(x, dy, dwd, (3, 3)); ops.WINO = keep
    fw(); fd()
    fl = 2.0 * N * Cout * Cin * 9 * S * S
    tw, td = timeit(fw), timeit(fd)
    print(f"wgrad N{N} {Cin}->{Cout} {S}^2: wino {tw:7.1f} us ({fl/tw/1e6:6.1f} TF eq) direct {td:7.1f} us ({fl/td/1e6:6.1f} TF) speedup {td/tw:.2f} rel diff {(dw-dwd).abs().max().item()/dwd.abs().max().item():.2e}", flush=True)

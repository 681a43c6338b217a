import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import ops
torch.manual_seed(0)
def run(N, Cin, Cout, D, H, W, zx=None, zd=None):
    x = torch.randn(N, Cin, D, H, W, dtype=torch.float64)
    dy = torch.randn(N, Cout, D, H, W, dtype=torch.float64)
    if zx is not None:
        m = torch.zeros(D, dtype=torch.float64); m[zx] = 1; x = x * m[None, None, :, None, None]
    if zd is not None:
        m = torch.zeros(D, dtype=torch.float64); m[zd] = 1; dy = dy * m[None, None, :, None, None]
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w, padding=1).backward(dy)
    dw = torch.full((Cout, Cin, 3, 3, 3), float("nan"), device="cuda")
    ops.conv_wgrad(x.float().cuda(), dy.float().cuda(), dw, (3, 3, 3))
    err = (dw.cpu().double() - w.grad).abs()
    print(f"N{N} {Cin}->{Cout} {D}x{H}x{W} zx={zx} zd={zd}: max err {err.max().item():.3e} (scale {w.grad.abs().max().item():.2e}); "
          f"err by tap z: {[round(err[:, :, k].max().item(), 3) for k in range(3)]}", flush=True)
pass

for zd in range(4):
    for zx in range(4):
        if abs(zd - zx) <= 1:
            run(8, 16, 16, 4, 4, 32, zx=[zx], zd=[zd])

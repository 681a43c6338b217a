"""Forward Winograd kernels on the small levels (6^3, 12^3, 24^3) of unet_3D / V-Net: direct kernel, the unsplit variants, and the
product path with its contraction slices (development tool)."""
import sys, os, torch
sys.path.insert(0, "/root/repo/cv-ssl-mis_amd"); sys.path.insert(0, "/root/repo/scripts")
import wino_bench as wb
from mis_hip import ops
for (N, Cin, Cout, S) in [(8,256,256,6),(4,256,256,6),(8,128,256,6),(8,256,128,6),(8,128,128,12),(4,128,128,12),(8,384,128,12),(8,64,64,24),(4,64,64,24)]:
    x = torch.randn(N, Cin, S, S, S, device="cuda"); w = torch.randn(Cout, Cin, 3,3,3, device="cuda")*0.05; b = torch.randn(Cout, device="cuda")
    fl = 2.0*N*Cout*Cin*27*S**3
    var = ops.conv_wino_select(N, Cin, Cout, S, S, S, (3,3,3))
    yd, rd = wb.direct(x, w, b); td = wb.timeit(rd)
    line = f"N{N} {Cin}->{Cout} {S}^3: select {var}  direct {td:7.1f} us ({fl/td/1e6:6.1f} TF)"
    for v in ([2] if S in (6,24) else [3,2]):
        try:
            yw, rw = wb.wino(x, w, b, v); tw = wb.timeit(rw)
            line += f"   v{v} {tw:7.1f} us ({fl/3.375/tw/157.3e6:.3f} of the pipe)"
        except Exception as e:
            line += f"   v{v} n/a"
    if var >= 0:      # the product path: mis_conv3d_wino_fwd_ws (contraction slices when the launch has few boxes)
        from mis_hip import lib as _l
        ks = _l.load().mis_conv3d_wino_fwd_splits(N, Cin, Cout, S, S, S, var)
        wt = ops.conv_pack(w, 4)
        y = torch.empty(N, Cout, S, S, S, device="cuda")
        rp = lambda: ops.conv_fwd(x, wt, b, y, Cin, Cout, (3, 3, 3), wino=var)
        rp()
        tp = wb.timeit(rp)
        line += f"   product (v{var}, {ks} slices) {tp:7.1f} us ({fl/3.375/tp/157.3e6:.3f})  maxdiff {(y - yd).abs().max().item():.1e}"
    print(line, flush=True)

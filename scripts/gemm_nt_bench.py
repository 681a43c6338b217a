"""Forward and dX GEMMs of SwinUnet (24 + 24 images of 224 x 224) through mis_gemm: time per shape (the staged NT kernels;
round 4 measured a register-only NT form against them with this script -- 2277 us against 1750 us -- and dropped it, gemm.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import tops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def _opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


B, IMG = _opt("--batch", 48), _opt("--img", 224)
shapes = []
for stage, C in enumerate((96, 192, 384, 768)):
    T = B * ((IMG // 4) >> stage) ** 2
    for cout, cin in ((3 * C, C), (C, C), (4 * C, C), (C, 4 * C)):
        shapes += [(T, cout, cin), (T, cin, cout)]          # forward, dX
tot = 0.0
seen = set()
for M, N, K in shapes:
    if (M, N, K) in seen:
        continue
    seen.add((M, N, K))
    a, w, bias = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(N, device="cuda")
    c = torch.empty(M, N, device="cuda")
    # --split: pre-split planes for the staged kernels; --rega: planes in the natural order where the register-A kernel serves
    b3 = tops.SplitB(w, rows=M if "--rega" in sys.argv else None).refresh() if ("--split" in sys.argv or "--rega" in sys.argv) else None
    t = timeit(lambda: tops.gemm(a, w, c, bias=bias, b3=b3))
    ref = a[:512].double() @ w.double().t() + bias.double()
    err = (c[:512].double() - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * M * N * K
    tot += t
    # the two floors of this shape: the fp32 matrix pipe at 2.4 GHz, and A + C (+ W once) over HBM at 5 TB/s (what the
    # streaming passes of this code reach)
    t_mfma, t_hbm = fl / 157.3e6, (M * K + M * N + N * K) * 4 / 5e6
    tag = "R" if (b3 is not None and b3.natural) else " "
    print(f"{tag} M={M:7d} N={N:5d} K={K:5d}: {t:8.1f} us  {fl / t / 1e6:6.1f} TF ({fl / t / 1e6 / 157.3:.3f})  floors: pipe {t_mfma:6.1f} us, "
          f"HBM {t_hbm:6.1f} us -> {t / max(t_mfma, t_hbm):.2f}x the larger, {t / (t_mfma + t_hbm):.2f}x their sum  rel err {err:.1e}", flush=True)
print(f"sum {tot:.1f} us")

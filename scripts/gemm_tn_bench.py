"""dW GEMMs of SwinUnet (24 + 24 images of 224 x 224) through mis_gemm_dw: time per shape.  MIS_GEMM_TN_REG=0 selects the staged
(LDS) kernel, default the register-only one (gemm.hip): run twice to compare."""
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


B = 48
shapes = []
for stage, C in enumerate((96, 192, 384, 768)):
    T = B * (56 >> stage) ** 2
    shapes += [(T, 3 * C, C), (T, C, C), (T, 4 * C, C), (T, C, 4 * C)]
shapes += [(B * 28 * 28, 192, 384), (B * 56 * 56, 1536, 96)]
tot = 0.0
for T, Cout, Cin in shapes:
    x, dy = torch.randn(T, Cin, device="cuda"), torch.randn(T, Cout, device="cuda")
    dw, db = torch.empty(Cout, Cin, device="cuda"), torch.empty(Cout, device="cuda")
    t = timeit(lambda: tops.gemm_dw(dy, x, dw, db))
    ref = dy.double().t() @ x.double()
    err = (dw.double() - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * T * Cout * Cin
    tot += t
    print(f"T={T:7d} {Cin:5d} -> {Cout:5d}: {t:8.1f} us  {fl / t / 1e6:6.1f} TF ({fl / t / 1e6 / 157.3:.3f})  rel err {err:.1e}", flush=True)
print(f"sum {tot:.1f} us  [MIS_GEMM_TN_REG={os.environ.get('MIS_GEMM_TN_REG', '1')}]")

"""Development harness: LayerNorm forward / backward and the fused LayerNorm + head on SwinUnet's row counts (HBM rates).
    python scripts/ln_bench.py        (MIS_HIP_LIB=<other build> for an A/B)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import tops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for (M, C) in [(150528, 96), (37632, 192), (9408, 384), (2352, 768), (602112, 96)]:
        x = torch.randn(M, C, device="cuda")
        dy = torch.randn(M, C, device="cuda")
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        mb = M * C * 4 / 1e6
        t = timeit(lambda: tops.layernorm_fwd(x, y, g, b, mean, rstd))
        print(f"M={M} C={C}  fwd {t:7.1f} us {2 * mb / t:5.2f} TB/s", end="  ")
        t = timeit(lambda: tops.layernorm_bwd(x, dy, dx, g, mean, rstd, dg, db))
        print(f"bwd {t:7.1f} us {3 * mb / t:5.2f} TB/s", flush=True)
    B, S, C, NC = 48, 50176, 96, 4
    x = torch.randn(B * S, C, device="cuda")
    g, b, w = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.randn(NC, C, device="cuda")
    mean, rstd = torch.empty(B * S, device="cuda"), torch.empty(B * S, device="cuda")
    logits = torch.empty(B, NC, 1, 224, 224, device="cuda")
    dl = torch.randn_like(logits)
    dx = torch.empty_like(x)
    dg, db, dw = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(NC, C, device="cuda")
    mb = B * S * C * 4 / 1e6
    t = timeit(lambda: tops.ln_head_fwd(x, g, b, w, mean, rstd, logits))
    print(f"ln_head fwd {t:7.1f} us {mb / t:5.2f} TB/s", end="  ")
    t = timeit(lambda: tops.ln_head_bwd(x, g, b, w, mean, rstd, dl, dx, dg, db, dw))
    print(f"bwd {t:7.1f} us {2 * mb / t:5.2f} TB/s")


if __name__ == "__main__":
    main()

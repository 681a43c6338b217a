"""Timing of the HBM-bound kernels of the 3-D step (config 3 shapes, student batch 8): achieved TB/s =
algorithmic bytes (each operand once) / HIP-event time.  Usage: python scripts/ew_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))
import torch

from mis_hip import ops


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rep(name, ms, nbytes):
    print(f"{name:46s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s")


def main():
    N = 8
    for C, D in ((32, 48), (64, 24), (128, 12)):          # up-sampling 48^3 -> 96^3 etc. (C = channels up-sampled)
        x = torch.randn(N, C, D, D, D, device="cuda")
        y = torch.empty(N, C, 2 * D, 2 * D, 2 * D, device="cuda")
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        b = (x.numel() + y.numel()) * 4
        rep(f"upsample tri2 fwd  C={C} {D}^3->{2 * D}^3", timeit(lambda: ops.upsample2_fwd(x, y, False)), b)
        rep(f"upsample tri2 bwd  C={C} {2 * D}^3->{D}^3", timeit(lambda: ops.upsample2_bwd(dy, dx, False)), b)
    for C, D in ((16, 96), (32, 48), (64, 24)):
        x = torch.randn(N, C, D, D, D, device="cuda")
        y = torch.empty(N, C, D // 2, D // 2, D // 2, device="cuda")
        idx = torch.empty(y.numel(), dtype=torch.uint8, device="cuda")
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        rep(f"maxpool fwd C={C} {D}^3", timeit(lambda: ops.maxpool2_fwd(x, y, idx)), (x.numel() + y.numel()) * 4 + idx.numel())
        rep(f"maxpool bwd C={C} {D}^3", timeit(lambda: ops.maxpool2_bwd(dy, idx, dx)), (x.numel() + y.numel()) * 4 + idx.numel())
        rep(f"maxpool bwd accumulate C={C} {D}^3", timeit(lambda: ops.maxpool2_bwd(dy, idx, dx, accumulate=True)),
            (2 * x.numel() + y.numel()) * 4 + idx.numel())
    for C, D in ((16, 96), (32, 48), (64, 24)):
        x = torch.randn(N, C, D, D, D, device="cuda")
        y = torch.empty_like(x)
        da = torch.randn_like(x)
        dx = torch.empty_like(x)
        mean = torch.empty(N * C, device="cuda")
        rstd = torch.empty(N * C, device="cuda")
        rep(f"norm stats (IN) C={C} {D}^3", timeit(lambda: ops.norm_stats(x, True, 1e-5, mean, rstd)), x.numel() * 4)
        rep(f"norm apply fwd (IN+ReLU) C={C} {D}^3", timeit(lambda: ops.norm_act_fwd(x, y, True, mean, rstd, None, None, 0.0)),
            2 * x.numel() * 4)
        rep(f"norm bwd (partial+final+apply) C={C} {D}^3",
            timeit(lambda: ops.norm_act_bwd(x, da, dx, True, mean, rstd, None, None, 0.0)), 3 * x.numel() * 4)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def chunked():
    """Infinity-cache experiment: InstanceNorm backward run per sample (x + da of one sample = 113 MB at 16 x 96^3), so
    that the apply pass re-reads what the partial-sum pass just streamed."""
    N, C, D = 8, 16, 96
    x = torch.randn(N, C, D, D, D, device="cuda")
    da = torch.randn_like(x)
    dx = torch.empty_like(x)
    mean = torch.zeros(N * C, device="cuda")
    rstd = torch.ones(N * C, device="cuda")
    whole = timeit(lambda: ops.norm_act_bwd(x, da, dx, True, mean, rstd, None, None, 0.0))
    for step in (1, 2, 4):
        def run():
            for n in range(0, N, step):
                ops.norm_act_bwd(x[n:n + step], da[n:n + step], dx[n:n + step], True, mean[n * C:(n + step) * C],
                                 rstd[n * C:(n + step) * C], None, None, 0.0)
        print(f"norm bwd 16x96^3: whole batch {whole * 1e3:.1f} us; {step} sample(s) per call {timeit(run) * 1e3:.1f} us")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "chunked":
    chunked()

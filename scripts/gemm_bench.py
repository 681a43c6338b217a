"""Development harness: NT GEMM (mis_gemm / mis_gemm_ex) on the SwinUnet Linear shapes.
    python scripts/gemm_bench.py            timing of the stage-1 / stage-2 shapes, plain and with the fused epilogues"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import tops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.manual_seed(0)
    for (M, N, K, ep) in [(150528, 384, 96, 0), (150528, 384, 96, 1), (150528, 384, 96, 2), (150528, 96, 384, 3),
                          (150528, 288, 96, 0), (150528, 96, 96, 3), (37632, 768, 192, 1), (37632, 192, 768, 3),
                          (37632, 576, 192, 0), (9408, 1536, 384, 1)]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        E = torch.randn(M, N, device="cuda") if ep >= 2 else None
        C2 = torch.empty(M, N, device="cuda") if ep == 1 else None
        if ep == 0:
            fn = lambda: tops.gemm(A, W, C, bias=b)
        elif ep == 1:
            fn = lambda: tops.gemm_ex(A, W, C, tops.EP_GELU_FWD, bias=b, C2=C2)
        elif ep == 2:
            fn = lambda: tops.gemm_ex(A, W, C, tops.EP_GELU_BWD, E1=E)
        else:
            fn = lambda: tops.gemm_ex(A, W, C, tops.EP_RESIDUAL, bias=b, E1=E)
        t = timeit(fn)
        byt = 4.0 * (M * K + N * K + M * N * (1 + (ep > 0)))
        print(f"M={M:7d} N={N:5d} K={K:4d} ep{ep}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF  {byt / t / 1e6:5.2f} TB/s (ideal bytes)",
              flush=True)


if __name__ == "__main__":
    main()

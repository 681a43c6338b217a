"""Development harness: the in-place kernel-2 / stride-2 kernels (conv_k2s2.hip) on V-Net's two largest levels.
    python scripts/k2s2_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import ops  # noqa: E402


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
    for (N, cf, cc, d) in [(8, 16, 32, 48), (8, 32, 64, 24)]:
        fine = torch.randn(N, cf, 2 * d, 2 * d, 2 * d, device="cuda")
        coarse = torch.randn(N, cc, d, d, d, device="cuda")
        w = torch.randn(cc, cf * 8, device="cuda") * 0.1
        y = torch.empty_like(coarse)
        dx = torch.empty_like(fine)
        dw = torch.empty(cc * cf * 8, device="cuda")
        mb = (fine.numel() + coarse.numel()) * 4 / 1e6
        t = timeit(lambda: ops.conv_k2s2_down(fine, w, None, y))
        print(f"N={N} {cf}->{cc} coarse {d}^3  down  {t:7.1f} us  {mb / t:6.2f} TB/s")
        t = timeit(lambda: ops.conv_k2s2_up(coarse, w, None, dx))
        print(f"N={N} {cc}->{cf}             up    {t:7.1f} us  {mb / t:6.2f} TB/s")
        t = timeit(lambda: ops.conv_k2s2_wgrad(coarse, fine, dw))
        print(f"                            wgrad {t:7.1f} us  {mb / t:6.2f} TB/s")


if __name__ == "__main__":
    main()

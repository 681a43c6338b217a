"""Development: NT GEMM time per shape under the MIS_GEMM_BM setting of this process (run once per setting and compare).
    MIS_GEMM_BM=64 python scripts/gemm_shapes_ab.py; MIS_GEMM_BM=128 python scripts/gemm_shapes_ab.py; python scripts/gemm_shapes_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import tops  # noqa: E402
from gemm_bench import timeit  # noqa: E402

SHAPES = [(37632, 576, 192), (37632, 192, 192), (37632, 768, 192), (37632, 192, 768),
          (9408, 1152, 384), (9408, 384, 384), (9408, 1536, 384), (9408, 384, 1536),
          (2352, 2304, 768), (2352, 768, 768), (2352, 3072, 768), (2352, 768, 3072),
          (12544, 576, 192), (12544, 768, 192), (12544, 192, 768), (3136, 1152, 384), (3136, 1536, 384), (3136, 384, 1536),
          (784, 2304, 768), (784, 768, 768), (784, 3072, 768), (784, 768, 3072),
          (1728, 2304, 768), (1728, 768, 768), (1728, 3072, 768), (1728, 768, 3072), (864, 768, 768), (864, 3072, 768),
          (864, 768, 3072), (4704, 768, 1536), (18816, 384, 768), (75264, 192, 384)]
for (M, N, K) in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; C = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    t = timeit(lambda: tops.gemm(A, W, C, bias=b), n=30)
    print(f"{M} {N} {K} {t:.1f}", flush=True)

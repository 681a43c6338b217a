"""Window-attention kernel timings at the SwinUnet stage geometries (48-image student batch).
Usage: python scripts/attn_bench.py   (MIS_ATTN_VALU=1 selects the vector-pipe kernels)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))
import torch

from mis_hip import tops


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = 48
for H, nH, shift in ((56, 3, 0), (56, 3, 3), (28, 6, 3), (14, 12, 3), (7, 24, 0)):
    C = nH * 32
    M = B * H * H
    qkv = torch.randn(M, 3 * C, device="cuda")
    out = torch.empty(M, C, device="cuda")
    dout = torch.randn(M, C, device="cuda")
    dqkv = torch.empty_like(qkv)
    table = torch.randn(169, nH, device="cuda") * 0.1
    dtable = torch.zeros_like(table)
    scale = 32 ** -0.5
    tf = timeit(lambda: tops.window_attention_fwd(qkv, out, table, B, H, H, nH, shift, scale))
    tb = timeit(lambda: tops.window_attention_bwd(qkv, dout, dqkv, table, dtable, B, H, H, nH, shift, scale))
    units = B * (H // 7) ** 2 * nH
    fl = units * 2 * 2 * 49 * 49 * 32
    # HBM floors at 5 TB/s: forward reads qkv, writes out; backward reads qkv and d(out), writes d(qkv)
    bf, bb = (M * 4 * C) * 4 / 5e6, (M * 7 * C) * 4 / 5e6
    print(f"H={H:3d} nH={nH:2d} shift={shift} units={units:6d}  fwd {tf:8.1f} us ({fl / tf / 1e6:6.2f} TF alg, {tf / bf:.2f}x its HBM floor)   "
          f"bwd {tb:8.1f} us ({2.5 * fl / tb / 1e6:6.2f} TF alg, {tb / bb:.2f}x its HBM floor)")

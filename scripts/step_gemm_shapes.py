"""Per-shape achieved rate of the event-timed GEMM launches inside one SwinUnet Mean-Teacher step."""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from mis_hip import ops, tops
from mis_hip.step import MeanTeacherTrainer

wl = bench.WORKLOADS["swin"]
torch.manual_seed(1337)
tr = bench.build_trainer("swin", wl, 1)
g = torch.Generator(device="cuda").manual_seed(1337)
vol = torch.rand(wl["shape"], generator=g, device="cuda")
lab = torch.randint(0, wl["classes"], (wl["shape"][0],) + wl["shape"][2:], generator=g, device="cuda").to(getattr(torch, wl["label"]))
for _ in range(3):
    tr.step(vol, lab)
orig = tops.gemm
geo = []


def wrapped(A, B, C, bias=None, trans=False, accumulate=False):
    geo.append((tuple(C.shape), A.shape[0] if trans else A.shape[1], bool(trans)))
    return orig(A, B, C, bias=bias, trans=trans, accumulate=accumulate)


orig_dw = tops.gemm_dw


def wrapped_dw(dy, x, dW, db, accumulate=False):
    geo.append((tuple(dW.shape), dy.shape[0], True))
    return orig_dw(dy, x, dW, db, accumulate=accumulate)


orig_ex, orig_expand = tops.gemm_ex, tops.gemm_expand


def wrapped_ex(A, B, C, epilogue, **kw):
    ok = orig_ex(A, B, C, epilogue, **kw)
    if ok:
        geo.append((tuple(C.shape), A.shape[1], False))
    return ok


def wrapped_expand(x, w, out, B, H, W, P, c):
    ok = orig_expand(x, w, out, B, H, W, P, c)
    if ok:
        geo.append(((x.shape[0], w.shape[0]), x.shape[1], False))
    return ok


tops.gemm = wrapped
tops.gemm_dw = wrapped_dw
tops.gemm_ex = wrapped_ex
tops.gemm_expand = wrapped_expand
prof = []
ops.PROFILE = prof
steps = 5
for _ in range(steps):
    tr.step(vol, lab)
torch.cuda.synchronize()
ops.PROFILE = None
gem = [p for p in prof if p[0].startswith("gemm")]
assert len(gem) == len(geo), (len(gem), len(geo))
acc = defaultdict(lambda: [0.0, 0.0, 0])
for (name, flops, e0, e1), gq in zip(gem, geo):
    k = (name, gq)
    acc[k][0] += flops
    acc[k][1] += e0.elapsed_time(e1) * 1e-3
    acc[k][2] += 1
tot = sum(v[1] for v in acc.values())
for (name, gq), (fl, t, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:32]:
    (M, N), K, tr_ = gq
    byt = 4.0 * (M * K + N * K + M * N)
    print(f"{name:20s} M={M:7d} N={N:5d} K={K:7d} {'TN' if tr_ else 'NT'} n/step={n // steps:2d} {t / n * 1e6:8.1f} us "
          f"{fl / t / 1e12:6.1f} TF  ideal-bytes {byt / (t / n) / 1e12:5.2f} TB/s  {100 * t / tot:5.1f}%")

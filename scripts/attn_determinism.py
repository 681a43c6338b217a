"""Window attention, bf16x3 vs fp32 MFMA form: run-to-run determinism (alone and beside a busy second stream) and agreement."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import tops

def run(B, H, nH, shift, ws, busy):
    C = nH * 32
    g = torch.Generator().manual_seed(1)
    qkv = (torch.rand(B * H * H, 3 * C, generator=g) * 3 - 1.5).cuda()
    dout = (torch.rand(B * H * H, C, generator=g) * 2 - 1).cuda()
    table = (torch.rand((2 * ws - 1) ** 2, nH, generator=g) - 0.5).cuda()
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    res = {}
    for mask in (0, 7):
        tops.set_split_precision(mask)
        outs = []
        for rep in range(4):
            out = torch.full((B * H * H, C), float("nan"), device="cuda")
            dqkv = torch.full((B * H * H, 3 * C), float("nan"), device="cuda")
            dt = torch.empty((2 * ws - 1) ** 2, nH, device="cuda")
            if busy:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(6):
                        big = torch.tanh(big @ big * 1e-4)
            tops.window_attention_fwd(qkv, out, table, B, H, H, nH, shift, 32 ** -0.5, window=ws)
            tops.window_attention_bwd(qkv, dout, dqkv, table, dt, B, H, H, nH, shift, 32 ** -0.5, window=ws)
            torch.cuda.synchronize()
            outs.append((out.clone(), dqkv.clone(), dt.clone()))
        same = all(all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:])
        fin = all(torch.isfinite(t).all().item() for t in outs[0])
        res[mask] = outs[0]
        print(f"  mask {mask}: deterministic {same} finite {fin}")
    for name, a, b in zip(("out", "dqkv", "dtable"), res[0], res[7]):
        print(f"  {name}: max |bf16x3 - fp32| = {(a - b).abs().max().item():.3e} (scale {a.abs().max().item():.3e})")

for cfg in ((32, 64, 3, 4, 8), (32, 32, 6, 4, 8), (32, 16, 12, 0, 8), (32, 8, 24, 0, 8), (48, 56, 3, 3, 7), (32, 64, 3, 0, 8)):
    for busy in (False, True):
        print(cfg, "busy" if busy else "alone")
        run(*cfg, busy)

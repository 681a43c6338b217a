"""Record, on the GPU, which ranges of the flat gradient buffer every op of a real plan WRITES in its backward and
which ranges ``mis_hip.dist.param_progress`` believes it owns (attribute scan) -> tests/golden/plan_grad_ranges.json.

tests/test_dist_cpu.py replays the recording over gloo (world 2): fake ops write the recorded ranges in the recorded
order while the real ``param_progress`` + ``GradBucketer`` decide when a bucket is exchanged.

    python scripts/record_grad_ranges.py        (on an MI355X; the JSON is data, not source)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

from mis_hip import plan as _plan

SENTINEL = 12345.678


def runs(mask):
    """[lo, hi) runs of a boolean vector."""
    idx = mask.nonzero().flatten().tolist()
    out = []
    for i in idx:
        if out and out[-1][1] == i:
            out[-1][1] = i + 1
        else:
            out.append([i, i + 1])
    return out


def record(kind, shape, C):
    from test_grad_progress_gpu import _make, _owned_ranges
    torch.manual_seed(3)
    model = _make(kind, C)
    model.train()
    model.dropout_enabled = False
    model.forward_raw(torch.rand(shape, device="cuda"))
    dl = model.logits_grad_buffer()
    dl.copy_(torch.randn(dl.shape, device="cuda") * 0.1)
    pl = model._last[0]
    model.flat_grad.fill_(SENTINEL)
    snap = [model.flat_grad.clone()]
    writes = {}
    for i, op in enumerate(pl.ops):
        def wrap(i=i, orig=op.bwd):
            def f(ctx):
                orig(ctx)
                torch.cuda.synchronize()
                cur = model.flat_grad.clone()
                changed = cur.view(torch.int32) != snap[0].view(torch.int32)
                writes[i] = runs(changed.cpu())
                snap[0] = cur
            return f
        op.bwd = wrap()
    model.backward_raw()
    torch.cuda.synchronize()
    owned = []
    base, esz = model.flat_grad.data_ptr(), 4
    for op in pl.ops:
        r = []
        for v in vars(op).values():
            g = getattr(v, "grad", None)
            if isinstance(g, torch.Tensor) and hasattr(v, "data") and g.numel() and \
                    base <= g.data_ptr() < base + model.flat_grad.numel() * esz:
                lo = (g.data_ptr() - base) // esz
                r.append([lo, lo + g.numel()])
        owned.append(sorted(r))
    never = runs((model.flat_grad == SENTINEL).cpu())
    return dict(total=model.flat_grad.numel(), ops=[dict(type=type(op).__name__, owns=owned[i], writes=writes.get(i, []))
                                                    for i, op in enumerate(pl.ops)],
                never_written=len(never))


def main():
    _plan.WGRAD_STREAM = False
    out = {}
    for kind, shape, C in [("unet2d", (2, 1, 64, 64), 4), ("unet3d", (2, 1, 32, 32, 32), 2),
                           ("vnet_groupnorm", (2, 1, 32, 32, 32), 2), ("swin", (2, 1, 224, 224), 4)]:
        out[kind] = record(kind, shape, C)
        late = 0
        print(kind, "ops", len(out[kind]["ops"]), "total", out[kind]["total"], "never written runs", out[kind]["never_written"])
    path = os.path.join(ROOT, "tests", "golden", "plan_grad_ranges.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        with open(os.path.join(ROOT, "gpurun_out", "plan_grad_ranges.json"), "w") as f:
            json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()

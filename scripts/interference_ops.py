"""Which op of the 2-D UNet's forward is disturbed by bf16x3 window-attention waves on a second stream?  Every op is re-run
from the quiet run's inputs (the plan's buffers hold them) beside the busy stream and its output compared bit for bit."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd")); sys.path.insert(0, ROOT)
from mis_hip import tops
from networks.net_factory import net_factory

torch.manual_seed(0)
if len(sys.argv) > 2 and sys.argv[2] == "3d":
    from networks.net_factory_3d import net_factory_3d
    net = net_factory_3d("unet_3D", 1, 2); x = torch.rand(4, 1, 96, 96, 96, device="cuda")
else:
    net = net_factory("unet", 1, 4); x = torch.rand(32, 1, 256, 256, device="cuda")
net.train(); net.dropout_enabled = False
net.forward_raw(x); torch.cuda.synchronize()
plan, ctx = net._last
side = torch.cuda.Stream()
M = 131072
qkv = torch.randn(M, 288, device="cuda"); out = torch.empty(M, 96, device="cuda")
table = torch.randn(225, 3, device="cuda") * 0.1
tops.set_split_precision(int(sys.argv[1]) if len(sys.argv) > 1 else 7)

def outputs(op):
    outs = []
    for name in ("y", "out", "dst"):
        a = getattr(op, name, None)
        if a is not None and hasattr(a, "t"):
            outs.append(a.t)
    st = getattr(op, "stat", None)
    if st is not None:
        outs.append(st[0])
    for name in ("mean", "rstd"):
        a = getattr(op, name, None)
        if isinstance(a, torch.Tensor):
            outs.append(a)
    return outs

# quiet reference of every op's outputs, in order (buffers are NOT reused between ops of one forward)
plan.forward(net._as5(x), ctx); torch.cuda.synchronize()
ref = [[t.clone() for t in outputs(op)] for op in plan.ops]
for i, op in enumerate(plan.ops):
    bad = 0
    for rep in range(6):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(20):
                tops.window_attention_fwd(qkv, out, table, 32, 64, 64, 3, 4, 32 ** -0.5, window=8)
        op.fwd(ctx)
        torch.cuda.synchronize()
        cur = outputs(op)
        if any(not torch.equal(a, b) for a, b in zip(cur, ref[i])):
            bad += 1
            w = [(a - b).abs().max().item() for a, b in zip(cur, ref[i])]
    for a, b in zip(outputs(op), ref[i]):       # the next op starts from the quiet run's values again
        a.copy_(b)
    desc = type(op).__name__ + " " + " ".join(str(tuple(getattr(op, n).t.shape)) for n in ("x", "y") if hasattr(getattr(op, n, None), "t"))
    if bad:
        print(f"op {i:3d} {desc}: differs in {bad}/6 (max per output {w})", flush=True)
print("done", len(plan.ops), "ops")

# ---- pattern of the differences of the first Winograd conv (op 2)
if len(sys.argv) > 3:
    i = int(sys.argv[3]); op = plan.ops[i]
    for a, b in zip(outputs(op), ref[i]):
        a.copy_(b)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(20):
            tops.window_attention_fwd(qkv, out, table, 32, 64, 64, 3, 4, 32 ** -0.5, window=8)
    op.fwd(ctx); torch.cuda.synchronize()
    y, r = outputs(op)[0], ref[i][0]
    d = (y != r)
    print("wrong elements", int(d.sum()), "of", d.numel(), "nan", int(torch.isnan(y).sum()))
    idx = torch.nonzero(d)
    for dim, name in ((0, "n"), (1, "c"), (3, "y"), (4, "x")):
        vals, cnt = torch.unique(idx[:, dim], return_counts=True)
        print(name, "distinct", vals.numel(), "first", vals[:24].tolist(), "counts", cnt[:24].tolist())
    # per box (8 x 32 pixels): how many boxes are touched, are whole boxes wrong?
    bx = (idx[:, 4] // 32) + 8 * (idx[:, 3] // 8) + 8 * 32 * idx[:, 0]
    ub, cb = torch.unique(bx, return_counts=True)
    print("boxes touched", ub.numel(), "of", 32 * 32 * 8, "elements per touched box: min", int(cb.min()), "max", int(cb.max()), "(box = 8*32*16ch = 4096)")
    e = (y - r).abs()
    print("max err", e.max().item(), "mean err over wrong", e[d].mean().item(), "mean |ref|", r.abs().mean().item())

"""Co-residency probe: does a foreign wave on the same SIMD change a kernel's results?  One op of a network's forward (default:
every op of the 2-D UNet at 32 x 256 x 256, or `3d`: unet_3D at 4 x 96^3) is re-run from the quiet run's inputs while
mis_debug_spin keeps one execution pipe busy on a second stream (kind 1 bf16 MFMA, 2 fp32 MFMA, 3 unpacked VALU, 4 packed
fp32 VALU), and its outputs are compared bit for bit with the quiet run.
    python scripts/interference.py [2d|3d] [bwd]"""
import ctypes, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd")); sys.path.insert(0, ROOT)
from mis_hip import lib as _l

torch.manual_seed(0)
if "3d" in sys.argv[1:]:
    from networks.net_factory_3d import net_factory_3d
    net = net_factory_3d("unet_3D", 1, 2); x = torch.rand(4, 1, 96, 96, 96, device="cuda")
else:
    from networks.net_factory import net_factory
    net = net_factory("unet", 1, 4); x = torch.rand(32, 1, 256, 256, device="cuda")
net.train(); net.dropout_enabled = False
net.forward_raw(x); torch.cuda.synchronize()
plan, ctx = net._last
side = torch.cuda.Stream()
sink = torch.zeros(1024, device="cuda")
L = _l.load()

def outputs(op):
    outs = []
    for name in ("y", "out", "dst"):
        a = getattr(op, name, None)
        if a is not None and hasattr(a, "t"):
            outs.append(a.t)
    st = getattr(op, "stat", None)
    if st is not None:
        outs.append(st[0])
    return outs

plan.forward(net._as5(x), ctx); torch.cuda.synchronize()
ref = [[t.clone() for t in outputs(op)] for op in plan.ops]
names = {1: "bf16 MFMA", 2: "fp32 MFMA", 3: "unpacked VALU", 4: "packed fp32 VALU"}
for kind in (1, 2, 3, 4):
    hit = []
    for i, op in enumerate(plan.ops):
        bad = 0
        for rep in range(3):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _l.check(L.mis_debug_spin(kind, 4096, 4000, _l.ptr(sink), _l.stream_ptr()), "mis_debug_spin")
            op.fwd(ctx)
            torch.cuda.synchronize()
            if any(not torch.equal(a, b) for a, b in zip(outputs(op), ref[i])):
                bad += 1
            for a, b in zip(outputs(op), ref[i]):
                a.copy_(b)
        if bad:
            hit.append(f"{i}:{type(op).__name__}{tuple(op.y.t.shape[1:2]) if hasattr(getattr(op, 'y', None), 't') else ''}x{bad}")
    print(f"beside {names[kind]:18s}: {len(hit)} of {len(plan.ops)} ops disturbed  {' '.join(hit[:20])}", flush=True)

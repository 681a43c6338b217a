"""Co-residency probe at network level: forward + backward of a CNN beside mis_debug_spin waves (kind 1 bf16 MFMA, 2 fp32 MFMA,
3 unpacked VALU, 4 packed fp32 VALU) on a second stream; logits and the flat gradient are compared bit for bit with a quiet run.
    python scripts/interference_step.py unet2d|unet3d|vnet|swin|unetr"""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd")); sys.path.insert(0, ROOT)
from mis_hip import lib as _l

kind = sys.argv[1] if len(sys.argv) > 1 else "unet2d"
torch.manual_seed(0)
if kind == "unet2d":
    from networks.net_factory import net_factory
    net = net_factory("unet", 1, 4); x = torch.rand(32, 1, 256, 256, device="cuda")
elif kind == "swin":
    from networks.net_factory import net_factory
    net = net_factory("ViT_Seg", 1, 4); x = torch.rand(16, 1, 224, 224, device="cuda")
elif kind == "unetr":
    from networks.net_factory_3d import net_factory_3d
    net = net_factory_3d("unetr", 1, 2); x = torch.rand(2, 1, 96, 96, 96, device="cuda")
else:
    from networks.net_factory_3d import net_factory_3d
    net = net_factory_3d("unet_3D" if kind == "unet3d" else "vnet", 1, 2); x = torch.rand(4, 1, 96, 96, 96, device="cuda")
net.train(); net.dropout_enabled = False
L = _l.load()
sink = torch.zeros(1024, device="cuda")
side = torch.cuda.Stream()

def run(spin):
    if spin:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _l.check(L.mis_debug_spin(spin, 4096, 60000, _l.ptr(sink), _l.stream_ptr()), "mis_debug_spin")
    y = net.forward_raw(x)
    dl = torch.full_like(y, 1e-3)
    net.flat_grad.zero_()
    net.backward_raw(dl)
    torch.cuda.synchronize()
    return y.clone(), net.flat_grad.clone()

ref = run(0)
again = run(0)
print(kind, "quiet rerun identical:", all(torch.equal(a, b) for a, b in zip(ref, again)))
for spin, name in ((1, "bf16 MFMA"), (2, "fp32 MFMA"), (3, "unpacked VALU"), (4, "packed fp32 VALU")):
    bad = [0, 0]
    for rep in range(3):
        cur = run(spin)
        for i in range(2):
            if not torch.equal(cur[i], ref[i]):
                bad[i] += 1
                d = (cur[i] - ref[i]).abs().max().item()
    print(f"  beside {name:18s}: logits differ {bad[0]}/3, gradient differs {bad[1]}/3" + (f" (last max diff {d:.2e})" if sum(bad) else ""), flush=True)

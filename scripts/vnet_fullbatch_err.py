"""Which kernel family carries the V-Net full-batch gradient noise?  (round 6, tests/test_fullbatch_gpu.py float64 arbiter)
    python scripts/vnet_fullbatch_err.py oracle /tmp/vnet_o.pt      float64 + fp32 CPU oracle gradients of config3_vnet_4+4_96
    MIS_WINO_FWD=0 python scripts/vnet_fullbatch_err.py hip /tmp/vnet_o.pt     the HIP step under the switches of the environment
prints, per gradient tensor above the gate, HIP error / the fp32 oracle's own error against float64."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in ("cv-ssl-mis_amd", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from oracle import filler
from oracle.step import mean_teacher_step
from test_fullbatch_gpu import MT_CASES, _states, _as64
from test_parity_gpu import _build

name = "config3_vnet_4+4_96"
kind, shape, L, C, ldt, it, cons_start, _ = MT_CASES[name]
torch.set_num_threads(32)
onet, make = _build(kind, C)
sd0, tsd0 = _states(onet), _states(onet, "t.")
volume = filler.image(shape, "volume"); label = filler.labels((shape[0],) + shape[2:], C, ldt)
noise = filler.noise((shape[0] - L,) + shape[1:], "noise")
mom = {n: filler.uniform(v.shape, "mom." + n, -0.01, 0.01) for n, v in sd0.items() if onet.is_param(n)}
mode, path = sys.argv[1], sys.argv[2]
if mode == "oracle":
    o32 = mean_teacher_step(onet, {k: v.clone() for k, v in sd0.items()}, {k: v.clone() for k, v in tsd0.items()}, {k: v.clone() for k, v in mom.items()},
                            volume, label, noise, it, labeled_bs=L, num_classes=C, cons_start_iter=cons_start, drop_student="off",
                            drop_teacher="off", apply_update=False)
    o64 = mean_teacher_step(onet, _as64(sd0), _as64(tsd0), _as64(mom), volume.double(), label, noise.double(), it, labeled_bs=L,
                            num_classes=C, cons_start_iter=cons_start, drop_student="off", drop_teacher="off", apply_update=False)
    torch.save(dict(g32=o32["grads"], g64=o64["grads"]), path)
    sys.exit(0)
from mis_hip.step import MeanTeacherTrainer
o = torch.load(path)
model, ema = make(), make()
model.train(); ema.train()
model.dropout_enabled = ema.dropout_enabled = False
model.load_state_dict(sd0); ema.load_state_dict(tsd0)
tr = MeanTeacherTrainer(model, ema, labeled_bs=L, num_classes=C, cons_start_iter=cons_start, iter_num=it, use_tape=False)
for n, v in model.named_flat(tr.momentum_buf):
    v.copy_(mom[n])
tr.step(volume.cuda(), label.cuda(), noise=noise.cuda())
gscale = max(float(g.abs().max()) for g in o["g64"].values())
rows = []
for n, g in model.named_flat(model.flat_grad):
    ref = o["g64"][n]; gmax = float(ref.abs().max())
    if gmax < 1e-4 * gscale:
        continue
    e32 = (o["g32"][n].double() - ref).abs().max().item() / gmax
    err = (g.cpu().double() - ref).abs().max().item() / gmax
    rows.append((err / max(e32, 1e-3), n, err, e32))
rows.sort(reverse=True)
print("switches:", {k: v for k, v in os.environ.items() if k.startswith("MIS_")})
for r in rows[:8]:
    print(f"  {r[1]:40s} HIP/fp32-oracle {r[0]:6.2f}  HIP err {r[2]:.3e}  fp32 oracle err {r[3]:.3e}")
print("  median ratio", sorted(r[0] for r in rows)[len(rows) // 2])

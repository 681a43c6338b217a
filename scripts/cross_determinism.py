"""Cross teaching (UNet + SwinUnet, two streams): repeat the same step from the same state and compare the logits / gradients
of both networks bit for bit with the first repetition (a race between the streams shows as a difference)."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd")); sys.path.insert(0, ROOT)
from config import lite_config
from mis_hip.step import CrossTeachingTrainer
from networks.net_factory import net_factory
from networks.vision_transformer import SwinUnet

size, window, B, L, C = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 8, 32, 16, 4
if size == 224:
    window = 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = lite_config(); cfg.DATA.IMG_SIZE, cfg.MODEL.SWIN.WINDOW_SIZE = size, window
torch.manual_seed(0)
models = [net_factory("unet", 1, C), SwinUnet(cfg, img_size=size, num_classes=C)]
for m in models:
    m.train(); m.dropout_enabled = False
sd = [{k: v.clone() for k, v in m.state_dict().items()} for m in models]
vol = torch.rand(B, 1, size, size, device="cuda"); lab = torch.randint(0, C, (B, size, size), device="cuda").to(torch.uint8)
tr = CrossTeachingTrainer(models[0], models[1], labeled_bs=L, num_classes=C, iter_num=1300)
first = None
bad = 0
for r in range(reps):
    for m, s in zip(models, sd):
        m.load_state_dict(s)
    tr.mom1.zero_(); tr.mom2.zero_()
    tr.step(vol, lab)
    torch.cuda.synchronize()
    cur = [models[0]._last[0].out.t.clone(), models[1]._last[0].out.t.clone(), models[0].flat_grad.clone(), models[1].flat_grad.clone()]
    if first is None:
        first = cur
        continue
    for name, a, b in zip(("logits1", "logits2", "grad1", "grad2"), first, cur):
        if not torch.equal(a, b):
            d = (a - b).abs()
            idx = torch.nonzero(d.reshape(-1) > 0).flatten()
            print(f"rep {r}: {name} differs at {idx.numel()} elements, max {d.max().item():.3e}, first flat index {int(idx[0])} last {int(idx[-1])} shape {tuple(a.shape)}")
            bad += 1
print(f"size {size}: {bad} differing tensors over {reps - 1} repetitions; BF3={os.environ.get('MIS_GEMM_BF3','default')} TWO_STREAM={os.environ.get('MIS_TWO_STREAM','default')}")

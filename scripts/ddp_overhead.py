"""What the data-parallel code path costs a rank BEFORE any communication: the step with a (single-rank) RCCL process group
and the gradient bucketer active -- per-op progress reports, bucket all-reduces issued behind the weight-gradient side
stream -- against the plain single-GPU step.  One GPU is enough: a world of 1 exercises the same stream choreography.

    python scripts/ddp_overhead.py [workload ...]           (MIS_PROGRESS_SYNC_MAIN=1: the pre-round-3 reporting)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as tdist

import bench
from mis_hip import dist as mdist
from mis_hip import step as mstep


def run(name, pg, steps=20, warmup=5):
    wl = bench.WORKLOADS[name]
    torch.manual_seed(1337)
    tr = bench.build_trainer(name, wl, 1)
    g = torch.Generator(device="cuda").manual_seed(1337)
    vol = torch.rand(wl["shape"], generator=g, device="cuda")
    lab = torch.randint(0, wl["classes"], (wl["shape"][0],) + wl["shape"][2:], generator=g, device="cuda").to(getattr(torch, wl["label"]))
    for _ in range(warmup):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pg = tdist.group.WORLD
    keep = mstep.make_bucketer
    for name in (sys.argv[1:] or ["unet3d", "swin"]):
        plain, ddp = [], []
        for _ in range(2):       # alternate: the chip's clocks drift over a run
            mstep.make_bucketer = keep
            plain.append(run(name, None))
            mstep.make_bucketer = lambda model, group, defer_tail=False: mdist.GradBucketer(model.flat_grad, group, defer_tail=defer_tail)      # also with one rank
            ddp.append(run(name, pg))
        plain, ddp = min(plain), min(ddp)
        print(f"{name}: single-GPU step {plain:.3f} ms, data-parallel code path (world 1) {ddp:.3f} ms  (+{ddp - plain:.3f})",
              flush=True)
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()

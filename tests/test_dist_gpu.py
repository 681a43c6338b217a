"""The data-parallel step on real GPU kernels with a real collective: two processes share the one GPU of the test box
and exchange gradients over gloo (RCCL needs one device per rank; the collective API, the asynchronous work handles and
the stream hand-over of ``mis_hip.dist.GradBucketer`` are the same).  Checks: both ranks end with bit-identical weights;
the bucketed exchange overlapped with the backward (and with the weight gradients on their side stream) gives bit for
bit the weights of one blocking all-reduce after the backward.

Reference: single-GPU; this is the standard DDP contract of SURVEY s.8 row e."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, overlap, kind, steps=3, taped=False):
    os.environ["MIS_GRAD_OVERLAP"] = "1" if overlap else "0"
    for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from mis_hip import step as mstep
    from mis_hip.step import CrossTeachingTrainer, MeanTeacherTrainer
    from mis_hip.dist import GradBucketer
    from test_grad_progress_gpu import _make
    assert mstep.GRAD_OVERLAP == bool(overlap)
    torch.manual_seed(5)                                   # identical initial weights on both ranks
    C = 2 if kind == "unet3d" else 4
    if kind == "cross":      # two students (UNet + SwinUnet), the second one's backward on the side stream (TWO_STREAM)
        model, ema = _make("unet2d", C), _make("swin", C)
        model.train(); ema.train()
        tr = CrossTeachingTrainer(model, ema, labeled_bs=1, num_classes=C, iter_num=1000, seed=7)
        assert mstep.TWO_STREAM and (tr._bucketers[0] is not None) == bool(overlap)
        if overlap:
            assert tr._bucketers[1].defer_tail and not tr._bucketers[0].defer_tail
            tr._bucketers = (GradBucketer(model.flat_grad, None, bucket_bytes=1 << 20),
                             GradBucketer(ema.flat_grad, None, bucket_bytes=8 << 20, defer_tail=True))
    else:
        model, ema = _make(kind, C), _make(kind, C)
        ema.load_state_dict(model.state_dict())
        model.train(); ema.train()
        tr = MeanTeacherTrainer(model, ema, labeled_bs=1, num_classes=C, cons_start_iter=0, iter_num=1000, seed=7)
        assert (tr._bucketer is not None) == bool(overlap)
        if tr._bucketer is not None:
            # small buckets: several collectives per backward even for the small test networks
            tr._bucketer = GradBucketer(model.flat_grad, None, bucket_bytes=1 << 20)
            assert len(tr._bucketer.buckets) >= 3
    g = torch.Generator(device="cuda").manual_seed(100 + rank)          # a different shard per rank
    shape = {"unet2d": (2, 1, 64, 64), "cross": (2, 1, 224, 224)}.get(kind, (2, 1, 32, 32, 32))
    vol = torch.rand(shape, generator=g, device="cuda")
    lab = torch.randint(0, C, (shape[0],) + shape[2:], generator=g, device="cuda").to(torch.int64 if kind == "unet3d" else torch.uint8)
    noise = torch.zeros((1,) + shape[1:], device="cuda")                 # injected: no device RNG in the comparison
    model.dropout_enabled = ema.dropout_enabled = False
    for _ in range(steps):
        if kind == "cross":
            out1, _ = tr.step(vol, lab)
        else:
            # taped: the teacher noise comes from the device RNG (an injected tensor keeps the step eager); the same seed on both
            # ranks, so only the shards differ
            tr.step(vol, lab, noise=None if taped else noise)
    torch.cuda.synchronize()
    if taped:
        # steps 4.. were replays of the recorded launch sequence, the bucketer's collectives among its entries
        assert mstep.STEP_TAPE and tr._tape is not None and len(tr._tape) > 50
    loss = float(out1[0]) if kind == "cross" else tr.losses()["loss"]
    torch.save(dict(student=model.flat_param.cpu(), teacher=ema.flat_param.cpu(), loss=loss),
               os.path.join(out_dir, f"{kind}_{int(overlap)}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["unet3d", "unet2d", "cross"])
def test_two_ranks_on_one_gpu_exchange_gradients(tmp_path, kind):
    world = 2
    for overlap in (0, 1):
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), overlap, kind), nprocs=world, join=True)
    r = {(o, k): torch.load(os.path.join(tmp_path, f"{kind}_{o}_{k}.pt")) for o in (0, 1) for k in (0, 1)}
    for o in (0, 1):
        assert torch.equal(r[(o, 0)]["student"], r[(o, 1)]["student"]), "ranks diverged"
        assert torch.equal(r[(o, 0)]["teacher"], r[(o, 1)]["teacher"])
        assert r[(o, 0)]["loss"] != r[(o, 1)]["loss"]                    # the shards really differ
    # overlapped, bucketed exchange == one blocking all-reduce
    assert torch.equal(r[(0, 0)]["student"], r[(1, 0)]["student"])
    assert torch.equal(r[(0, 0)]["teacher"], r[(1, 0)]["teacher"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["unet2d", "cross"])
def test_two_ranks_replay_the_launch_tape_with_the_exchange_inside(tmp_path, kind):
    """Six steps on two ranks (gloo): the third is recorded, steps 4 - 6 are REPLAYS of the launch tape whose entries include the
    bucketer's advance / finish calls -- the collectives are issued from the tape in the recorded order on both ranks.  Ranks end
    with identical weights, and the overlapped bucketed exchange equals the blocking all-reduce bit for bit, as in the eager test."""
    world = 2
    for overlap in (0, 1):
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), overlap, kind, 6, True), nprocs=world, join=True)
    r = {(o, k): torch.load(os.path.join(tmp_path, f"{kind}_{o}_{k}.pt")) for o in (0, 1) for k in (0, 1)}
    for o in (0, 1):
        assert torch.equal(r[(o, 0)]["student"], r[(o, 1)]["student"]), "ranks diverged"
        assert torch.equal(r[(o, 0)]["teacher"], r[(o, 1)]["teacher"])
        assert r[(o, 0)]["loss"] != r[(o, 1)]["loss"]
    assert torch.equal(r[(0, 0)]["student"], r[(1, 0)]["student"])
    assert torch.equal(r[(0, 0)]["teacher"], r[(1, 0)]["teacher"])


def _rccl_worker(rank, port, out_dir, bucketed, kind):
    """ONE rank, backend nccl (= RCCL) on the one GPU: ProcessGroupNCCL's own stream, the event hand-over of
    ``async_op=True`` work objects, ``work.wait()`` on the compute stream and collectives enqueued from a callback that
    runs behind a SIDE stream -- none of which gloo exercises (it stages through the host after a stream sync)."""
    import datetime
    os.environ["MIS_GRAD_OVERLAP"] = "1"
    os.environ["MIS_FORCE_BUCKETER"] = "1" if bucketed else "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    from mis_hip import plan as mplan, step as mstep
    from mis_hip.dist import GradBucketer
    from mis_hip.step import CnnMeetVitTrainer, CrossTeachingTrainer, MeanTeacherTrainer, UAMTTrainer
    from test_grad_progress_gpu import _make
    assert mstep.TWO_STREAM and mplan.WGRAD_STREAM          # both side streams on (the product configuration)
    torch.manual_seed(5)
    C = 2 if kind in ("unet3d", "uamt3d") else 4
    pg = dist.group.WORLD
    nets = []
    if kind in ("cross", "cnnvit"):
        m1, m2 = _make("unet2d", C), _make("swin", C)
        nets = [m1, m2]
        if kind == "cnnvit":
            t = _make("swin", C)
            t.load_state_dict(m2.state_dict())
            nets.append(t)
            tr = CnnMeetVitTrainer(m1, m2, t, labeled_bs=1, num_classes=C, iter_num=1200, seed=7, process_group=pg)
        else:
            tr = CrossTeachingTrainer(m1, m2, labeled_bs=1, num_classes=C, iter_num=1000, seed=7, process_group=pg)
        assert (tr._bucketers[0] is not None) == bool(bucketed)
        if bucketed:
            tr._bucketers = (GradBucketer(m1.flat_grad, pg, bucket_bytes=1 << 20, defer_tail=tr._bucketers[0].defer_tail),
                             GradBucketer(m2.flat_grad, pg, bucket_bytes=8 << 20, defer_tail=tr._bucketers[1].defer_tail))
    else:
        base = "unet3d" if kind == "uamt3d" else kind
        m, e = _make(base, C), _make(base, C)
        e.load_state_dict(m.state_dict())
        nets = [m, e]
        cls = UAMTTrainer if kind == "uamt3d" else MeanTeacherTrainer
        tr = cls(m, e, labeled_bs=1, num_classes=C, iter_num=1000, seed=7, process_group=pg)
        assert (tr._bucketer is not None) == bool(bucketed)
        if bucketed:
            tr._bucketer = GradBucketer(m.flat_grad, pg, bucket_bytes=1 << 20)
            assert len(tr._bucketer.buckets) >= 3
    for n in nets:
        n.train()
        n.dropout_enabled = False
    g = torch.Generator(device="cuda").manual_seed(100)
    shape = {"unet2d": (2, 1, 64, 64), "cross": (2, 1, 224, 224), "cnnvit": (2, 1, 224, 224)}.get(kind, (2, 1, 32, 32, 32))
    vol = torch.rand(shape, generator=g, device="cuda")
    lab = torch.randint(0, C, (shape[0],) + shape[2:], generator=g, device="cuda").to(
        torch.int64 if C == 2 else torch.uint8)
    noise = torch.zeros((1,) + shape[1:], device="cuda")
    for _ in range(3):
        if kind == "cross":
            tr.step(vol, lab)
        elif kind == "uamt3d":
            tr.step(vol, lab, noise=noise, mc_noise=[torch.zeros((2,) + shape[1:], device="cuda")] * 4)
        else:
            tr.step(vol, lab, noise=noise)
    torch.cuda.synchronize()
    torch.save([n.flat_param.cpu() for n in nets], os.path.join(out_dir, f"rccl_{kind}_{int(bucketed)}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["unet3d", "unet2d", "cross", "cnnvit", "uamt3d"])
def test_single_rank_rccl_group_bucketed_step_is_bit_identical(tmp_path, kind):
    """The step with a real RCCL process group (one rank) and the gradient bucketer forced on -- small buckets, teacher /
    second student and weight gradients on their side streams -- leaves bit for bit the weights of the plain
    single-GPU step after three iterations."""
    for bucketed in (0, 1):
        mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path), bucketed, kind), nprocs=1, join=True)
    a = torch.load(os.path.join(tmp_path, f"rccl_{kind}_0.pt"))
    b = torch.load(os.path.join(tmp_path, f"rccl_{kind}_1.pt"))
    assert len(a) == len(b) >= 2
    for x, y in zip(a, b):
        assert torch.isfinite(x).all() and torch.equal(x, y)


@pytest.mark.timeout(1200)
def test_bench_multi_gpu_code_path_on_a_single_rank_rccl_group():
    """``bench.py``'s N > 1 path -- RCCL group with a timeout, weight broadcast, barrier + all-gathered timings, the bit
    fingerprint of the weights, the blocking / no-exchange regions, cross teaching behind the default workload -- executed in
    a process group of one rank on the one GPU (MIS_BENCH_DIST_AT_WORLD1), gradient bucketers forced on: what the driver's
    2 / 4 / 8-GPU launches run, minus the peers."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MIS_BENCH_DIST_AT_WORLD1="1", MIS_FORCE_BUCKETER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-kernel-events"], capture_output=True, text=True, timeout=1100, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    d = out["distributed"]
    assert out["n_gpus"] == 1 and out["value"] > 0 and d["backend"] == "nccl (RCCL)" and d["params_identical"] is True
    assert d["blocking_allreduce_ms_per_step"] > 0 and d["no_exchange_ms_per_step"] > 0
    assert all(v == v for v in out["losses_last_step"].values())            # finite losses after the extra regions
    c = out["others"]["cross"]
    assert c["value"] > 0 and c["distributed"]["params_identical"] is True and c["distributed"]["blocking_allreduce_ms_per_step"] > 0
    assert "cpu_baseline" not in out

"""Data-parallel semantics of the step on CPU with gloo, world_size = 2.

The HIP kernels cannot run here, so the compute of each rank is the CPU oracle; what is under test is
the product's host-side exchange (``mis_hip.dist``: one all-reduce of the flat gradient bucket, the
1/world scale handed to the optimizer, state broadcast, index sharding) and the claim that the
resulting update equals "reference step on each shard, then average the gradients".
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_inputs(rank):
    from oracle import filler
    vol = filler.image((2, 1, 32, 32), f"volume_r{rank}")
    lab = filler.labels((2, 32, 32), 4, torch.uint8)
    noise = filler.noise((1, 1, 32, 32), f"noise_r{rank}")
    return vol, lab, noise


def _flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from mis_hip import dist as mdist
    from oracle import filler
    from oracle.nets import OracleUNet2D
    from oracle.step import mean_teacher_step
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    onet = OracleUNet2D(1, 4)
    # rank 1 starts from different weights on purpose: broadcast_state must fix that
    student = filler.fill_state_dict({(f"r{rank}." if rank else "") + k: v for k, v in onet.new_state().items()})
    student = {k.split(".", 1)[1] if rank else k: v for k, v in student.items()}
    pnames = [n for n in student if onet.is_param(n)]
    flat_p = _flat([student[n] for n in pnames])
    mdist.broadcast_state([flat_p])
    off = 0
    for n in pnames:
        k = student[n].numel()
        student[n] = flat_p[off:off + k].view_as(student[n]).clone()
        off += k
    teacher = {k: v.clone() for k, v in student.items()}
    vol, lab, noise = _shard_inputs(rank)

    def hook(grads):                      # what MeanTeacherTrainer does between backward and SGD+EMA
        flat_g = _flat([grads[n] for n in pnames])
        scale = mdist.sync_gradients(flat_g)
        assert scale == 1.0 / world
        flat_g.mul_(scale)                # the HIP kernel folds this factor into the update
        out, o = {}, 0
        for n in pnames:
            k = grads[n].numel()
            out[n] = flat_g[o:o + k].view_as(grads[n])
            o += k
        return out

    r = mean_teacher_step(onet, student, teacher, {}, vol, lab, noise, 1200, labeled_bs=1, num_classes=4,
                          drop_student="off", drop_teacher="off", grad_hook=hook)
    torch.save(dict(student={n: student[n] for n in pnames}, teacher={n: teacher[n] for n in pnames},
                    loss=r["loss"]), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_equals_shardwise_reference_with_averaged_grads(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    # (1) every rank holds the same student and teacher after the step
    for n in res[0]["student"]:
        assert torch.equal(res[0]["student"][n], res[1]["student"][n]), n
        assert torch.equal(res[0]["teacher"][n], res[1]["teacher"][n]), n
    # (2) equals: oracle step on each shard from rank 0's weights, gradients averaged, one update
    from oracle import filler
    from oracle.nets import OracleUNet2D
    from oracle.step import mean_teacher_step
    onet = OracleUNet2D(1, 4)
    sd0 = filler.fill_state_dict(onet.new_state())
    pnames = [n for n in sd0 if onet.is_param(n)]
    grads = []
    for rank in range(world):
        vol, lab, noise = _shard_inputs(rank)
        st = {k: v.clone() for k, v in sd0.items()}
        te = {k: v.clone() for k, v in sd0.items()}
        r = mean_teacher_step(onet, st, te, {}, vol, lab, noise, 1200, labeled_bs=1, num_classes=4,
                              drop_student="off", drop_teacher="off", apply_update=False)
        grads.append(r["grads"])
        assert abs(r["loss"] - res[rank]["loss"]) < 1e-6
    avg = {n: (grads[0][n] + grads[1][n]) / 2 for n in pnames}
    st = {k: v.clone() for k, v in sd0.items()}
    te = {k: v.clone() for k, v in sd0.items()}
    vol, lab, noise = _shard_inputs(0)
    mean_teacher_step(onet, st, te, {}, vol, lab, noise, 1200, labeled_bs=1, num_classes=4, drop_student="off",
                      drop_teacher="off", grad_hook=lambda g: avg)
    for n in pnames:
        assert torch.allclose(st[n], res[0]["student"][n], rtol=0, atol=1e-7), n
        assert torch.allclose(te[n], res[0]["teacher"][n], rtol=0, atol=1e-7), n


def test_shard_indices_are_disjoint_and_cover():
    sys.path.insert(0, os.path.join(ROOT, "cv-ssl-mis_amd"))
    from mis_hip.dist import shard_indices, world_size
    lab, unl = list(range(0, 25)), list(range(25, 250))
    seen_l, seen_u = [], []
    for r in range(8):
        a, b = shard_indices(lab, unl, r, 8)
        seen_l += a
        seen_u += b
    assert sorted(seen_l) == lab and sorted(seen_u) == unl
    assert world_size() == 1


class _FakeOp:
    """Stands in for a plan op: holds parameter references (``.data`` / ``.grad`` views into the flat buffers) and
    writes their gradients in ``bwd`` -- what mis_hip.plan ops do on the device."""

    def __init__(self, refs, src):
        for i, r in enumerate(refs):
            setattr(self, f"p{i}", r)
        self.refs, self.src = refs, src
        self.activation = torch.zeros(3)         # non-parameter attributes must be ignored by param_progress

    def bwd(self):
        for r in self.refs:
            r.grad.copy_(r.true_grad)


class _Ref:
    def __init__(self, data, grad, true_grad):
        self.data, self.grad, self.true_grad = data, grad, true_grad


def _bucket_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from mis_hip import dist as mdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    sizes = [300, 7, 4096, 33, 9000, 2, 512, 12000, 64]          # parameter tensors in forward order
    total = sum((n + 3) // 4 * 4 for n in sizes)
    flat_p, flat_g = torch.zeros(total), torch.zeros(total)
    true = torch.randn(total, generator=g)
    refs, off = [], 0
    for n in sizes:
        refs.append(_Ref(flat_p[off:off + n], flat_g[off:off + n], true[off:off + n]))
        off += (n + 3) // 4 * 4
    # ops in forward order; op 1 has no parameters, op 3 owns two tensors
    ops = [_FakeOp(refs[0:2], flat_g), _FakeOp([], flat_g), _FakeOp(refs[2:3], flat_g), _FakeOp(refs[3:5], flat_g),
           _FakeOp(refs[5:7], flat_g), _FakeOp(refs[7:9], flat_g)]
    done = mdist.param_progress(ops, flat_g)
    assert done[0] == 0 and done == sorted(done) and done[-1] == refs[6].grad.data_ptr() // 4 - flat_g.data_ptr() // 4 + 512
    b = mdist.GradBucketer(flat_g, bucket_bytes=16384)           # 4096-float buckets -> several collectives
    assert b.buckets[0][1] == total and b.buckets[-1][0] == 0 and len(b.buckets) >= 5
    assert all(b.buckets[i][0] == b.buckets[i + 1][1] for i in range(len(b.buckets) - 1))
    issued = []
    b.begin()
    for i in range(len(ops) - 1, -1, -1):                        # Plan.backward(on_progress=b.advance)
        ops[i].bwd()
        b.advance(done[i])
        issued.append(b._next)
    scale = b.finish()
    assert scale == 1.0 / world and b._next == len(b.buckets)
    assert issued[0] > 0 and issued == sorted(issued)            # buckets left while the "backward" was still running
    ref = true.clone()
    pad = torch.ones(total, dtype=torch.bool)
    for r in refs:
        o = r.grad.data_ptr() // 4 - flat_g.data_ptr() // 4
        pad[o:o + r.grad.numel()] = False
    ref[pad] = 0
    dist.all_reduce(ref)                                         # the single blocking all-reduce it replaces
    assert torch.equal(flat_g, ref)
    torch.save(flat_g, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_overlapped_allreduce_equals_single_allreduce(tmp_path):
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(world))
    assert torch.equal(a, b)


# ---- the real plans' gradient ordering, recorded on the GPU (scripts/record_grad_ranges.py), through the real
# ---- param_progress + GradBucketer over gloo
class _RecordedOp:
    """An op of a recorded plan: ``owns`` = the parameter-gradient views it holds as attributes (what param_progress
    scans), ``writes`` = the ranges of the flat gradient buffer its backward actually wrote on the GPU."""

    def __init__(self, flat_p, flat_g, true, owns, writes):
        for i, (lo, hi) in enumerate(owns):
            setattr(self, f"p{i}", _Ref(flat_p[lo:hi], flat_g[lo:hi], None))
        self.flat_g, self.true, self.writes = flat_g, true, writes

    def bwd(self):
        for lo, hi in self.writes:
            self.flat_g[lo:hi] = self.true[lo:hi]


def _recorded_worker(rank, world, port, out_dir):
    import json
    for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from mis_hip import dist as mdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    with open(os.path.join(ROOT, "tests", "golden", "plan_grad_ranges.json")) as f:
        rec = json.load(f)
    for kind, r in sorted(rec.items()):
        total = r["total"]
        g = torch.Generator().manual_seed(1000 + rank)
        flat_p, flat_g = torch.zeros(total), torch.zeros(total)
        true = torch.randn(total, generator=g)
        ops = [_RecordedOp(flat_p, flat_g, true, o["owns"], o["writes"]) for o in r["ops"]]
        done = mdist.param_progress(ops, flat_g)
        assert done == sorted(done) and done[0] == 0, kind
        b = mdist.GradBucketer(flat_g, bucket_bytes=1 << 20)        # 1 MiB: many buckets even for the small nets
        b.begin()
        shipped_at = []
        for i in range(len(ops) - 1, -1, -1):                       # Plan.backward(on_progress=b.advance)
            ops[i].bwd()
            if i == 0 or done[i] != done[i - 1] or i == len(ops) - 1:
                b.advance(done[i])
            shipped_at.append(b._next)
        b.finish()
        written = torch.zeros(total, dtype=torch.bool)
        for o in r["ops"]:
            for lo, hi in o["writes"]:
                written[lo:hi] = True
        ref = torch.where(written, true, torch.zeros(()))
        dist.all_reduce(ref)                                        # the single blocking all-reduce of the final buffer
        assert torch.equal(flat_g, ref), f"{kind}: a bucket left before one of its gradients was written"
        assert shipped_at[len(shipped_at) // 2] > 0, f"{kind}: nothing was exchanged during the backward"
        if rank == 0:
            with open(os.path.join(out_dir, f"{kind}.ok"), "w") as f:
                f.write(f"{len(ops)} ops, {len(b.buckets)} buckets\n")
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_recorded_plan_backward_order_through_bucketer(tmp_path):
    """Every op's owned and actually written gradient ranges of the real unet / unet_3D / V-Net (GroupNorm) / SwinUnet
    plans: the bucketed, overlapped exchange equals the blocking all-reduce of the finished buffer."""
    world = 2
    mp.spawn(_recorded_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["swin.ok", "unet2d.ok", "unet3d.ok", "vnet_groupnorm.ok"]


def _bucket_layout_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT):
        sys.path.insert(0, p)
    import torch.distributed as dist
    from mis_hip.dist import GradBucketer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {}
    # (1) two models whose collectives interleave: A's backward is enqueued first with its tail deferred (the side-stream
    # student of cross teaching / CNN-meets-ViT), B's second; sums must equal one all-reduce of each buffer
    g = torch.Generator().manual_seed(10 + rank)
    a, b = torch.randn(700_000, generator=g), torch.randn(300_000, generator=g)
    ref_a, ref_b = a.clone(), b.clone()
    dist.all_reduce(ref_a)
    dist.all_reduce(ref_b)
    ba = GradBucketer(a, None, bucket_bytes=1 << 20, defer_tail=True)
    bb = GradBucketer(b, None, bucket_bytes=1 << 19)
    ba.begin(); bb.begin()
    for lo in (600_000, 300_000, 5, 0):
        ba.advance(lo)
    issued_before_finish = ba._next
    assert ba.buckets[-1][0] == 0 and issued_before_finish == len(ba.buckets) - 1      # the offset-0 bucket is held back
    assert not ba.would_issue(0)
    for lo in (200_000, 0):
        bb.advance(lo)
    assert bb._next == len(bb.buckets)                                                   # B's tail went with its backward
    ba.finish(); bb.finish()
    res["sum_ok"] = bool(torch.equal(a, ref_a) and torch.equal(b, ref_b))
    # (2) ranks that disagree on MIS_BUCKET_MB must fail loudly at construction, not hang in mismatched collectives
    os.environ["MIS_BUCKET_MB"] = "1" if rank == 0 else "2"
    try:
        GradBucketer(torch.zeros(1 << 20), None)
        res["mismatch_raised"] = False
    except RuntimeError as e:
        res["mismatch_raised"] = "disagree" in str(e)
    del os.environ["MIS_BUCKET_MB"]
    torch.save(res, os.path.join(out_dir, f"layout{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_deferred_tail_bucket_and_rank_agreement_on_the_layout(tmp_path):
    world = 2
    mp.spawn(_bucket_layout_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"layout{r}.pt"))
        assert res["sum_ok"] and res["mismatch_raised"], res


def test_bucket_layout_and_exchange_model():
    """bucket_layout is what GradBucketer cuts (a pure function of the sizes); predict_exchange prices those buckets."""
    from mis_hip import dist as mdist
    n = (100 << 20) // 4
    lay = mdist.bucket_layout(n, 4, mdist.default_bucket_bytes(n * 4))
    assert lay[0][1] == n and lay[-1] == (0, (2 << 20) // 4)                 # issued from the end; 2 MiB exposed tail
    assert all(a[0] == b[1] for a, b in zip(lay, lay[1:]))                   # contiguous, descending
    flat = torch.zeros(n // 64)                                              # small buffer: one bucket
    b = mdist.GradBucketer(flat, bucket_bytes=1 << 30)
    assert b.buckets == mdist.bucket_layout(flat.numel(), 4, 1 << 30) == [(0, flat.numel())]
    p = mdist.predict_exchange([23536208, 108_000_000], 8)
    assert p["world"] == 8 and len(p["models"]) == 2
    m0 = p["models"][0]
    assert sum(m0["bucket_bytes"]) == 23536208 and m0["bucket_bytes"][-1] == 2 << 20
    f = 2 * 7 / 8
    assert abs(m0["ring_ms"][0] - (f * m0["bucket_bytes"][0] / 153e9 * 1e3 + 14 * 25e-3)) < 1e-3
    assert abs(m0["direct_ms"][0] - (f * m0["bucket_bytes"][0] / (7 * 153e9) * 1e3 + 2 * 25e-3)) < 1e-3
    assert p["exposed_tail_ring_ms"] < p["total_ring_ms"] and p["total_direct_ms"] < p["total_ring_ms"]
    assert mdist.predict_exchange([1000], 1)["models"] == []


def test_rank_pinning_follows_the_gpus_numa_node(tmp_path):
    """pin_rank_to_numa against a fabricated sysfs: ranks whose GPUs share a NUMA node split its cores in rank order,
    shares are disjoint, an unknown node falls back to an even split of the allowed cores."""
    from mis_hip import dist as mdist
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("needs >= 4 usable cores")
    half = len(allowed) // 2
    nodes = {0: allowed[:half], 1: allowed[half:]}
    for k, cpus in nodes.items():
        d = tmp_path / "devices/system/node" / f"node{k}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(map(str, cpus)) + "\n")
    ids = ["0000:05:00.0", "0000:15:00.0", "0000:85:00.0", "0000:95:00.0"]
    for bid, node in zip(ids, (0, 0, 1, 1)):
        d = tmp_path / "bus/pci/devices" / bid
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    infos = [mdist.pin_rank_to_numa(r, 4, ids, sysfs=str(tmp_path), apply=False) for r in range(4)]
    assert [i["numa_node"] for i in infos] == [0, 0, 1, 1] and all(i["sharers"] == 2 and not i["applied"] for i in infos)

    def cpus_of(i):
        return set(mdist._parse_cpulist(i["cpus"]))
    sets = [cpus_of(i) for i in infos]
    assert sets[0] | sets[1] <= set(nodes[0]) and sets[2] | sets[3] <= set(nodes[1])
    assert all(not (sets[a] & sets[b]) for a in range(4) for b in range(a + 1, 4))
    unknown = [mdist.pin_rank_to_numa(r, 2, None, sysfs=str(tmp_path), apply=False) for r in range(2)]
    assert all(i["numa_node"] == -1 for i in unknown) and not (cpus_of(unknown[0]) & cpus_of(unknown[1]))
    before, threads, omp = os.sched_getaffinity(0), torch.get_num_threads(), os.environ.get("OMP_NUM_THREADS")
    try:
        i = mdist.pin_rank_to_numa(1, 4, ids, sysfs=str(tmp_path), apply=True)
        assert i["applied"] and os.sched_getaffinity(0) == cpus_of(i)
        # every thread of the process carries the mask, and torch's intra-op pool is sized to the share
        assert all(os.sched_getaffinity(int(t)) == cpus_of(i) for t in os.listdir("/proc/self/task"))
        assert torch.get_num_threads() == i["n_cpus"] == i["threads"] and os.environ["OMP_NUM_THREADS"] == str(i["n_cpus"])
    finally:
        for t in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(t), before)
            except OSError:
                pass
        torch.set_num_threads(threads)
        if omp is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = omp

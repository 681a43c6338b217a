"""Data-parallel plumbing of the Mean-Teacher step: one process per GPU, RCCL over xGMI.

The reference is single-process/single-GPU (``DistributedDataParallel`` is imported but never used,
SURVEY.md s.0 item 7); its ``--batch_size`` / ``--labeled_bs`` are already "per gpu".  Here every rank
owns its own labeled+unlabeled shard and the ONLY exchange of a step is one all-reduce (sum) of the
flat fp32 gradient bucket of the student (7.3 MB UNet ... 23.5 MB unet_3D); the 1/world averaging is
folded into the fused SGD+EMA kernel (``grad_scale``), the teacher needs no communication because
every rank applies the identical update.  Semantics = standard DDP: per-rank loss (Dice is a ratio
of per-rank sums), per-rank BatchNorm statistics (the reference has no SyncBN).

``backend``: "nccl" is RCCL on ROCm; the CPU tests drive the same functions over "gloo".
"""
import os

import torch
import torch.distributed as dist


def initialized():
    return dist.is_available() and dist.is_initialized()


def world_size(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def sync_gradients(flat_grad, group=None):
    """All-reduce (sum) the flat gradient bucket in place; returns the scale (1/world) the optimizer
    kernel must apply.  No-op for a single process."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w


class GradBucketer:
    """Overlap of the gradient exchange with the backward pass.

    The flat gradient buffer is laid out in parameter (= forward) order and the backward pass finishes it from the
    END towards the start, so the finished part is always a suffix.  ``advance(lo)`` -- called by ``Plan.backward``
    after every op with the start of the finished suffix -- issues an asynchronous all-reduce (sum) for every bucket
    that now lies completely inside it; RCCL runs it on the process group's own stream while the remaining backward
    kernels keep the compute stream busy.  ``finish()`` issues what is left and makes the current stream wait for all
    of them (no host synchronisation).  The reduction is element-wise, so the result is identical to one all-reduce
    of the whole buffer; every rank cuts the same buckets (a pure function of the buffer size).

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound and needs
    messages of several MB to reach its bandwidth, and every collective costs the LAUNCHING thread ~0.1 ms of host time
    (c10d work object + events) during which no kernel is enqueued: SwinUnet's 700 launches per 28 ms step leave no
    slack for 13 of them (8 MiB buckets: +0.6 ms per step, scripts/ddp_overhead.py; 32 MiB: +0.0).  Buckets are
    max(16 MiB, a quarter of the buffer): 2 (unet_3D, 23.5 MB) to 4 (SwinUnet, 108.7 MB; UNETR, 371 MB) collectives per
    step plus the small tail bucket below.  MIS_BUCKET_MB overrides."""

    def __init__(self, flat_grad, group=None, bucket_bytes=None, defer_tail=False):
        if bucket_bytes is None:
            if "MIS_BUCKET_MB" in os.environ:
                bucket_bytes = int(float(os.environ["MIS_BUCKET_MB"]) * (1 << 20))
            else:       # at most ~4 collectives (+ the small tail) per model and step
                bucket_bytes = max(16 << 20, -(-flat_grad.numel() * flat_grad.element_size() // 4))
        bucket_bytes = int(bucket_bytes)
        if world_size(group) > 1:
            # the cut points must be the same on every rank (MIS_BUCKET_MB is a per-process environment variable): ranks
            # that disagree would issue collectives of different sizes and counts -- a hang or silently wrong sums
            probe = torch.tensor([bucket_bytes, -bucket_bytes, flat_grad.numel(), -flat_grad.numel()],
                                 dtype=torch.int64, device=flat_grad.device)
            dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
            hi_b, lo_b, hi_n, lo_n = (int(v) for v in probe.cpu())
            if hi_b != -lo_b or hi_n != -lo_n:
                raise RuntimeError(f"GradBucketer: ranks disagree on the bucket layout (bucket bytes {-lo_b}..{hi_b}, "
                                   f"gradient elements {-lo_n}..{hi_n}); set MIS_BUCKET_MB identically on every rank")
        self.flat, self.group = flat_grad, group
        # defer_tail: ``advance`` never issues the bucket at offset 0 (complete only when the backward is); ``finish``
        # does.  For a backward whose collectives are ENQUEUED before another network's (cross teaching: the side-stream
        # student): the in-order RCCL stream would otherwise hold the other network's early buckets behind this tail
        self.defer_tail = bool(defer_tail)
        n, per = flat_grad.numel(), max(1, bucket_bytes // flat_grad.element_size())
        # cut from the end (the part that finishes first); the first bucket takes the remainder
        cuts = list(range(n, 0, -per)) + [0]
        self.buckets = [(cuts[i + 1], cuts[i]) for i in range(len(cuts) - 1)]     # (lo, hi), descending
        # the bucket at offset 0 is complete only when the backward is: its all-reduce is the exposed one.  Keep it small
        # (2 MiB: a latency-bound ring step) by splitting the remainder
        tail = max(1, (2 << 20) // flat_grad.element_size())
        lo, hi = self.buckets[-1]
        if hi - lo > 2 * tail:
            self.buckets[-1:] = [(lo + tail, hi), (lo, lo + tail)]
        self._next, self._works = 0, []

    def begin(self):
        self._next, self._works = 0, []

    def _ready(self, lo, final=False):
        if self._next >= len(self.buckets):
            return False
        b_lo = self.buckets[self._next][0]
        return b_lo >= lo and (final or not (self.defer_tail and b_lo == 0))

    def would_issue(self, lo):
        """True when ``advance(lo)`` would start an all-reduce (Plan.backward skips the stream hand-off otherwise)."""
        return self._ready(lo)

    def advance(self, lo, final=False):
        """All gradient elements at offsets >= ``lo`` are final (on the CURRENT stream: the all-reduce orders itself
        behind it)."""
        while self._ready(lo, final):
            b_lo, b_hi = self.buckets[self._next]
            self._works.append(dist.all_reduce(self.flat[b_lo:b_hi], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            self._next += 1

    def finish(self):
        self.advance(0, final=True)
        for w in self._works:
            w.wait()
        self._works = []
        return 1.0 / world_size(self.group)


def param_progress(ops, flat_grad):
    """For a plan's op list (forward order): ``done[i]`` = start of the finished suffix of ``flat_grad`` once the
    backward pass has executed ops ``i, i+1, ...`` (it runs them in reverse), i.e. the highest end offset of any
    parameter gradient written by the ops still to run (``0`` when none is left).  An op writes the gradients of the
    parameter references it holds as attributes (objects with ``.grad`` views into ``flat_grad``)."""
    base, esz, total = flat_grad.data_ptr(), flat_grad.element_size(), flat_grad.numel()
    ends = []
    for op in ops:
        hi = 0
        for v in vars(op).values():
            g = getattr(v, "grad", None)
            if isinstance(g, torch.Tensor) and hasattr(v, "data") and g.numel() and \
                    base <= g.data_ptr() < base + total * esz:
                hi = max(hi, (g.data_ptr() - base) // esz + g.numel())
        ends.append(hi)
    done, run = [], 0
    for hi in ends:              # prefix maximum over the ops BEFORE i
        done.append(run)
        run = max(run, hi)
    return done


def broadcast_state(flat_tensors, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters / buffers."""
    if world_size(group) > 1:
        for t in flat_tensors:
            dist.broadcast(t, src, group=group)


def shard_indices(labeled_idxs, unlabeled_idxs, rank, world):
    """Disjoint per-rank index shards for the two-stream sampler (labeled and unlabeled pools are
    split round-robin, so every rank keeps the reference's labeled:unlabeled ratio)."""
    return list(labeled_idxs)[rank::world], list(unlabeled_idxs)[rank::world]

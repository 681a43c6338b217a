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


def default_bucket_bytes(grad_bytes, bucket_bytes=None):
    """max(16 MiB, a quarter of the buffer) unless given / MIS_BUCKET_MB: at most ~4 collectives (+ the small tail) per
    model and step."""
    if bucket_bytes is None:
        if "MIS_BUCKET_MB" in os.environ:
            bucket_bytes = int(float(os.environ["MIS_BUCKET_MB"]) * (1 << 20))
        else:
            bucket_bytes = max(16 << 20, -(-int(grad_bytes) // 4))
    return int(bucket_bytes)


def bucket_layout(numel, esize, bucket_bytes):
    """(lo, hi) element ranges of the all-reduce buckets, in the order they are issued (descending offsets): cut from
    the END of the flat buffer (the part the backward finishes first), the first bucket takes the remainder; the bucket
    at offset 0 is complete only when the backward is -- its all-reduce is the exposed one -- so it is kept small (2 MiB:
    a latency-bound ring step) by splitting the remainder.  A pure function of the sizes: every rank cuts the same."""
    n, per = int(numel), max(1, int(bucket_bytes) // int(esize))
    cuts = list(range(n, 0, -per)) + [0]
    buckets = [(cuts[i + 1], cuts[i]) for i in range(len(cuts) - 1)]
    tail = max(1, (2 << 20) // int(esize))
    lo, hi = buckets[-1]
    if hi - lo > 2 * tail:
        buckets[-1:] = [(lo + tail, hi), (lo, lo + tail)]
    return buckets


XGMI_LINK_GBPS = 153.0        # per direction-pair and link, 7 links per GPU (MI355X_MICROARCH.md / the task's figure)


def predict_exchange(grad_bytes, world, bucket_bytes=None, link_gbps=XGMI_LINK_GBPS, latency_us=25.0):
    """A MODEL (not a measurement) of the gradient exchange of one step at ``world`` GPUs of one node, to read the
    driver's scaling numbers against.  ``grad_bytes``: flat gradient bytes of every model that is exchanged.  An
    all-reduce of B bytes moves 2 (N-1)/N B per GPU; xGMI is a point-to-point mesh (N-1 peers, one ~153 GB/s link each):
    ``ring`` = one ring, every hop on ONE link (per-link bound); ``direct`` = the N-1 peers' links used at once
    (reduce-scatter + all-gather straight to the peers).  Each collective adds ``latency_us`` x 2 (N-1) ring steps
    (``ring``) or x 2 (``direct``).  ``exposed``: only the 2 MiB tail bucket of every model cannot hide behind the
    backward; everything else overlaps if total <= the backward's duration."""
    world = int(world)
    out = dict(world=world, link_GBps=link_gbps, latency_us_per_step=latency_us, models=[],
               model="all-reduce bytes per GPU = 2 (N-1)/N x bucket; ring: / one link; direct: / (N-1) links; + latency per "
                     "ring step; a model, not a measurement")
    if world < 2:
        return out
    f = 2.0 * (world - 1) / world
    tot_ring = tot_direct = exp_ring = exp_direct = 0.0
    for gb in grad_bytes:
        lay = bucket_layout(int(gb) // 4, 4, default_bucket_bytes(gb, bucket_bytes))
        sizes = [(hi - lo) * 4 for lo, hi in lay]
        ring = [f * b / (link_gbps * 1e9) * 1e3 + 2 * (world - 1) * latency_us * 1e-3 for b in sizes]
        direct = [f * b / ((world - 1) * link_gbps * 1e9) * 1e3 + 2 * latency_us * 1e-3 for b in sizes]
        out["models"].append(dict(grad_bytes=int(gb), bucket_bytes=sizes, ring_ms=[round(t, 4) for t in ring],
                                  direct_ms=[round(t, 4) for t in direct]))
        tot_ring, tot_direct = tot_ring + sum(ring), tot_direct + sum(direct)
        exp_ring, exp_direct = exp_ring + ring[-1], exp_direct + direct[-1]
    out.update(total_ring_ms=round(tot_ring, 4), total_direct_ms=round(tot_direct, 4),
               exposed_tail_ring_ms=round(exp_ring, 4), exposed_tail_direct_ms=round(exp_direct, 4))
    return out


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def _numa_node_of(bus_id, sysfs):
    if not bus_id:
        return -1
    bid = bus_id.lower()
    if len(bid.split(":")) == 2:
        bid = "0000:" + bid
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", bid, "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def device_bus_ids(n):
    """PCI bus ids ("dddd:bb:dd.0") of HIP devices 0..n-1 as torch reports them, None where it does not."""
    out = []
    for i in range(n):
        try:
            p = torch.cuda.get_device_properties(i)
            out.append("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
        except Exception:
            out.append(None)
    return out


def pin_rank_to_numa(local_rank, local_world, pci_bus_ids=None, sysfs="/sys", apply=True):
    """CPU affinity of this rank := its share of the cores of the NUMA node its GPU hangs off (one process per GPU: the
    launch thread, the c10d watchdog and the input pipeline stay next to the device's PCIe root; 8 Python processes
    enqueue 300-700 launches per step each).  ``pci_bus_ids[i]`` = bus id of local rank i's GPU (``device_bus_ids``);
    the node comes from ``<sysfs>/bus/pci/devices/<id>/numa_node``, its cores from
    ``<sysfs>/devices/system/node/node<k>/cpulist``.  Ranks whose GPUs share a node split its cores evenly in local-rank
    order -- every rank evaluates every rank's node, so the shares are disjoint without any communication.  Unknown node
    (-1 / no sysfs entry / no bus id): the cores this process may run on are split evenly over ``local_world``.
    MIS_PIN_NUMA=0 leaves the affinity alone.  Returns a dict for the log; never raises."""
    local_rank, local_world = int(local_rank), max(1, int(local_world))
    info = dict(local_rank=local_rank, local_world=local_world, numa_node=None, cpus=None, n_cpus=None, applied=False)
    try:
        allowed = sorted(os.sched_getaffinity(0))
        ids = list(pci_bus_ids) if pci_bus_ids else []
        ids += [None] * (local_world - len(ids))
        nodes = [_numa_node_of(b, sysfs) for b in ids[:local_world]]
        node = nodes[local_rank] if local_rank < len(nodes) else -1
        info["pci_bus_id"], info["numa_node"] = ids[local_rank] if local_rank < len(ids) else None, node
        pool = None
        if node >= 0:
            try:
                with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
                    ok = set(allowed)
                    pool = [c for c in _parse_cpulist(f.read()) if c in ok] or None
            except (OSError, ValueError):
                pool = None
        if pool is not None:
            sharers = [r for r in range(local_world) if nodes[r] == node]
        else:
            pool, sharers = allowed, list(range(local_world))
        slot = sharers.index(local_rank) if local_rank in sharers else 0
        per = max(1, len(pool) // len(sharers))
        mine = pool[slot * per:(slot + 1) * per] or pool
        info["sharers"], info["n_cpus"] = len(sharers), len(mine)
        info["cpus"] = (f"{mine[0]}-{mine[-1]}" if mine == list(range(mine[0], mine[-1] + 1)) else
                        ",".join(map(str, mine)))
        if apply and os.environ.get("MIS_PIN_NUMA", "1") != "0":
            # every thread of the process, not only the caller (threads created before this call keep their own mask)
            try:
                tids = [int(t) for t in os.listdir("/proc/self/task")]
            except OSError:
                tids = []
            for tid in tids or [0]:
                try:
                    os.sched_setaffinity(tid, mine)
                except OSError:
                    pass                                     # a thread that exited meanwhile
            os.sched_setaffinity(0, mine)
            # ... and torch's intra-op pool sized to the share (augmentation / validation metrics on the host would
            # otherwise run the machine-wide default thread count on these few cores)
            torch.set_num_threads(max(1, len(mine)))
            os.environ["OMP_NUM_THREADS"] = str(max(1, len(mine)))
            info["applied"], info["threads"] = True, len(mine)
    except Exception as e:       # affinity is an optimisation: never fail a run over it
        info["error"] = f"{type(e).__name__}: {e}"[:200]
    return info


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
        bucket_bytes = default_bucket_bytes(flat_grad.numel() * flat_grad.element_size(), bucket_bytes)
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
        self.buckets = bucket_layout(flat_grad.numel(), flat_grad.element_size(), bucket_bytes)
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

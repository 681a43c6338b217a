"""bench.py -- throughput of the Mean-Teacher training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload unet3d|unet2d|vnet|swin|cross|...]

``--gpus N`` with N > 1 is self-launching: when the process was not started by ``torch.distributed.run``
(no ``WORLD_SIZE`` in the environment) it re-executes itself as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``
-- one rank per GPU over RCCL -- and forwards the single JSON line of rank 0.  Started by ``torch.distributed.run``
directly it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A "step" is one full Mean-Teacher iteration (noise, student fwd on labeled+unlabeled, EMA-teacher fwd, fused
CE+Dice+consistency loss tail, student bwd, [RCCL all-reduce of the flat gradient bucket, issued per bucket while
the backward is still running], fused SGD+EMA, schedule advance) on one resident synthetic batch.

Default workload = BASELINE.json configs[2], the north-star target: Mean-Teacher ``unet_3D``, BraTS-like 96^3,
2 classes, 4 labeled + 4 unlabeled volumes per GPU.  Pure data parallel: every rank owns its own 4+4 shard -> weak
scaling; value = samples of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     live HIP-event timing of the dominant kernel: achieved / frac = the flops the matrix pipe EXECUTES over the
               launches' summed duration against that pipe's peak (<= 1): fp32 MFMA 157.3 TF with Winograd launches at 1 / 3.375
               (1 / 2.25) of their direct-convolution count; the Linear GEMMs / window attention of SwinUnet in their bf16x3 form
               at 6 executed multiply-adds per algorithmic one against the 2.5 PF dense bf16 peak.  ``algorithmic_tflops`` = the
               direct rate of the same launches; ``executed_step_frac`` = pipe-seconds at peak per step / step seconds.  The timed
               region runs the product configuration (teacher forward and weight gradients on side streams); the events are
               taken in a second region of the same run with the side streams off (``measured_in``, ``serial_ms_per_step``).
               ``traffic`` = HBM bytes per launch of the dominant kernel from the PMC counters of THIS run: bench.py re-executes
               itself for 1 + 2 serial steps under ``rocprofv3 --pmc FETCH_SIZE`` and again under ``WRITE_SIZE`` (N = 1, bounded
               to 90 s, ``traffic_unavailable`` says why when it could not)
  others       (N=1) the other single-GPU configurations of BASELINE.json -- configs[1] 2-D UNet (with its own CPU baseline),
               configs[3] SwinUnet, configs[4] per-GPU cross teaching (both also timed with the GEMMs on the fp32 MFMA
               instruction: ``ms_per_step_fp32_mfma``), and V-Net -- same protocol as the headline workload
  cpu_baseline the CPU oracle (a port of the reference arithmetic on stock torch CPU ops) timed on this host's cores on the
               SAME batch as the GPU line: thread count swept over {16, 32, 64} on a reduced batch, then 2 warm-up + 5 timed
               steps of the full batch at the best count, median (rank 0, N=1 only)
  host_enqueue_ms_per_step   the host time to ENQUEUE one step into an empty queue (median of 5 single steps, device idle before each)
  distributed  world size seen, per-rank times, ``affinity_rank0`` (NUMA pinning, N > 1), ``predicted`` = a MODEL of the gradient
               exchange (mis_hip/dist.py::predict_exchange; at N = 1 for 8 GPUs), and at N > 1 the measured exchange figures
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "cv-ssl-mis_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: bf16 MFMA, dense (the 5 PF headline includes 2:1 sparsity)

# step_gflop: algorithmic FLOP of one step, SURVEY.md s.8d: F_fwd * (3*(L+U) + U) (student fwd+dgrad+wgrad, teacher
# fwd); cross teaching: both students train on the whole batch.
WORKLOADS = {
    "unet3d": dict(config="Mean-Teacher 3D UNet (unet_3D), synthetic BraTS 96x96x96 2-class, bs=4+4 "
                          "(BASELINE configs[2], the north-star target)",
                   shape=(8, 1, 96, 96, 96), labeled=4, classes=2, cons_start=0, label="int64",
                   cpu_sample=(2, 1), step_gflop=121.98 * 28),
    "unet2d": dict(config="Mean-Teacher 2D UNet, synthetic ACDC 256x256 4-class, bs=24+24 (BASELINE configs[1])",
                   shape=(48, 1, 256, 256), labeled=24, classes=4, cons_start=1000, label="uint8",
                   cpu_sample=(8, 4), step_gflop=5.90 * 168),
    "vnet": dict(config="Mean-Teacher 3D V-Net (--model vnet), synthetic BraTS 96x96x96 2-class, bs=4+4 "
                        "(BASELINE configs[2] geometry with the reference's other 3-D backbone)",
                 shape=(8, 1, 96, 96, 96), labeled=4, classes=2, cons_start=0, label="int64",
                 cpu_sample=(2, 1), step_gflop=70.72 * 28),
    "uamt3d": dict(config="UA-MT 3D UNet (unet_3D), synthetic BraTS 96x96x96 2-class, bs=4+4, T=8 MC-dropout teacher "
                          "passes (SURVEY s.8 row n1; BASELINE configs[2] geometry)",
                   shape=(8, 1, 96, 96, 96), labeled=4, classes=2, cons_start=0, label="int64",
                   cpu_sample=None, step_gflop=121.98 * (28 + 32)),
    # UNETR forward: ViT 12 x (qkv + proj + MLP) on 216 tokens 18.3 GMAC + attention core 0.9 + patch embedding 0.7,
    # conv decoder / encoders (MONAI res blocks, k2s2 transposed convs) 52.9 GMAC -> 145.6 GFLOP per 96^3 volume
    "unetr": dict(config="Mean-Teacher UNETR (--model unetr), synthetic BraTS 96x96x96 2-class, bs=4+4 (SURVEY s.8 row "
                         "n4; BASELINE configs[2] geometry; parity unpinned: MONAI-based in the reference)",
                  shape=(8, 1, 96, 96, 96), labeled=4, classes=2, cons_start=0, label="int64",
                  cpu_sample=None, step_gflop=145.6 * 28),
    # SwinUNETR forward per 96^3 volume: UnetResBlock / UnetrUpBlock convolutions 581 GFLOP (the 96^3 level alone 454), Swin
    # encoder (Linear layers 30, 343-token window attention 20) ~50 -> ~630 GFLOP
    "swinunetr": dict(config="Mean-Teacher SwinUNETR (--model swinunetr), synthetic BraTS 96x96x96 2-class, bs=4+4 (SURVEY "
                             "s.8 rows n4 / f; BASELINE configs[2] geometry; parity unpinned: MONAI's network in the reference)",
                      shape=(8, 1, 96, 96, 96), labeled=4, classes=2, cons_start=0, label="int64",
                      cpu_sample=None, step_gflop=630.0 * 28),
    "swin": dict(config="Mean-Teacher ViT (SwinUNet 2D), synthetic ACDC 224x224 4-class, bs=24+24 "
                        "(BASELINE configs[3])",
                 shape=(48, 1, 224, 224), labeled=24, classes=4, cons_start=1000, label="uint8",
                 cpu_sample=(4, 2), step_gflop=12.17 * 168),
    # SwinUnet at 256^2 = DATA.IMG_SIZE 256 + MODEL.SWIN.WINDOW_SIZE 8 (reference config.py:194-195): 4096 tokens per
    # image instead of 3136 (Linear layers x1.306), 64-token windows (attention core x1.306 x 64/49)
    "cross": dict(config="Cross-teaching CNN+ViT 2D (UNet + SwinUNet window 8), synthetic ACDC 256x256 4-class, "
                         "bs=16+16 per GPU (BASELINE configs[4])",
                  shape=(32, 1, 256, 256), labeled=16, classes=4, cons_start=0, label="uint8", swin=(256, 8),
                  cpu_sample=None, step_gflop=(5.90 + 11.74 * 1.306 + 0.43 * 1.306 * 64 / 49) * 96),
    "cross224": dict(config="Cross-teaching CNN+ViT 2D (UNet + SwinUNet window 7), synthetic ACDC 224x224 4-class, "
                            "bs=16+16 per GPU (BASELINE configs[4] at the reference yaml's 224)",
                     shape=(32, 1, 224, 224), labeled=16, classes=4, cons_start=0, label="uint8",
                     cpu_sample=None, step_gflop=(4.52 + 12.17) * 96),
    "cnnvit": dict(config="CNN + ViT students with an EMA ViT teacher (train_cnn_meet_vit_2D: UNet + 2x SwinUNet), "
                          "synthetic ACDC 224x224 4-class, bs=8+8 per GPU (the script's defaults; SURVEY s.8 row n2)",
                   shape=(16, 1, 224, 224), labeled=8, classes=4, cons_start=1000, label="uint8",
                   cpu_sample=None, step_gflop=(4.52 + 12.17) * 48 + 12.17 * 8),
}
OTHERS = ("unet2d", "swin", "cross", "vnet")     # reported under "others" beside the default workload (N=1)
UNIT = {"unet3d": "volumes/s", "vnet": "volumes/s", "uamt3d": "volumes/s", "unetr": "volumes/s", "swinunetr": "volumes/s"}


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pg_timeout():
    """Explicit process-group timeout (default 600 s; MIS_PG_TIMEOUT_S): a rank that never reaches a collective aborts
    the run with RCCL's watchdog message instead of hanging the node until the driver's limit."""
    import datetime
    return datetime.timedelta(seconds=int(os.environ.get("MIS_PG_TIMEOUT_S", "600")))


def nccl_debug_env(env):
    """NCCL_DEBUG=WARN into per-process files (unless the caller configured RCCL's logging): a failed multi-GPU run
    reports RCCL's own warnings in its JSON line."""
    if "NCCL_DEBUG" in env or "NCCL_DEBUG_FILE" in env:
        return None
    import tempfile
    d = tempfile.mkdtemp(prefix="mis_bench_nccl_")
    env["NCCL_DEBUG"] = "WARN"
    env["NCCL_DEBUG_FILE"] = os.path.join(d, "nccl.%h.%p.log")
    return d


def collect_nccl_warnings(log_dir, limit=40):
    lines = []
    if log_dir and os.path.isdir(log_dir):
        for fn in sorted(os.listdir(log_dir)):
            try:
                with open(os.path.join(log_dir, fn), errors="replace") as f:
                    lines += [f"{fn}: {ln.rstrip()}" for ln in f if ln.strip()]
            except OSError:
                pass
    return lines[-limit:]


def _stderr_digest(text, keep=25):
    """The ranks' own failure lines (torch.distributed.run appends a long summary behind them) + the last lines."""
    lines = text.strip().splitlines()
    hits = [l for l in lines if any(w in l for w in ("FAILED", "Error", "error:", "Exception", "NCCL WARN", "Traceback"))]
    out = hits[:15]
    for l in lines[-keep:]:
        if l not in out:
            out.append(l)
    return [l[:400] for l in out]


def self_launch(argv, gpus):
    """Re-execute this script under torch.distributed.run with one rank per GPU; forward rank 0's output.  A failed run
    still prints ONE JSON line: the exit status, the tail of the ranks' stderr and RCCL's warnings."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this host
    env.setdefault("OMP_NUM_THREADS", "8")
    env["MIS_BENCH_LAUNCHED"] = "1"          # the ranks leave the failure report to this process
    log_dir = nccl_debug_env(env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True, errors="replace")
    sys.stderr.write(p.stderr)
    if p.returncode != 0:
        print(json.dumps({"metric": "training images-or-volumes/sec/node (Mean-Teacher step)", "value": None,
                          "n_gpus": gpus, "error": f"multi-GPU run failed with exit status {p.returncode}",
                          "stderr_tail": _stderr_digest(p.stderr),
                          "nccl_warnings": collect_nccl_warnings(log_dir)}), flush=True)
    return p.returncode


# ------------------------------------------------------------------------------------------------ workloads
def build_trainer(name, wl, world, stub=False):
    import torch
    if stub:
        return _StubTrainer(wl)
    from mis_hip.step import MeanTeacherTrainer
    C, L = wl["classes"], wl["labeled"]
    vit_teacher = None
    if name in ("cross", "cross224", "cnnvit"):
        from mis_hip.step import CnnMeetVitTrainer, CrossTeachingTrainer
        from networks.net_factory import net_factory
        if "swin" in wl:           # SwinUnet at another image size: the reference's --opts overrides
            from config import lite_config
            from networks.vision_transformer import SwinUnet
            cfg = lite_config()
            cfg.DATA.IMG_SIZE, cfg.MODEL.SWIN.WINDOW_SIZE = wl["swin"]
            model, ema = net_factory("unet", 1, C), SwinUnet(cfg, img_size=wl["swin"][0], num_classes=C)
        else:
            model, ema = net_factory("unet", 1, C), net_factory("ViT_Seg", 1, C)
        if name == "cnnvit":      # here `ema` is the Transformer STUDENT, vit_teacher its EMA
            vit_teacher = net_factory("ViT_Seg", 1, C)
            vit_teacher.load_state_dict(ema.state_dict())
    elif name in ("swin", "unet2d"):
        from networks.net_factory import net_factory
        key = "ViT_Seg" if name == "swin" else "unet"
        model, ema = net_factory(key, 1, C), net_factory(key, 1, C)
        ema.load_state_dict(model.state_dict())
    else:
        from networks.net_factory_3d import net_factory_3d
        key = {"vnet": "vnet", "unetr": "unetr", "swinunetr": "swinunetr"}.get(name, "unet_3D")
        model, ema = net_factory_3d(key, 1, C), net_factory_3d(key, 1, C)
        ema.load_state_dict(model.state_dict())
    if _dist_on(world):   # identical initial weights on every rank
        for m in (model, ema, vit_teacher):
            if m is not None:
                torch.distributed.broadcast(m.flat_param, 0)
    if name in ("cross", "cross224"):
        return CrossTeachingTrainer(model, ema, labeled_bs=L, num_classes=C, seed=1337, iter_num=1000)
    if name == "cnnvit":
        return CnnMeetVitTrainer(model, ema, vit_teacher, labeled_bs=L, num_classes=C, seed=1337, iter_num=1000)
    if name == "uamt3d":
        from mis_hip.step import UAMTTrainer
        return UAMTTrainer(model, ema, labeled_bs=L, num_classes=C, seed=1337, iter_num=1000)
    return MeanTeacherTrainer(model, ema, labeled_bs=L, num_classes=C, cons_start_iter=wl["cons_start"], seed=1337,
                              iter_num=1000, use_graph=os.environ.get("MIS_BENCH_GRAPH", "0") == "1")


def _dist_on(world):
    """The N > 1 code path.  MIS_BENCH_DIST_AT_WORLD1=1 (tests only) runs it in a process group of ONE rank: on the test box's
    single GPU that is a real RCCL group -- initialisation, broadcast, all-gather, barrier, the bucketed all-reduce -- under the
    very code an 8-GPU launch executes (tests/test_dist_gpu.py)."""
    return world > 1 or os.environ.get("MIS_BENCH_DIST_AT_WORLD1") == "1"


class _StubTrainer:
    """CPU/gloo stand-in for the step (``--stub``): exercises the launcher, the rendezvous, the barrier + max-over-ranks
    timing and the JSON contract without a GPU (tests/test_bench_cpu.py).  Never used for a measurement."""

    def __init__(self, wl):
        import torch
        self.w = torch.zeros(1 << 14)
        self.g = torch.zeros(1 << 14)

    def step(self, vol, lab):
        from mis_hip import dist
        self.g.copy_(vol.reshape(-1)[:self.g.numel()])
        if os.environ.get("MIS_STUB_SKIP_SYNC") == "1":      # test hook: a step whose exchange is missing must be caught
            scale = 1.0
        else:
            scale = dist.sync_gradients(self.g)
        self.w.add_(self.g, alpha=-0.01 * scale)

    def losses(self):
        return dict(loss=float(self.w.abs().mean()))


def run_workload(name, args, rank, world, kernel_events=True):
    """Warm-up, then EXACTLY args.steps timed steps between barrier + device sync; returns (rank 0) the result dict."""
    import torch
    import torch.distributed as tdist
    stub = args.stub
    dev = "cpu" if stub else "cuda"
    wl = WORKLOADS[name]
    torch.manual_seed(1337 + rank)
    tr = build_trainer(name, wl, world, stub)
    g = torch.Generator(device=dev).manual_seed(1337 + rank)
    shape = (4, 1, 64, 64) if stub else wl["shape"]
    vol = torch.rand(shape, generator=g, device=dev)
    lab = torch.randint(0, wl["classes"], (shape[0],) + shape[2:], generator=g,
                        device=dev).to(getattr(torch, wl["label"]))

    def sync():
        if not stub:
            torch.cuda.synchronize()

    enqueue = []

    def timed(steps):
        """`steps` steps between barrier + device sync on both sides; this rank's seconds.  The host's share -- the time
        the Python loop needs to ENQUEUE the steps (no wait for the device inside it) -- is kept in `enqueue`."""
        sync()
        if _dist_on(world):
            tdist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(vol, lab)
        enqueue.append((time.perf_counter() - t0, steps))
        sync()
        if _dist_on(world):
            tdist.barrier()
        sync()
        return time.perf_counter() - t0

    if not stub:
        # once per run, in front of the warm-up: NaNs into every CU's LDS (the test suite's fixture, tests/conftest.py) -- a kernel
        # that reads an LDS cell it never wrote shows up as non-finite losses here instead of depending on the previous process
        from mis_hip import lib as _lib
        _lib.check(_lib.load().mis_debug_poison_lds(None, _lib.stream_ptr()), "mis_debug_poison_lds")
    for _ in range(args.warmup):
        tr.step(vol, lab)
    # the product step is a launch tape (mis_hip/step.py::_TapedStep: two eager steps, one recorded, then replays): the timed
    # region must be replays whatever --warmup says (more untimed steps than asked for, never fewer)
    pmc_child = bool(getattr(args, "pmc_child", False))     # the re-execution under rocprofv3 --pmc (inrun_traffic): as few launches as
    if pmc_child and hasattr(tr, "use_tape"):               # possible -- every launch is serialised around its counter reads there
        tr.use_tape = False
    for _ in range(4):
        if not getattr(tr, "use_tape", False) or getattr(tr, "_tape", None) is not None:
            break
        tr.step(vol, lab)
    dt_local = timed(args.steps)
    host_loop_ms = enqueue[-1][0] / enqueue[-1][1] * 1e3
    # the host's own cost of enqueueing ONE step: into an empty queue (device idle), median of 5.  The loop figure above is
    # not that once the host is faster than the device: the runtime's queues fill up and the loop waits for the device in them
    # (20 steps of ~600 launches), i.e. it converges to the device time
    host_enqueue_ms = host_loop_ms
    if not stub and not pmc_child:
        one = []
        for _ in range(5):
            sync()
            t0 = time.perf_counter()
            tr.step(vol, lab)
            one.append((time.perf_counter() - t0) * 1e3)
        sync()
        host_enqueue_ms = sorted(one)[len(one) // 2]
    dt, per_rank = dt_local, [dt_local]
    if _dist_on(world):
        t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        tdist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        dt = max(per_rank)                         # MAX over ranks

    # ---- data parallel: prove the exchange happened and price it (before any diagnostic region changes the weights)
    dist_info = None
    if _dist_on(world):
        dist_info = _distributed_checks(tr, timed, args, rank, world, dev, stub, dt_local, retape=lambda: _retape(tr, vol, lab))

    # ---- roofline region: the same step with the side streams off, every MFMA launch bracketed by HIP events on its
    # launch stream.  A launch's duration measures the kernel only when it has the chip to itself: in the timed region
    # above the teacher forward and the weight gradients run on side streams beside the main stream's launches.
    prof, serial_dt, serial_steps = None, None, 0
    if not stub and kernel_events:
        from mis_hip import ops, plan as _plan, step as _stepmod
        keep = (_stepmod.TWO_STREAM, _plan.WGRAD_STREAM, getattr(tr, "use_tape", False))
        _stepmod.TWO_STREAM, _plan.WGRAD_STREAM = False, False
        if keep[2]:
            tr.use_tape = False          # the events are recorded by the eager op graph (a replayed tape has no Python around its launches)
        try:
            tr.step(vol, lab)
            ops.PROFILE = prof = []
            serial_steps = args.steps if args.serial else max(3, min(args.steps, 10))
            serial_dt = timed(serial_steps)
        finally:
            ops.PROFILE = None
            _stepmod.TWO_STREAM, _plan.WGRAD_STREAM = keep[:2]
            if keep[2]:
                tr.use_tape = True
    losses = tr.losses()
    assert all(map(lambda v: v == v and abs(v) < 1e6, losses.values())), f"non-finite losses {losses}"

    roofline = None
    if prof:
        per = {}
        for kname, flops, e0, e1, nbytes in prof:
            d = per.setdefault(kname, [0.0, 0.0, 0, 0.0])
            d[0] += flops
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += nbytes
        def reduction(k):
            """Winograd F(2x2x2, 3x3x3) / F(2x2, 3x3): 64 (16) multiplies per 2x2x2 (2x2) outputs instead of 216 (36)."""
            return (3.375 if k.startswith(("wino_fwd_kernel", "wino_wgrad_")) else
                    2.25 if k.startswith(("wino2d_fwd_kernel", "wino2d_wgrad_kernel")) else 1.0)

        def bf3(k):
            """The Linear GEMMs in their bf16x3 form (gemm.hip: last template argument 1, or 2 = B pre-split by
            mis_gemm_split_batch): every algorithmic multiply-add is six bf16 piece products (v_mfma_f32_16x16x32_bf16;
            16x16x16 in the short-contraction kernel), priced against the dense bf16 matrix peak."""
            targs = k[k.index("<") + 1:].rstrip(" >").split(", ") if "<" in k else []
            if k.startswith("gemm_nt_kernel<"):          # <BM, BN, EP, PREC, waves per workgroup>
                return len(targs) >= 4 and targs[3] in ("1", "2")
            if k.startswith("gemm_nt_short_kernel<"):    # <BN, EP, PREC>
                return bool(targs) and targs[-1] in ("1", "2")
            return k.startswith("gemm_tn_reg_kernel<1")

        def pipe_seconds(k, flops):
            """Time the matrix pipe needs for this launch's EXECUTED flops at its peak."""
            if bf3(k):
                return flops * 6.0 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
            return flops / reduction(k) / (PEAK_FP32_MFMA_TFLOPS * 1e12)

        fam_alg = sum(d[0] for d in per.values())
        fam_exec = sum(d[0] * (6.0 if bf3(k) else 1.0 / reduction(k)) for k, d in per.items())
        fam_pipe_s = sum(pipe_seconds(k, d[0]) for k, d in per.items())
        fam_time = sum(d[1] for d in per.values())
        dom = max(per, key=lambda k: per[k][1])
        red = reduction(dom)
        alg_tf = per[dom][0] / per[dom][1] / 1e12           # direct-convolution (algorithmic) flops over time
        dom_bf3 = bool(bf3(dom))
        peak = PEAK_BF16_MFMA_TFLOPS if dom_bf3 else PEAK_FP32_MFMA_TFLOPS
        achieved = alg_tf * 6.0 if dom_bf3 else alg_tf / red  # flops the matrix pipe executes over time: <= its peak
        # HBM bytes per launch come from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes): main()
        # re-executes this command under rocprofv3 for them and fills `traffic` (inrun_traffic); the figure of the last
        # committed PMC pass stays beside it, labelled as such
        traffic_prof = None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tfile = os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc_traffic.json")
            if os.path.exists(tfile):
                with open(tfile) as f:
                    ent = json.load(f)["kernels"].get(dom)
                if ent:
                    traffic_prof = dict(hbm_bytes_per_launch=ent["hbm_bytes_per_launch"],
                                        source=os.path.relpath(tfile, ROOT),
                                        note="a separate rocprofv3 PMC pass of this command, committed under profiles/; "
                                             "NOT a measurement of this run")
                    break
        # `achieved` / `frac` = EXECUTED matrix-pipe flops (algorithmic flops / winograd_reduction) over the kernel's
        # summed launch durations, against the fp32 MFMA peak: a fraction of the pipe, <= 1 by construction and the
        # quantity SQ_VALU_MFMA_BUSY_CYCLES measures (profiles/r03_*_pmc_mfma.csv).  The algorithmic rate (what a direct
        # convolution would have to sustain for the same time) is kept beside it.
        roofline = dict(bound="mfma", kernel=dom, achieved=round(achieved, 3), peak=peak,
                        unit="TFLOP/s", frac=round(achieved / peak, 4),
                        algorithmic_tflops=round(alg_tf, 3), winograd_reduction=red,
                        pipe=("bf16 MFMA (bf16x3: exact 3-way split of the fp32 operands, 6 piece products per multiply-add, fp32 "
                              "accumulation; `achieved` = 6 x algorithmic)" if dom_bf3 else "fp32 MFMA"),
                        traffic=None, traffic_from_profiles=traffic_prof,
                        # the same number as `achieved` under the name that says what it counts (a bf16x3 launch executes six
                        # bf16 piece products per algorithmic multiply-add, zero padding of K to 32 included): tables quote
                        # `algorithmic_tflops`, the useful rate
                        **({"executed_bf16_piece_tflops": round(achieved, 3)} if dom_bf3 else {}),
                        launches=per[dom][2], avg_launch_ms=round(per[dom][1] / per[dom][2] * 1e3, 4),
                        algorithmic_flops_per_launch_avg=per[dom][0] / per[dom][2],
                        algorithmic_bytes_per_launch_avg=per[dom][3] / per[dom][2],
                        executed_flops_per_launch_avg=per[dom][0] / per[dom][2] * (6.0 if dom_bf3 else 1.0 / red),
                        family=dict(kernel="all event-timed MFMA launches (wino*_fwd_kernel<*> / conv_fwd_kernel<*> forward + "
                                           "data gradient, wino*_wgrad_kernel<*> / conv_wgrad_kernel<*> weight gradient; "
                                           "gemm_nt_kernel<*> / gemm_tn_reg_kernel / gemm_tn_kernel<*> for SwinUnet)",
                                    achieved=round(fam_exec / fam_time / 1e12, 3),
                                    frac=round(fam_pipe_s / fam_time, 4),     # pipe-seconds at peak / seconds (fp32 and bf16 launches each against their own peak)
                                    algorithmic_tflops=round(fam_alg / fam_time / 1e12, 3),
                                    share_of_step_time=round(fam_time / serial_dt, 4)))
        roofline["measured_in"] = (f"a second region of {serial_steps} steps of this run with the side streams off (teacher "
                                   "forward and weight gradients on the main stream), HIP events on the launch stream "
                                   "around every MFMA launch")
        roofline["serial_ms_per_step"] = round(serial_dt / serial_steps * 1e3, 3)
        roofline["peak_note"] = ("157.3 TFLOP/s = the fp32 MFMA pipe at 2.4 GHz (MI355X_MICROARCH.md); in-kernel counters show the chip "
                                 "running the HBM-streaming Winograd kernels at 1.7-2.2 GHz (power budget: DESIGN.md s.3 'Power'), "
                                 "so a fraction of this peak understates the pipe's occupancy in cycles by 10-35 %")
    step_s = dt / args.steps
    step_frac = wl["step_gflop"] * 1e9 / step_s / (PEAK_FP32_MFMA_TFLOPS * 1e12)
    exec_frac = None
    if roofline is not None:
        roofline["step_algorithmic_flops"] = wl["step_gflop"] * 1e9
        roofline["step_algorithmic_flop_frac"] = round(step_frac, 4)      # direct-convolution flops: may exceed 1
        # the flops the matrix pipe EXECUTES per step (every event-timed launch of the serial region, Winograd launches
        # at 1/3.375 resp. 1/2.25 of their direct-convolution count) over the TIMED region's step time: <= 1
        exec_frac = fam_pipe_s / serial_steps / step_s
        roofline["executed_step_flops"] = fam_exec / serial_steps
        roofline["executed_step_frac"] = round(exec_frac, 4)
    grad_bytes = [] if stub else [int(m.flat_param.numel()) * 4 for m in
                                  (getattr(tr, a, None) for a in ("model", "model1", "model2")) if m is not None]
    res = dict(value=round(shape[0] * world * args.steps / dt, 3), unit=UNIT.get(name, "images/s"), grad_bytes=grad_bytes,
               ms_per_step=round(step_s * 1e3, 3), step_flop_frac=round(step_frac, 4),
               host_enqueue_ms_per_step=round(host_enqueue_ms, 3), host_enqueue_loop_ms_per_step=round(host_loop_ms, 3),
               step_enqueue=("launch tape: the recorded C-ABI launch sequence of one eager step, replayed (%d entries; "
                             "MIS_STEP_TAPE=0: the eager op graph)" % len(tr._tape)) if getattr(tr, "_tape", None) is not None
               else "eager op graph",
               executed_step_frac=None if exec_frac is None else round(exec_frac, 4), roofline=roofline,
               losses={k: round(v, 6) for k, v in losses.items()},
               per_rank_ms_per_step=[round(t / args.steps * 1e3, 3) for t in per_rank], distributed=dist_info)
    del tr
    if not stub:
        torch.cuda.empty_cache()
    return res


def _retape(tr, vol, lab):
    """After a configuration change (bucketers swapped out, exchange stubbed): drop the recorded step and record it again, untimed."""
    if getattr(tr, "use_tape", False):
        tr._tape, tr._tape_warm = None, 0
        for _ in range(4):
            if tr._tape is not None:
                break
            tr.step(vol, lab)


def _distributed_checks(tr, timed, args, rank, world, dev, stub, dt_overlap, retape=lambda: None):
    """N > 1: (1) every rank must hold bit-identical student (and teacher) weights after the timed steps -- the
    gradients were exchanged and the same update applied everywhere; the run FAILS otherwise.  (2) The price of the
    exchange: the same steps with one blocking all-reduce after the backward instead of the bucketed one overlapped with
    it (MIS_GRAD_OVERLAP=0), and with no exchange at all (what a single GPU does; run last, the ranks' weights drift
    apart in it).  exposed = ms/step of the timed region - ms/step without exchange."""
    import torch
    import torch.distributed as tdist
    from mis_hip import dist as mdist
    seen = tdist.get_world_size()
    if seen != world:
        raise SystemExit(f"process group has {seen} ranks, --gpus says {world}")
    flats = [tr.w] if stub else [m.flat_param for m in (getattr(tr, a, None) for a in
                                                        ("model", "ema_model", "model1", "model2")) if m is not None]
    # bit-level fingerprint: sum and xor-fold of the raw words, in int64 (exact, order-independent)
    words = torch.cat([f.detach().reshape(-1).view(torch.int32).to(torch.int64) for f in flats])
    fp = torch.stack([words.sum(), (words * (torch.arange(words.numel(), device=words.device) % 8191 + 1)).sum()])
    allfp = [torch.zeros_like(fp) for _ in range(world)]
    tdist.all_gather(allfp, fp)
    identical = all(bool(torch.equal(allfp[0], x)) for x in allfp)
    if not identical:
        raise SystemExit("data-parallel check FAILED: the ranks hold different weights after the timed steps "
                         "(the gradient exchange did not reach every parameter)")
    n = max(3, min(args.steps, 10))

    def region():
        retape()
        t = torch.tensor([timed(n)], device=dev, dtype=torch.float64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        return float(t.item()) / n * 1e3

    out = dict(world_size_seen=seen, params_identical=True, weights_fingerprint=[int(v) for v in allfp[0].tolist()],
               grad_bytes=sum(int(f.numel()) * 4 for f in flats if f.requires_grad or True),
               overlapped_ms_per_step=round(dt_overlap / args.steps * 1e3, 3), regions_steps=n)
    keep = {a: getattr(tr, a) for a in ("_bucketer", "_bucketers") if hasattr(tr, a)}
    if keep:      # one blocking all-reduce of the whole flat gradient after the backward
        for a, v in keep.items():
            setattr(tr, a, None if a == "_bucketer" else (None,) * len(v))
        out["blocking_allreduce_ms_per_step"] = round(region(), 3)
        for a, v in keep.items():
            setattr(tr, a, v)
    real_sync = mdist.sync_gradients
    try:          # no exchange at all: the single-GPU step on this rank's shard
        mdist.sync_gradients = lambda flat_grad, group=None: 1.0 / world
        for a, v in keep.items():
            setattr(tr, a, None if a == "_bucketer" else (None,) * len(v))
        out["no_exchange_ms_per_step"] = round(region(), 3)
    finally:
        mdist.sync_gradients = real_sync
        for a, v in keep.items():
            setattr(tr, a, v)
    retape()          # back to the product configuration
    out["exposed_allreduce_ms_per_step"] = round(out["overlapped_ms_per_step"] - out["no_exchange_ms_per_step"], 3)
    return out


def _predicted_exchange(res, world):
    """The exchange MODEL of mis_hip/dist.py for this workload's gradient buffers, at the run's world size (at N = 1: for
    8 GPUs, the node the scaling run uses), with the backward time it would have to hide in."""
    try:
        from mis_hip import dist as _mdist
        pred = _mdist.predict_exchange(res.get("grad_bytes") or [], world if world > 1 else 8)
        pred["step_ms_without_exchange"] = res["ms_per_step"] if world == 1 else None
        return pred
    except Exception as e:      # a model line must never fail a measurement
        return {"error": f"{type(e).__name__}: {e}"[:200]}


# ------------------------------------------------------------------------------------------------ CPU baseline
def _physical_cores():
    try:
        seen = set()
        with open("/proc/cpuinfo") as f:
            phys = core = None
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(kind, wl, timed=5, warm=2, budget_s=150.0, cands=(16, 32, 64), full_batch=True, also_threads=None, deadline=None,
                 min_timed=3):
    """The oracle step (reference arithmetic on stock torch CPU ops, dropout + noise active) on this host's cores.
    ``torch.set_num_threads`` is swept over ``cands`` on the reduced batch ``wl["cpu_sample"]`` of the same geometry
    (1 warm-up + 1 timed step each; 8 and the physical core count lost every sweep of rounds 2-4 and are no longer
    tried); the best count then times the FULL batch of the GPU line: ``warm`` warm-up + ``timed`` timed steps
    (2 + >= 5), median -- ``value`` is that full-batch figure."""
    import torch
    from oracle.nets import OracleUNet2D, OracleUNet3D, OracleVNet
    from oracle.step import mean_teacher_step
    C = wl["classes"]
    g = torch.Generator().manual_seed(1337)
    if kind == "swin":
        from oracle import filler
        from oracle.swin import OracleSwinUnet
        onet = OracleSwinUnet(C)
        student = filler.fill_state_dict(onet.new_state())
    else:
        onet = {"unet2d": lambda: OracleUNet2D(1, C), "unet3d": lambda: OracleUNet3D(C, 1),
                "vnet": lambda: OracleVNet(C, 1)}[kind]()
        student = onet.new_state()
        for n, t in student.items():
            if t.dim() >= 2:
                torch.nn.init.kaiming_normal_(t, generator=g)
            elif n.endswith("weight") or n.endswith("running_var"):
                t.fill_(1.0)
    teacher = {k: v.clone() for k, v in student.items()}
    mom = {}
    it = [1000]

    def make(B, L):
        shape = (B,) + wl["shape"][1:]
        vol = torch.rand(shape, generator=g)
        lab = torch.randint(0, C, (B,) + shape[2:], generator=g).to(getattr(torch, wl["label"]))

        def one():
            noise = torch.clamp(torch.randn((B - L,) + shape[1:], generator=g) * 0.1, -0.2, 0.2)
            t0 = time.perf_counter()
            mean_teacher_step(onet, student, teacher, mom, vol, lab, noise, it[0], labeled_bs=L, num_classes=C,
                              cons_start_iter=wl["cons_start"])
            it[0] += 1
            return time.perf_counter() - t0
        return one

    Bs, Ls = wl["cpu_sample"]
    Bf, Lf = wl["shape"][0], wl["labeled"]
    small, full = make(Bs, Ls), make(Bf, Lf)
    logical, physical = os.cpu_count() or 1, _physical_cores()
    default_threads = torch.get_num_threads()
    cands = sorted({n for n in cands if 1 <= n <= logical}) or [min(logical, 16)]
    sweep, t_start = {}, time.perf_counter()
    for n in cands:
        torch.set_num_threads(n)
        small()
        sweep[n] = small()
        if time.perf_counter() - t_start > budget_s * 0.4:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    ts = sweep[best]
    sp = "x".join(map(str, wl["shape"][2:]))
    if not full_batch:
        # token workloads (north_star names ACDC / BraTS for the CPU figure; this completes the table): the reduced batch only
        # -- images/s of these per-sample networks barely depends on the batch -- 1 warm-up + 3 timed steps, median
        small()
        times = sorted(small() for _ in range(3))
        t = times[1]
        torch.set_num_threads(default_threads)
        return dict(value=Bs / t, unit=UNIT.get(kind, "images/s"), cores=best, kind="port", warmup_steps=1, timed_steps=3,
                    s_per_step_all=[round(x, 3) for x in times], host=dict(logical_cpus=logical, physical_cores=physical),
                    thread_sweep_s_per_step={str(k): round(v, 3) for k, v in sweep.items()},
                    extrapolation=f"measured on the reduced batch {Ls}+{Bs - Ls}; quoted as images/s for the GPU line's {Lf}+{Bf - Lf} "
                                  "(the full batch on the CPU is not run: ~12 x the time for the same rate)",
                    sample=f"oracle.step.mean_teacher_step on the REDUCED batch {Ls}+{Bs - Ls} of {sp}: 1 warm-up + 3 timed steps, "
                           f"median {t:.3f} s/step, at {best} threads (best of {sorted(sweep)} on this batch), torch "
                           f"{torch.__version__} CPU")
    # ``deadline`` (time.perf_counter() value, the caller's budget for the whole CPU leg): host speed differs by 2x between boxes
    # of the pool, so a slow host gives up the second warm-up step and the timed steps beyond ``min_timed`` rather than the run's
    # "minutes" contract; what was actually run is what `sample` / warmup_steps / timed_steps report
    late = (lambda: deadline is not None and time.perf_counter() > deadline)
    first = full()
    warm_done = 1
    while warm_done < warm and not (deadline is not None and time.perf_counter() + (timed + 1) * first > deadline):
        full()
        warm_done += 1
    warm = warm_done
    times = []
    for k in range(timed):
        if k >= min_timed and late():
            break
        times.append(full())
    timed = len(times)
    times = sorted(times)
    t = times[len(times) // 2]
    other = None
    if also_threads and also_threads != best and also_threads <= logical and \
            not (deadline is not None and time.perf_counter() + 2.2 * t > deadline):
        # the thread count was chosen on the reduced batch: one full-batch step at another count settles whether it holds there
        torch.set_num_threads(also_threads)
        full()
        to = full()
        other = dict(threads=also_threads, s_per_step=round(to, 3), value=Bf / to, warmup_steps=1, timed_steps=1)
    torch.set_num_threads(default_threads)
    return dict(value=Bf / t, unit=UNIT.get(kind, "images/s"), cores=best, kind="port", warmup_steps=warm, timed_steps=timed,
                s_per_step_all=[round(x, 3) for x in times],
                host=dict(logical_cpus=logical, physical_cores=physical),
                thread_sweep_s_per_step={str(k): round(v, 3) for k, v in sweep.items()},
                reduced_batch=dict(batch=f"{Ls}+{Bs - Ls}", value=Bs / ts, s_per_step=round(ts, 3)),
                full_batch_at_other_thread_count=other,
                sample=f"oracle.step.mean_teacher_step on the GPU line's batch {Lf}+{Bf - Lf} of {sp}: {warm} warm-up + "
                       f"{timed} timed steps, median {t:.3f} s/step, at {best} threads -- the BEST OF THE SWEEP "
                       f"{sorted(sweep)} ON THE REDUCED BATCH {Ls}+{Bs - Ls} ({ts:.3f} s/step there), applied to the full batch"
                       + (f"; one full-batch step at {other['threads']} threads: {other['s_per_step']:.3f} s" if other else "")
                       + f"; torch {torch.__version__} CPU")


# ------------------------------------------------------------------------------------------------ in-run HBM traffic
def _short_kernel_name(name):
    """rocprofv3's kernel name -> the short form ops.py / the *_kernel_name functions use (scripts/pmc_traffic.py)."""
    import re
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace(" >", ">").strip()


def inrun_traffic(workload, dom, alg_bytes, budget_s=90.0):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected BY THIS RUN: bench.py re-executes
    itself for 1 warm-up + 2 steps (side streams off) under ``rocprofv3 --kernel-trace --pmc FETCH_SIZE`` and again
    under ``--pmc WRITE_SIZE`` (the two counters do not fit one pass; no other trace domain is enabled), as
    MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests
    of wide coalesced reads at 64 bytes, so read bytes = FETCH_SIZE x 1024 x 2; WRITE_SIZE x 1024 at face value.
    Bounded by ``budget_s`` and failure-tolerant: returns (dict | None, reason | None)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    t_start = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="mis_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    want = dom.replace(" >", ">")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            left = budget_s - (time.perf_counter() - t_start)
            if left < 20:
                return None, f"time budget ({budget_s:.0f} s) spent before the {counter} pass"
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--serial", "--steps", "2",
                   "--warmup", "1", "--no-cpu-baseline", "--no-others", "--no-kernel-events", "--no-traffic", "--pmc-child"]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                   errors="replace", timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"the {counter} pass exceeded the time budget ({budget_s:.0f} s for both passes)"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"the {counter} pass failed (exit {p.returncode}): " + " | ".join(_stderr_digest(p.stderr, 4))[:400]
            n = tot = 0
            for fn in files:
                with open(fn, newline="") as f:
                    for row in csv.DictReader(f):
                        if row.get("Counter_Name") == counter and _short_kernel_name(row["Kernel_Name"]) == want:
                            n += 1
                            tot += float(row["Counter_Value"])
            if n == 0:
                return None, f"kernel {dom!r} has no {counter} rows in the counter collection"
            got[counter] = (n, tot)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd = got["FETCH_SIZE"][1] * 1024 * 2 / got["FETCH_SIZE"][0]
    wr = got["WRITE_SIZE"][1] * 1024 / got["WRITE_SIZE"][0]
    return dict(hbm_bytes_per_launch=round(rd + wr), read_bytes_per_launch=round(rd), write_bytes_per_launch=round(wr),
                launches_sampled=got["FETCH_SIZE"][0], algorithmic_bytes_per_launch=round(alg_bytes),
                over_algorithmic=round((rd + wr) / alg_bytes, 3) if alg_bytes else None,
                method="this run: bench.py re-executed for 1 warm-up + 2 serial steps under rocprofv3 --kernel-trace --pmc "
                       "FETCH_SIZE, then --pmc WRITE_SIZE (separate passes); read = FETCH_SIZE KiB x 1024 x 2 (gfx950 "
                       "half-count of wide reads), written = WRITE_SIZE KiB x 1024; averaged over the kernel's launches",
                seconds=round(time.perf_counter() - t_start, 1)), None


# ------------------------------------------------------------------------------------------------ main
def main():
    marks = [("start", time.perf_counter())]          # wall-clock phases of this process, reported as `timing_s` (rank 0, N = 1)

    def mark(name):
        marks.append((name, time.perf_counter()))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unet3d", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the 'others' block (the other single-GPU configs)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 PMC passes of this command that fill roofline.traffic (N=1 only)")
    ap.add_argument("--serial", action="store_true",
                    help="side streams off in the timed region too (MIS_TWO_STREAM=0 MIS_WGRAD_STREAM=0): every kernel has "
                         "the chip to itself -- what the rocprofv3 kernel statistics under profiles/ are collected with")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # CPU/gloo launcher test only
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # set by inrun_traffic for its re-execution
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")

    import torch
    backend = None
    if not args.stub:
        torch.cuda.set_device(local_rank)
    affinity = None
    if world > 1:
        # this rank's threads next to its GPU's PCIe root: its share of the cores of the device's NUMA node (mis_hip/dist.py)
        from mis_hip import dist as _mdist
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        ids = [] if args.stub else _mdist.device_bus_ids(min(local_world, torch.cuda.device_count()))
        affinity = _mdist.pin_rank_to_numa(local_rank, local_world, ids)
    nccl_logs = None
    if _dist_on(world):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not args.stub and "MIS_BENCH_LAUNCHED" not in os.environ:
            nccl_logs = nccl_debug_env(os.environ)       # launched by torchrun directly: RCCL's warnings on failure
        if args.stub:
            backend = "gloo"
            torch.distributed.init_process_group("gloo", timeout=_pg_timeout())
        else:
            backend = "nccl (RCCL)"
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                                 timeout=_pg_timeout())

    two_stream = wgrad_stream = False
    if not args.stub:
        from mis_hip import plan as _plan, step as _step
        if args.serial:
            _step.TWO_STREAM = _plan.WGRAD_STREAM = False
        two_stream, wgrad_stream = _step.TWO_STREAM, _plan.WGRAD_STREAM

    wl = WORKLOADS[args.workload]
    try:
        if os.environ.get("MIS_MAIN_PRIORITY") and not args.stub:
            # experiment: the whole workload on a stream of this HIP priority (the side streams keep the default one)
            import torch as _t
            with _t.cuda.stream(_t.cuda.Stream(priority=int(os.environ["MIS_MAIN_PRIORITY"]))):
                res = run_workload(args.workload, args, rank, world, kernel_events=not args.no_kernel_events)
        else:
            res = run_workload(args.workload, args, rank, world, kernel_events=not args.no_kernel_events)
    except BaseException as e:
        if _dist_on(world) and rank == 0 and "MIS_BENCH_LAUNCHED" not in os.environ:
            print(json.dumps({"metric": "training images-or-volumes/sec/node (Mean-Teacher step)", "value": None,
                              "n_gpus": world, "error": f"{type(e).__name__}: {e}"[:2000],
                              "nccl_warnings": collect_nccl_warnings(nccl_logs)}), flush=True)
        raise

    if rank == 0:
        out = {
            "metric": "training images-or-volumes/sec/node (Mean-Teacher step)",
            "value": res["value"], "unit": res["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (U[0,1) images, uniform labels, random-init weights, resident in HBM)",
            "config": {"workload": wl["config"], "per_gpu_batch": f"{wl['labeled']}+{wl['shape'][0] - wl['labeled']}",
                       "global_batch": wl["shape"][0] * world, "parallelism": f"dp{world}",
                       "dropout": "on (Philox)", "teacher_noise": "on", "iter_num_start": 1000,
                       "teacher_forward": "side stream (beside the student forward)" if two_stream else "same stream",
                       "weight_gradients": "side stream (beside the data-gradient chain)" if wgrad_stream else "same stream",
                       "step_enqueue": res.get("step_enqueue", "eager op graph")},
            "distributed": dict(res["distributed"] or {"world_size_seen": 1}, backend=backend,
                                per_rank_ms_per_step=res["per_rank_ms_per_step"], max_ms_per_step=res["ms_per_step"],
                                affinity_rank0=affinity, predicted=_predicted_exchange(res, world)),
            "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"],
            "host_enqueue_frac": round(res["host_enqueue_ms_per_step"] / res["ms_per_step"], 3),
            "host_enqueue_note": "one step into an empty queue (median of 5); host_enqueue_loop_ms_per_step = the timed loop's own "
                                 "enqueue time per step, which includes waiting in full queues once the host outruns the device",
            "host_enqueue_loop_ms_per_step": res.get("host_enqueue_loop_ms_per_step"),
            "losses_last_step": res["losses"],
            "roofline": res["roofline"],
        }
        if not args.stub and args.workload in ("swin", "cross", "cross224", "cnnvit", "unetr", "swinunetr"):
            from mis_hip import tops as _tops
            mask = _tops.set_split_precision(-1)
            out["config"]["linear_gemm_arithmetic"] = (
                "fp32 MFMA (v_mfma_f32_16x16x4_f32)" if mask == 0 else
                f"bf16x3 (mask {mask}: exact 3-way bf16 split of the fp32 operands, 6 piece products per multiply-add on "
                "v_mfma_f32_16x16x32_bf16, fp32 accumulation; error vs float64 not above the fp32 MFMA form's -- "
                "mis_gemm_set_split_precision(0) / MIS_GEMM_BF3=0 selects the fp32 MFMA)")
        if args.stub:
            out["data"] = "stub (CPU/gloo launcher test, not a measurement)"
    mark("setup_and_headline")
    if not _dist_on(world) and not args.stub and not args.no_others and args.workload == "unet3d":
        # the other single-GPU configurations of BASELINE.json, shorter runs of the same protocol
        import copy
        oargs = copy.copy(args)      # the same protocol as the headline workload (a short warm-up leaves first-launch costs in the timed steps)
        others = {}
        for name in OTHERS:
            r = run_workload(name, oargs, rank, world, kernel_events=not args.no_kernel_events)
            rf = r["roofline"] or {}
            extra = {}
            if name in ("swin", "cross"):
                # the nn.Linear GEMMs run as bf16x3 split products by default (gemm.hip): the same workload with them on
                # the fp32 MFMA instruction, timed the same way, so that the line carries both
                from mis_hip import tops as _tops
                prev = _tops.set_split_precision(0)
                try:
                    r32 = run_workload(name, oargs, rank, world, kernel_events=False)
                finally:
                    _tops.set_split_precision(prev)
                extra = dict(linear_gemm_arithmetic="bf16x3: every fp32 operand cut exactly into three bf16 pieces, six piece "
                             "products per multiply-add on v_mfma_f32_16x16x32_bf16, fp32 accumulation -- error against float64 "
                             "no larger than the fp32 MFMA kernels' (tests/test_token_kernels_gpu.py, both forms)",
                             ms_per_step_fp32_mfma=r32["ms_per_step"], value_fp32_mfma=r32["value"],
                             losses_fp32_mfma=r32["losses"], losses=r["losses"])
            if name in ("swin", "cross") and rf and not args.no_traffic and not args.no_kernel_events:
                # HBM bytes of THIS workload's dominant kernel over its algorithmic bytes, from two PMC passes of this run
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                tr, why = inrun_traffic(name, rf["kernel"], rf["algorithmic_bytes_per_launch_avg"], budget_s=60.0)
                extra["dominant_kernel_hbm_over_algorithmic"] = None if tr is None else tr["over_algorithmic"]
                extra["dominant_kernel_traffic"] = tr if tr is not None else dict(unavailable=why)
            others[name] = dict(workload=WORKLOADS[name]["config"], value=r["value"], unit=r["unit"], **extra,
                                ms_per_step=r["ms_per_step"], steps=oargs.steps,
                                host_enqueue_ms_per_step=r["host_enqueue_ms_per_step"],
                                serial_ms_per_step=rf.get("serial_ms_per_step"),
                                grad_bytes=r["grad_bytes"],
                                executed_step_frac=r["executed_step_frac"],          # matrix-pipe flops executed: <= 1
                                algorithmic_step_flop_frac=r["step_flop_frac"],      # direct-convolution flops: may exceed 1
                                dominant_kernel=rf.get("kernel"), dominant_kernel_frac=rf.get("frac"),
                                dominant_kernel_algorithmic_tflops=rf.get("algorithmic_tflops"))
        out["others"] = others
        mark("others")
    if _dist_on(world) and not args.no_others and args.workload == "unet3d":
        # config 5 (cross teaching, the BASELINE configuration that is DEFINED on 8 GPUs: 16+16 images per GPU) behind the
        # default workload: two students, two gradient bucketers, the second student's backward on a side stream
        import copy
        oargs = copy.copy(args)
        oargs.steps, oargs.warmup = max(5, args.steps // 2), min(args.warmup, 2)
        r = run_workload("cross", oargs, rank, world, kernel_events=False)
        if rank == 0:
            out["others"] = {"cross": dict(workload=WORKLOADS["cross"]["config"], value=r["value"], unit=r["unit"],
                                           ms_per_step=r["ms_per_step"], steps=oargs.steps, n_gpus=world,
                                           per_rank_ms_per_step=r["per_rank_ms_per_step"],
                                           host_enqueue_ms_per_step=r["host_enqueue_ms_per_step"],
                                           distributed=dict(r["distributed"] or {},
                                                            predicted=_predicted_exchange(r, world)))}
    if rank == 0:
        single = not _dist_on(world) and not args.stub
        if single and not args.no_traffic and not args.no_kernel_events and out["roofline"]:
            # HBM bytes of the dominant kernel from the PMC counters, collected by re-executing this command (before the
            # CPU baseline occupies the host's cores)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            rf = out["roofline"]
            tr, why = inrun_traffic(args.workload, rf["kernel"], rf["algorithmic_bytes_per_launch_avg"])
            rf["traffic"] = tr
            if tr is None:
                rf["traffic_unavailable"] = why
            mark("traffic")
        if single and not args.no_cpu_baseline and wl["cpu_sample"] is not None:
            # one budget for the CPU legs of the line (MIS_BENCH_CPU_BUDGET_S, default 170 s; 120 of them for the headline workload)
            cpu_t0 = time.perf_counter()
            cpu_budget = float(os.environ.get("MIS_BENCH_CPU_BUDGET_S", "170"))
            out["cpu_baseline"] = cpu_baseline(args.workload, wl, also_threads=64 if args.workload == "unet3d" else None,
                                               deadline=cpu_t0 + cpu_budget * 0.7)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
            if "others" in out and "swin" in out["others"] and args.workload == "unet3d":
                o4 = out["others"]["swin"]
                o4["cpu_baseline"] = cpu_baseline("swin", WORKLOADS["swin"], budget_s=40.0, cands=(16, 32), full_batch=False)      # ~5 s
                o4["gpu_over_cpu"] = round(o4["value"] / o4["cpu_baseline"]["value"], 2)
            if "others" in out and "unet2d" in out["others"] and args.workload == "unet3d":
                # the ACDC figure the north star asks for beside the BraTS one: the same oracle step on config 2's batch
                o2 = out["others"]["unet2d"]
                o2["cpu_baseline"] = cpu_baseline("unet2d", WORKLOADS["unet2d"], budget_s=60.0, deadline=cpu_t0 + cpu_budget)
                o2["gpu_over_cpu"] = round(o2["value"] / o2["cpu_baseline"]["value"], 2)
        if single:
            mark("cpu_baseline")
            out["timing_s"] = {b[0]: round(b[1] - a[1], 1) for a, b in zip(marks, marks[1:])}
            out["timing_s"]["total"] = round(marks[-1][1] - marks[0][1], 1)
        print(json.dumps(out), flush=True)
    if _dist_on(world):
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

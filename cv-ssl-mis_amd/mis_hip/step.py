"""The Mean-Teacher training step as one fused device-side sequence.

Replaces the loop body of the reference (code/train_mean_teacher_2D.py:202-236,
code/train_mean_teacher_3D.py:134-166):

    noise -> student forward (labeled+unlabeled) -> EMA-teacher forward (noised unlabeled)
    -> softmax / CE / Dice / softmax-MSE consistency (+ dlogits)  [one fused loss tail]
    -> student backward -> [RCCL all-reduce of the flat gradient bucket] -> fused SGD + EMA
    -> poly-LR / EMA-alpha / consistency-weight schedule advanced on the device

Nothing in the sequence allocates or synchronises with the host: scalars stay in a small device
buffer (``trainer.out``) that the caller reads when it wants to log (the reference forces >= 3 + C
host syncs per step).  Because of that the whole step can be captured once into a
hipGraph (``use_graph=True``) and replayed; the RNG offset, learning rate, EMA alpha and the
consistency weight live in a device-resident ``MisStepState`` so replays stay correct.
"""
import os

import torch

from . import dist, ops
from . import lib as _lib

# teacher forward (cross teaching: the second student) on a side stream, see _run: bit-identical training, the second
# network's launches fill the CUs the first one's launch tails and small deep layers leave idle (MIS_TWO_STREAM=0: off)
TWO_STREAM = os.environ.get("MIS_TWO_STREAM", "1") != "0"
# the step as a launch tape (lib.LaunchTape): after two eager steps the trainer records one step's launches and replays them --
# one ctypes call per launch instead of the Python op graph (host enqueue 8.6 -> ~2 ms of a 20 ms SwinUnet step; a captured
# hipGraph costs the host MORE than the eager step on this stack).  MIS_STEP_TAPE=0: eager.  Bit-identical training.
STEP_TAPE = os.environ.get("MIS_STEP_TAPE", "1") != "0"
# data parallel: issue the all-reduce of finished gradient buckets while the backward is still running
# (dist.GradBucketer); MIS_GRAD_OVERLAP=0 falls back to one blocking all-reduce after the backward
GRAD_OVERLAP = os.environ.get("MIS_GRAD_OVERLAP", "1") != "0"


def backward_and_sync(model, pg, bucketer=None):
    """Student backward + the step's only exchange: the all-reduce (sum) of the flat gradient bucket.  Returns the
    1/world scale the optimizer kernel folds in.  With a bucketer the finished buckets travel during the backward."""
    if bucketer is None:
        model.backward_raw()
        return _lib.tape_call(dist.sync_gradients, model.flat_grad, pg)
    _lib.tape_call(bucketer.begin)
    model.backward_raw(on_progress=bucketer.advance)
    return _lib.tape_call(bucketer.finish)


def make_bucketer(model, pg, defer_tail=False):
    """MIS_FORCE_BUCKETER=1: the bucketer also for a single-rank group (tests / scripts/ddp_overhead.py run the RCCL
    stream machinery on one GPU)."""
    on = dist.world_size(pg) > 1 or (os.environ.get("MIS_FORCE_BUCKETER", "0") == "1" and dist.initialized())
    return dist.GradBucketer(model.flat_grad, pg, defer_tail=defer_tail) if (GRAD_OVERLAP and on) else None


class _TapedStep:
    """Mixin: ``_tape_step(run, tensors)`` runs ``run(*static)`` eagerly twice (plans, scratch buffers and lazily built job
    tables settle), records the third run as a lib.LaunchTape and replays it from then on.  ``tensors`` are the step's inputs:
    the recording sees static copies (or the caller's own tensors when they are the same storage every step, as in bench.py)."""

    _tape = None
    _tape_warm = 0
    _tape_static = None
    TAPE_WARMUP = 2

    def _tape_step(self, run, tensors):
        if self._tape is None:
            if self._tape_warm < self.TAPE_WARMUP:
                self._tape_warm += 1
                run(*tensors)
                return
            self._tape_static = tuple(t.clone() for t in tensors)
            self._tape_src = tuple((t.data_ptr(), tuple(t.shape)) for t in tensors)
            tape = _lib.LaunchTape()
            with tape.recording():
                run(*self._tape_static)
            self._tape = tape
            return
        for t, st in zip(tensors, self._tape_static):
            if t.shape != st.shape or t.dtype != st.dtype:
                raise RuntimeError("the taped step was recorded for inputs of another shape; construct the trainer with "
                                   "use_tape=False (MIS_STEP_TAPE=0) for varying batch geometry")
            st.copy_(t)
        self._tape.replay()


class MeanTeacherTrainer(_TapedStep):
    def __init__(self, model, ema_model, *, labeled_bs, num_classes, base_lr=0.01, max_iterations=30000,
                 ema_decay=0.99, consistency=0.1, consistency_rampup=200.0, cons_start_iter=0, seed=1337,
                 iter_num=0, momentum=0.9, weight_decay=1e-4, process_group=None, use_graph=False, use_tape=None):
        if model.flat_param.numel() != ema_model.flat_param.numel():
            raise RuntimeError("student and teacher must be the same architecture")
        self.model, self.ema_model = model, ema_model
        self.labeled_bs, self.num_classes = labeled_bs, num_classes
        self.hyper = dict(base_lr=float(base_lr), max_iterations=float(max_iterations),
                          ema_decay=float(ema_decay), consistency=float(consistency),
                          rampup=float(consistency_rampup), ramp_div=150, cons_start_iter=int(cons_start_iter))
        self.momentum, self.weight_decay = momentum, weight_decay
        self.pg = process_group
        self.world = dist.world_size(process_group)
        self.state = ops.new_step_state()
        ops.step_init(self.state, seed, iter_num, self.hyper["base_lr"], self.hyper["max_iterations"],
                      self.hyper["ema_decay"], self.hyper["consistency"], self.hyper["rampup"],
                      self.hyper["ramp_div"], self.hyper["cons_start_iter"])
        model.step_state = self.state
        ema_model.step_state = self.state
        model.rng_stream, ema_model.rng_stream = 1, 2   # seed-reproducible, distinct dropout streams
        self.momentum_buf = torch.zeros_like(model.flat_param)
        self.out = torch.zeros(16, dtype=torch.float32, device="cuda")
        self.iter_num = iter_num
        # a captured replay of a step that contains an RCCL collective is not verified on hardware: single-GPU only
        self.use_graph = bool(use_graph) and self.world == 1
        self.use_tape = (STEP_TAPE if use_tape is None else bool(use_tape)) and not self.use_graph
        self._graph = None
        self._static = None
        self._ema_in = None
        self._side = None
        self._bucketer = make_bucketer(model, process_group)

    # ---- the step (eager form; also what gets captured) ----
    def _run(self, volume, label, noise):
        L = self.labeled_bs
        unl = volume[L:]
        if self._ema_in is None or self._ema_in.shape != unl.shape:
            self._ema_in = torch.empty_like(unl)
        if noise is None:
            ops.teacher_noise(unl.contiguous(), self._ema_in, self.state)
        else:
            torch.add(unl, noise, out=self._ema_in)     # injected noise: parity tests only
        if TWO_STREAM:
            # the two forwards are independent: the teacher's (half the batch, no backward) runs on a side stream
            # and fills the CUs the student's small deep layers leave idle
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _lib.side_stream("side")
            _lib.wait_stream(self._side, main)
            with torch.cuda.stream(self._side):
                t_logits = self.ema_model.forward_raw(self._ema_in, no_backward=True)
            s_logits = self.model.forward_raw(volume)
            _lib.wait_stream(main, self._side)
        else:
            s_logits = self.model.forward_raw(volume)
            t_logits = self.ema_model.forward_raw(self._ema_in, no_backward=True)
        ops.loss_tail(s_logits, t_logits, label[:L].contiguous(), L, self.out,
                      dlogits=self.model.logits_grad_buffer(), state=self.state)
        grad_scale = backward_and_sync(self.model, self.pg, self._bucketer)   # the step's only exchange
        ops.sgd_ema_step(self.model.flat_param, self.model.flat_grad, self.momentum_buf,
                         self.ema_model.flat_param, momentum=self.momentum, weight_decay=self.weight_decay,
                         grad_scale=grad_scale, state=self.state)
        h = self.hyper
        ops.step_advance(self.state, h["base_lr"], h["max_iterations"], h["ema_decay"], h["consistency"],
                         h["rampup"], h["ramp_div"], h["cons_start_iter"])

    def step(self, volume_batch, label_batch, noise=None):
        """One iteration on device tensors; returns the device scalar buffer
        ``[loss, loss_ce, loss_dice, consistency_loss, consistency_weight, ...]`` (no host sync)."""
        if not self.model.training or not self.ema_model.training:
            raise RuntimeError("Mean-Teacher step runs both networks in train mode (reference never calls .eval())")
        if self.use_graph and noise is None:
            self._step_graph(volume_batch, label_batch)
        elif self.use_tape and noise is None and type(self) is MeanTeacherTrainer:
            self._tape_step(lambda v, l: self._run(v, l, None), (volume_batch, label_batch))
        else:
            self._run(volume_batch, label_batch, noise)
        self.iter_num += 1
        return self.out

    # ---- hipGraph capture / replay ----
    def _step_graph(self, volume, label):
        if self._graph is None:
            self._static = (volume.clone(), label.clone())
            # one eager warm-up builds plans and scratch buffers outside the capture
            # (it is a real training step on the real batch)
            self._run(self._static[0], self._static[1], None)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            self._pending_first = True
            with torch.cuda.graph(self._graph):
                self._run(self._static[0], self._static[1], None)
            # capture does not execute: the warm-up above WAS this call's step
            return
        self._static[0].copy_(volume)
        self._static[1].copy_(label)
        self._graph.replay()

    def losses(self):
        """Host copy of the last step's scalars (one small D2H)."""
        o = self.out.cpu()
        return dict(loss=o[0].item(), loss_ce=o[1].item(), loss_dice=o[2].item(),
                    consistency_loss=o[3].item(), consistency_weight=o[4].item())


class UAMTTrainer(MeanTeacherTrainer):
    """Uncertainty-aware Mean Teacher (reference code/train_uncertainty_aware_mean_teacher_3D.py:134-189,
    code/train_uncertainty_aware_mean_teacher_2D.py:146-201): the Mean-Teacher step plus T = 8 MC-dropout
    teacher predictions (4 forwards on ``repeat(unlabeled, 2)`` with fresh noise), whose mean-probability
    entropy masks the consistency term.  Every teacher forward runs in train mode (dropout active, BatchNorm
    running statistics updated 5 times per step, as in the reference)."""

    T = 8

    def __init__(self, *args, **kw):
        kw.pop("use_graph", None)
        kw["use_tape"] = False          # (the MC passes cycle the teacher through RNG sub-streams set from Python: eager)
        super().__init__(*args, **kw)
        self._rep_in = None
        self._mean_probs = None

    def _run(self, volume, label, noise, mc_noise=None):
        L = self.labeled_bs
        unl = volume[L:].contiguous()
        U = unl.shape[0]
        if self._ema_in is None or self._ema_in.shape != unl.shape:
            self._ema_in = torch.empty_like(unl)
            self._rep_in = torch.empty((2 * U,) + tuple(unl.shape[1:]), dtype=unl.dtype, device=unl.device)
        if noise is None:
            ops.teacher_noise(unl, self._ema_in, self.state)
        else:
            torch.add(unl, noise, out=self._ema_in)
        if self._mean_probs is None or self._mean_probs.shape[0] != U:
            self._mean_probs = None

        def teacher_passes():
            self.ema_model.rng_stream = 2
            t_logits = self.ema_model.forward_raw(self._ema_in, no_backward=True)
            if self._mean_probs is None or self._mean_probs.shape != t_logits.shape:
                self._mean_probs = torch.empty_like(t_logits)
            for i in range(self.T // 2):
                for r in range(2):
                    half = self._rep_in[r * U:(r + 1) * U]
                    if mc_noise is None:
                        ops.teacher_noise(unl, half, self.state, salt=0x7EAC4E5 + 1 + 2 * i + r)
                    else:
                        torch.add(unl, mc_noise[i][r * U:(r + 1) * U], out=half)
                self.ema_model.rng_stream = 3 + i          # a fresh dropout stream per MC pass
                mc_logits = self.ema_model.forward_raw(self._rep_in, no_backward=True)
                ops.softmax_mean_accumulate(mc_logits, self._mean_probs, 2, 1.0 / self.T, first=(i == 0))
            self.ema_model.rng_stream = 2
            return t_logits

        if TWO_STREAM:
            # the five teacher forwards (one plain, four MC passes: sequential, they share the teacher's buffers) only meet
            # the student in the loss tail: they run on a side stream beside the student's forward (bit-identical)
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _lib.side_stream("side")
            _lib.wait_stream(self._side, main)
            with torch.cuda.stream(self._side):
                t_logits = teacher_passes()
            s_logits = self.model.forward_raw(volume)
            _lib.wait_stream(main, self._side)
        else:
            s_logits = self.model.forward_raw(volume)
            t_logits = teacher_passes()
        ops.uamt_tail(s_logits, t_logits, self._mean_probs, label[:L].contiguous(), L, self.out,
                      self.hyper["max_iterations"], dlogits=self.model.logits_grad_buffer(), state=self.state)
        grad_scale = backward_and_sync(self.model, self.pg, self._bucketer)
        ops.sgd_ema_step(self.model.flat_param, self.model.flat_grad, self.momentum_buf,
                         self.ema_model.flat_param, momentum=self.momentum, weight_decay=self.weight_decay,
                         grad_scale=grad_scale, state=self.state)
        h = self.hyper
        ops.step_advance(self.state, h["base_lr"], h["max_iterations"], h["ema_decay"], h["consistency"],
                         h["rampup"], h["ramp_div"], h["cons_start_iter"])

    def step(self, volume_batch, label_batch, noise=None, mc_noise=None):
        if not self.model.training or not self.ema_model.training:
            raise RuntimeError("UA-MT runs both networks in train mode (MC dropout needs the teacher's dropout)")
        self._run(volume_batch, label_batch, noise, mc_noise)
        self.iter_num += 1
        return self.out

    def losses(self):
        d = super().losses()
        o = self.out.cpu()
        C = self.num_classes
        d.update(unmasked_voxels=o[5 + C].item(), threshold=o[6 + C].item())
        return d


class CrossTeachingTrainer(_TapedStep):
    """Cross teaching between a CNN and a Transformer (reference
    code/train_cross_teaching_between_cnn_transformer_2D.py:216-263): two students see the whole batch, each is
    supervised on the labeled half and by the OTHER network's arg-max pseudo labels (Dice) on the unlabeled
    half; ``loss = model1_loss + model2_loss``, two SGD steps, no EMA, no noise.  The learning rate follows the
    post-increment rule of that script (:257-263)."""

    def __init__(self, model1, model2, *, labeled_bs, num_classes, base_lr=0.01, max_iterations=30000,
                 consistency=0.1, consistency_rampup=200.0, seed=1337, iter_num=0, momentum=0.9, weight_decay=1e-4,
                 process_group=None, pseudo_ce=False, use_tape=None):
        self.use_tape = STEP_TAPE if use_tape is None else bool(use_tape)
        # pseudo_ce=True: cross pseudo supervision (code/train_cross_pseudo_supervision_{2D,3D}.py): the same step
        # with a cross-entropy pseudo-supervision term instead of Dice
        self.pseudo_ce = bool(pseudo_ce)
        self.model1, self.model2 = model1, model2
        self.labeled_bs, self.num_classes = labeled_bs, num_classes
        self.hyper = dict(base_lr=float(base_lr), max_iterations=float(max_iterations), ema_decay=0.0,
                          consistency=float(consistency), rampup=float(consistency_rampup), ramp_div=150,
                          cons_start_iter=0, lr_post_increment=True)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.pg = process_group
        self.state = ops.new_step_state()
        h = self.hyper
        ops.step_init(self.state, seed, iter_num, h["base_lr"], h["max_iterations"], h["ema_decay"], h["consistency"],
                      h["rampup"], h["ramp_div"], h["cons_start_iter"], h["lr_post_increment"])
        for i, m in enumerate((model1, model2)):
            m.step_state = self.state
            m.rng_stream = 1 + i
        self.mom1 = torch.zeros_like(model1.flat_param)
        self.mom2 = torch.zeros_like(model2.flat_param)
        self.out1 = torch.zeros(16, dtype=torch.float32, device="cuda")
        self.out2 = torch.zeros(16, dtype=torch.float32, device="cuda")
        self.iter_num = iter_num
        self._side = None
        # model2's backward is enqueued first (side stream): its exposed tail bucket is issued by finish(), after model1's
        # buckets, so that the in-order RCCL stream does not hold model1's early buckets behind the end of model2's backward
        self._bucketers = (make_bucketer(model1, process_group),
                           make_bucketer(model2, process_group, defer_tail=TWO_STREAM))

    def step(self, volume_batch, label_batch):
        if not (self.model1.training and self.model2.training):
            raise RuntimeError("cross teaching trains both networks (train mode)")
        if self.use_tape:
            self._tape_step(self._run, (volume_batch, label_batch))
        else:
            self._run(volume_batch, label_batch)
        self.iter_num += 1
        return self.out1, self.out2

    def _run(self, volume_batch, label_batch):
        L = self.labeled_bs
        lab = label_batch[:L].contiguous()
        if TWO_STREAM:
            # the two students only meet in the loss tails: model2's forward and backward run on a side stream
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _lib.side_stream("side")
            _lib.wait_stream(self._side, main)
            with torch.cuda.stream(self._side):
                o2 = self.model2.forward_raw(volume_batch)
            o1 = self.model1.forward_raw(volume_batch)
            _lib.wait_stream(main, self._side)
        else:
            o1 = self.model1.forward_raw(volume_batch)
            o2 = self.model2.forward_raw(volume_batch)
        ops.cross_teaching_tail(o1, o2, lab, L, self.out1, dlogits=self.model1.logits_grad_buffer(), state=self.state,
                                pseudo_ce=self.pseudo_ce)
        ops.cross_teaching_tail(o2, o1, lab, L, self.out2, dlogits=self.model2.logits_grad_buffer(), state=self.state,
                                pseudo_ce=self.pseudo_ce)
        if TWO_STREAM and self._bucketers[0] is None:
            _lib.wait_stream(self._side, main)
            with torch.cuda.stream(self._side):
                self.model2.backward_raw()
            self.model1.backward_raw()
            _lib.wait_stream(main, self._side)
            scales = [_lib.tape_call(dist.sync_gradients, m.flat_grad, self.pg) for m in (self.model1, self.model2)]
        elif self._bucketers[0] is not None:
            # both students' buckets are in flight while the other student's backward runs; one wait at the end.  With
            # TWO_STREAM the second student's backward runs on the side stream as on one GPU: its collectives are still
            # ENQUEUED by this thread in program order (model2's buckets, then model1's), the same on every rank.
            b1, b2 = self._bucketers
            _lib.tape_call(b1.begin)
            _lib.tape_call(b2.begin)
            if TWO_STREAM:
                _lib.wait_stream(self._side, main)
                with torch.cuda.stream(self._side):
                    self.model2.backward_raw(on_progress=b2.advance)
                self.model1.backward_raw(on_progress=b1.advance)
                _lib.wait_stream(main, self._side)
                scales = [_lib.tape_call(b1.finish), _lib.tape_call(b2.finish)]      # b1's tail went out with its backward; b2's goes now
            else:
                self.model1.backward_raw(on_progress=b1.advance)
                _lib.tape_call(b1.advance, 0, True)
                self.model2.backward_raw(on_progress=b2.advance)
                scales = [_lib.tape_call(b1.finish), _lib.tape_call(b2.finish)]
        else:
            self.model1.backward_raw()
            self.model2.backward_raw()
            scales = [_lib.tape_call(dist.sync_gradients, m.flat_grad, self.pg) for m in (self.model1, self.model2)]
        for m, mom, scale in ((self.model1, self.mom1, scales[0]), (self.model2, self.mom2, scales[1])):
            ops.sgd_ema_step(m.flat_param, m.flat_grad, mom, None, momentum=self.momentum,
                             weight_decay=self.weight_decay, grad_scale=scale, state=self.state)
        h = self.hyper
        ops.step_advance(self.state, h["base_lr"], h["max_iterations"], h["ema_decay"], h["consistency"], h["rampup"],
                         h["ramp_div"], h["cons_start_iter"], h["lr_post_increment"])

    def losses(self):
        a, b = self.out1.cpu(), self.out2.cpu()
        return dict(loss=a[0].item() + b[0].item(), model1_loss=a[0].item(), model2_loss=b[0].item(),
                    loss1_ce=a[1].item(), loss1_dice=a[2].item(), pseudo_supervision1=a[3].item(),
                    loss2_ce=b[1].item(), loss2_dice=b[2].item(), pseudo_supervision2=b[3].item(),
                    consistency_weight=a[4].item())


def linear_rampup(current, rampup_length):
    """reference code/utils/ramps.py:49-55"""
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


class CnnMeetVitTrainer(_TapedStep):
    """CNN student + Transformer student + EMA Transformer teacher (reference code/train_cnn_meet_vit_2D.py:293-352).

    ``model1`` (CNN) and ``model2`` (SwinUnet) see the whole batch and cross-teach through Dice on each other's
    arg-max pseudo labels with weight ``7 * consistency * linear_rampup(iter_num // 150, rampup)`` (:322-323,
    :336-337); both are also pulled towards ``ema_model`` -- the EMA of ``model2`` (:345), fed the noised unlabeled
    half (:298-309) -- by a softmax-MSE term with weight ``consistency * linear_rampup(...)`` that is zero while
    ``iter_num < 1000`` (:326-333).  ``loss = model1_loss + model2_loss``, two SGD steps, learning rate computed
    before ``iter_num`` is incremented (:347-348).  The two ramp weights are host floats of ``iter_num`` (no host
    sync: ``iter_num`` is the trainer's own counter)."""

    def __init__(self, model1, model2, ema_model, *, labeled_bs, num_classes, base_lr=0.01, max_iterations=30000,
                 ema_decay=0.99, consistency=0.1, consistency_rampup=200.0, seed=1337, iter_num=0, momentum=0.9,
                 weight_decay=1e-4, process_group=None, use_tape=None):
        if model2.flat_param.numel() != ema_model.flat_param.numel():
            raise RuntimeError("the teacher is the EMA of model2: same architecture required")
        # the two ramp weights are HOST floats of iter_num and arguments of the loss tails: the tape is recorded again whenever
        # they change (every ramp_div = 150 iterations, and at iteration 1000)
        self.use_tape = STEP_TAPE if use_tape is None else bool(use_tape)
        self._tape_weights = None
        self.model1, self.model2, self.ema_model = model1, model2, ema_model
        self.labeled_bs, self.num_classes = labeled_bs, num_classes
        self.hyper = dict(base_lr=float(base_lr), max_iterations=float(max_iterations), ema_decay=float(ema_decay),
                          consistency=float(consistency), rampup=float(consistency_rampup), ramp_div=150,
                          cons_start_iter=1000)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.pg = process_group
        self.state = ops.new_step_state()
        h = self.hyper
        ops.step_init(self.state, seed, iter_num, h["base_lr"], h["max_iterations"], h["ema_decay"], h["consistency"],
                      h["rampup"], h["ramp_div"], h["cons_start_iter"])
        for i, m in enumerate((model1, model2, ema_model)):
            m.step_state = self.state
            m.rng_stream = 1 + i
        self.mom1 = torch.zeros_like(model1.flat_param)
        self.mom2 = torch.zeros_like(model2.flat_param)
        self.out1 = torch.zeros(16, dtype=torch.float32, device="cuda")
        self.out2 = torch.zeros(16, dtype=torch.float32, device="cuda")
        self.iter_num = iter_num
        self._ema_in = None
        self._side = None
        self._bucketers = (make_bucketer(model1, process_group, defer_tail=TWO_STREAM),
                           make_bucketer(model2, process_group))

    def weights(self):
        """(pseudo-supervision weight, mean-teacher weight) of the current iteration"""
        h = self.hyper
        w = h["consistency"] * linear_rampup(self.iter_num // h["ramp_div"], h["rampup"])
        return 7 * w, (w if self.iter_num >= h["cons_start_iter"] else 0.0)

    def step(self, volume_batch, label_batch, noise=None):
        if not (self.model1.training and self.model2.training and self.ema_model.training):
            raise RuntimeError("train_cnn_meet_vit runs all three networks in train mode")
        if self.use_tape and noise is None:
            w = self.weights()
            if self._tape is not None and w != self._tape_weights:
                self._tape, self._tape_warm = None, self.TAPE_WARMUP       # plans and buffers are warm: record at once
            if self._tape is None and self._tape_warm >= self.TAPE_WARMUP:
                self._tape_weights = w
            self._tape_step(lambda v, l: self._run(v, l, None), (volume_batch, label_batch))
        else:
            self._run(volume_batch, label_batch, noise)
        self.iter_num += 1
        return self.out1, self.out2

    def _run(self, volume_batch, label_batch, noise):
        L = self.labeled_bs
        unl = volume_batch[L:].contiguous()
        if self._ema_in is None or self._ema_in.shape != unl.shape:
            self._ema_in = torch.empty_like(unl)
        if noise is None:
            ops.teacher_noise(unl, self._ema_in, self.state)
        else:
            torch.add(unl, noise, out=self._ema_in)     # injected noise: parity tests only
        two = TWO_STREAM
        if two:
            # three independent forwards: the CNN student and the (half-batch) teacher on a side stream beside the
            # Transformer student -- roughly equal work; later the CNN's backward beside the Transformer's.  Bit-identical
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _lib.side_stream("side")
            _lib.wait_stream(self._side, main)
            with torch.cuda.stream(self._side):
                o1 = self.model1.forward_raw(volume_batch)
                t = self.ema_model.forward_raw(self._ema_in, no_backward=True)
            o2 = self.model2.forward_raw(volume_batch)
            _lib.wait_stream(main, self._side)
        else:
            o1 = self.model1.forward_raw(volume_batch)
            o2 = self.model2.forward_raw(volume_batch)
            t = self.ema_model.forward_raw(self._ema_in, no_backward=True)
        lab = label_batch[:L].contiguous()
        w_cps, w_mt = self.weights()
        ops.cross_teaching_tail(o1, o2, lab, L, self.out1, dlogits=self.model1.logits_grad_buffer(),
                                cons_weight=w_cps, teacher=t, mt_weight=w_mt)
        ops.cross_teaching_tail(o2, o1, lab, L, self.out2, dlogits=self.model2.logits_grad_buffer(),
                                cons_weight=w_cps, teacher=t, mt_weight=w_mt)
        b1, b2 = self._bucketers
        if two:
            # model1's backward is enqueued first, on the side stream: with bucketers its tail bucket is deferred to
            # finish() (defer_tail), as for the side-stream student of cross teaching
            _lib.wait_stream(self._side, main)
            if b1 is not None:
                _lib.tape_call(b1.begin)
                _lib.tape_call(b2.begin)
            with torch.cuda.stream(self._side):
                self.model1.backward_raw(on_progress=None if b1 is None else b1.advance)
            self.model2.backward_raw(on_progress=None if b2 is None else b2.advance)
            _lib.wait_stream(main, self._side)
            if b1 is not None:
                scales = [_lib.tape_call(b1.finish), _lib.tape_call(b2.finish)]
            else:
                scales = [_lib.tape_call(dist.sync_gradients, m.flat_grad, self.pg) for m in (self.model1, self.model2)]
        else:
            scales = [backward_and_sync(self.model1, self.pg, b1), backward_and_sync(self.model2, self.pg, b2)]
        for m, mom, ema, scale in ((self.model1, self.mom1, None, scales[0]),
                                   (self.model2, self.mom2, self.ema_model.flat_param, scales[1])):
            ops.sgd_ema_step(m.flat_param, m.flat_grad, mom, ema, momentum=self.momentum,
                             weight_decay=self.weight_decay, grad_scale=scale, state=self.state)
        h = self.hyper
        ops.step_advance(self.state, h["base_lr"], h["max_iterations"], h["ema_decay"], h["consistency"], h["rampup"],
                         h["ramp_div"], h["cons_start_iter"])

    def losses(self):
        a, b = self.out1.cpu(), self.out2.cpu()
        return dict(loss=a[0].item() + b[0].item(), model1_loss=a[0].item(), model2_loss=b[0].item(),
                    loss1_ce=a[1].item(), loss1_dice=a[2].item(), pseudo_supervision1=a[3].item(),
                    consistency_loss1=a[5].item(), loss2_ce=b[1].item(), loss2_dice=b[2].item(),
                    pseudo_supervision2=b[3].item(), consistency_loss2=b[5].item(),
                    consistency_weight=a[4].item() / 7.0, mt_weight=a[6].item())

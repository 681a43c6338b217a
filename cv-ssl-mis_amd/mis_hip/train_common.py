"""Shared driver of the ``train_mean_teacher_{2D,3D}.py`` command lines.

Keeps what the reference scripts do around the hot loop (seeding, snapshot directory, ``log.txt``,
periodic checkpoints with the reference's file names, the two-stream "labeled first" batch contract)
and replaces the loop body by ``MeanTeacherTrainer.step``.  Batches come from the dataset under ``--root_path``
held resident in HBM (dataloaders/: two-stream sampler + one augmentation gather launch per batch) or, when no
dataset is there, from a synthetic two-stream source with the reference's shapes/dtypes.
"""
import logging
import os
import random
import sys
import time

import numpy as np
import torch


class SyntheticTwoStream:
    """Stands in for DataLoader(BaseDataSets / BraTS2019, TwoStreamBatchSampler): yields
    ``{'image': f32 [B,1,*patch], 'label': u8|i64 [B,*patch]}`` with the labeled samples FIRST
    (reference code/dataloaders/dataset.py:247-294).  Data are resident on the device."""

    def __init__(self, batch_size, patch_size, num_classes, label_dtype, seed, pool=4):
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.batches = []
        for _ in range(pool):
            img = torch.rand((batch_size, 1) + tuple(patch_size), generator=g, device="cuda")
            lab = torch.randint(0, num_classes, (batch_size,) + tuple(patch_size), generator=g,
                                device="cuda").to(label_dtype)
            self.batches.append({"image": img, "label": lab})
        self.i = 0

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for b in self.batches:
            yield b


def patients_to_slices(dataset, patiens_num):
    """Number of labeled slices for a number of labeled patients (reference train_mean_teacher_2D.py:106-116; like
    there, every dataset name without "ACDC" gets the Prostate table)."""
    if "ACDC" in dataset:
        ref_dict = {"3": 68, "7": 136, "14": 256, "21": 396, "28": 512, "35": 664, "140": 1312}
    else:
        ref_dict = {"2": 27, "4": 53, "8": 120, "12": 179, "16": 256, "21": 312, "42": 623}
    return ref_dict[str(patiens_num)]


def check_shards(n_labeled, n_unlabeled, labeled_bs, unlabeled_bs, world):
    """Every rank's labeled / unlabeled shard (``idxs[rank::world]``) must hold at least one batch, or
    ``TwoStreamBatchSampler`` asserts on SOME ranks only and the others hang in the first all-reduce.  The sizes are
    a pure function of (len, world), so every rank raises the same error together."""
    small = [(r, len(range(r, n_labeled, world)), len(range(r, n_unlabeled, world))) for r in range(world)]
    bad = [(r, l, u) for r, l, u in small if l < labeled_bs or u < unlabeled_bs]
    if bad:
        r, l, u = bad[0]
        raise RuntimeError(
            f"data-parallel shards too small for world={world}: rank {r} would get {l} labeled / {u} unlabeled "
            f"samples but a batch needs {labeled_bs} + {unlabeled_bs} (per GPU).  Use fewer GPUs, a smaller "
            f"--labeled_bs / --batch_size, or more labeled data (--labeled_num).")


def make_loader(args, label_dtype, rank, world=1):
    """The training batches: the dataset under ``--root_path`` resident in HBM with the reference's two-stream sampler
    and augmentation as one gather launch per batch (dataloaders/), or -- when no dataset is there (the list file
    ``train_slices.list`` / ``train.txt`` is missing) -- synthetic resident batches of the same shapes/dtypes."""
    three_d = len(args.patch_size) == 3
    listfile = os.path.join(args.root_path, "train.txt" if three_d else "train_slices.list")
    if not os.path.exists(listfile):
        return SyntheticTwoStream(args.batch_size, args.patch_size, args.num_classes, label_dtype,
                                  args.seed + 1000 * rank), "synthetic two-stream source (no dataset at %s)" % args.root_path
    if rank:                                   # data parallel: every rank draws its own batches / augmentations
        random.seed(args.seed + rank)
        np.random.seed(args.seed + rank)
    if three_d:
        from dataloaders.brats2019 import (BraTS2019, DeviceTwoStreamLoader3D, DeviceVolumePool, RandomRotFlipCrop,
                                           TwoStreamBatchSampler)
        db_train = BraTS2019(base_dir=args.root_path, split='train', num=None)
        check_shards(args.labeled_num, len(db_train) - args.labeled_num, args.labeled_bs,
                     args.batch_size - args.labeled_bs, world)
        labeled = list(range(0, args.labeled_num))[rank::world]           # train_mean_teacher_3D.py:109-112;
        unlabeled = list(range(args.labeled_num, len(db_train)))[rank::world]   # disjoint shard per rank (SURVEY s.8e)
        sampler = TwoStreamBatchSampler(labeled, unlabeled, args.batch_size, args.batch_size - args.labeled_bs)
        loader = DeviceTwoStreamLoader3D(DeviceVolumePool.from_dataset(db_train), sampler,
                                         RandomRotFlipCrop(args.patch_size), label_dtype=label_dtype)
    else:
        from dataloaders.dataset import (BaseDataSets, DeviceSlicePool, DeviceTwoStreamLoader, RandomGenerator,
                                         TwoStreamBatchSampler)
        db_train = BaseDataSets(base_dir=args.root_path, split="train", num=None)
        labeled_slice = patients_to_slices(args.root_path, args.labeled_num)   # train_mean_teacher_2D.py:172-180
        check_shards(labeled_slice, len(db_train) - labeled_slice, args.labeled_bs,
                     args.batch_size - args.labeled_bs, world)
        labeled = list(range(0, labeled_slice))[rank::world]
        unlabeled = list(range(labeled_slice, len(db_train)))[rank::world]
        sampler = TwoStreamBatchSampler(labeled, unlabeled, args.batch_size, args.batch_size - args.labeled_bs)
        loader = DeviceTwoStreamLoader(DeviceSlicePool.from_dataset(db_train), sampler,
                                       RandomGenerator(args.patch_size))
    return loader, "%d cases of %s resident in HBM, %d labeled" % (len(db_train), args.root_path, len(labeled))


def setup_distributed():
    """One process per GPU (torchrun): returns (rank, world, local_rank); initialises RCCL if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        # this rank's threads next to its GPU's PCIe root: its share of the cores of the device's NUMA node
        from . import dist as _dist
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
        if local_world <= 0 and world <= torch.cuda.device_count():
            local_world = world                               # one node, launched without torchrun's LOCAL_WORLD_SIZE
        if local_world > 0:
            pin = _dist.pin_rank_to_numa(local_rank, local_world, _dist.device_bus_ids(min(local_world,
                                                                                             torch.cuda.device_count())))
        else:                                                 # several nodes and no LOCAL_WORLD_SIZE: the share is unknown
            pin = dict(applied=False, reason="LOCAL_WORLD_SIZE unset and WORLD_SIZE > local device count")
        logging.info("rank %d: CPU affinity %s", rank, pin)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return rank, world, local_rank


def broadcast_model_state(*models, src=0):
    """Every rank starts from (or resumes with) rank ``src``'s state: the flat parameter buffer AND the floating-point
    buffers (BatchNorm running statistics -- per-rank during training, rank 0's are the canonical ones a checkpoint holds)."""
    for m in models:
        if m is None:
            continue
        torch.distributed.broadcast(m.flat_param, src)
        bufs = [b for b in m.buffers() if b.is_floating_point() and b.numel()]
        if bufs:
            flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
            torch.distributed.broadcast(flat, src)
            off = 0
            with torch.no_grad():
                for b in bufs:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()


def _init_conv_weights(model, init_fn):
    """Shared body of the reference's ``kaiming_normal_init_weight`` / ``xavier_normal_init_weight``
    (train_cross_pseudo_supervision_3D.py:79-96, train_cross_pseudo_supervision_2D.py:85-102): ``init_fn`` on the
    weight of every nn.Conv2d / nn.Conv3d (NOT nn.ConvTranspose3d: ``isinstance(m, nn.Conv3d)`` is false for it),
    BatchNorm weight := 1, bias := 0.  Written in place through the flat-parameter views; draws from torch's global
    CPU generator in module order like the reference."""
    buffers = {n for n, _ in model.named_buffers()}
    transposed = getattr(model, "transposed_convs", ())
    for name, p in model.named_parameters():
        if p.dim() >= 4 and name.endswith(".weight") and name not in transposed:
            w = torch.empty(tuple(p.shape))
            init_fn(w)
            p.data.copy_(w)
        elif name.endswith(".weight") and p.dim() == 1 and name[:-len("weight")] + "running_mean" in buffers:
            p.data.fill_(1.0)
        elif name.endswith(".bias") and p.dim() == 1 and name[:-len("bias")] + "running_mean" in buffers:
            p.data.zero_()
    return model


def kaiming_normal_init_weight(model):
    return _init_conv_weights(model, torch.nn.init.kaiming_normal_)


def xavier_normal_init_weight(model):
    return _init_conv_weights(model, torch.nn.init.xavier_normal_)


def seed_everything(args):
    """reference train_mean_teacher_2D.py:316-326"""
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)


class ScalarLog:
    """Tensorboard-free sink for the per-step scalars the reference sends to ``SummaryWriter.add_scalar``
    (train_mean_teacher_2D.py:239-250: info/lr, info/total_loss, info/loss_ce, info/loss_dice,
    info/consistency_loss, info/consistency_weight; the validation scalars :270-279): one CSV row
    ``iter_num,tag,value`` per call in ``<snapshot>/scalars.csv`` (tensorboardX is not in this image)."""

    def __init__(self, snapshot_path, enabled=True):
        self.f = open(os.path.join(snapshot_path, "scalars.csv"), "a") if enabled else None

    def add_scalar(self, tag, value, iter_num):
        if self.f is not None:
            self.f.write("%d,%s,%.9g\n" % (iter_num, tag, float(value)))

    def close(self):
        if self.f is not None:
            self.f.close()
            self.f = None


class Validator:
    """The in-training validation of the reference: every 200 iterations (``iter_num > 0 and iter_num % 200 == 0``)
    the model goes to eval mode and is scored on the validation split -- 2-D: ``val_2D.test_single_volume`` over
    ``BaseDataSets(split='val')`` (train_mean_teacher_2D.py:262-294), 3-D: ``val_3D.test_all_case(...,
    test_list='val.txt', stride_xy=64, stride_z=64)`` (train_mean_teacher_3D.py:201-222) -- and a new best mean Dice
    writes ``<prefix>iter_{n}_dice_{d}.pth`` and ``{model}_best_<name>.pth``.  Data-parallel runs (the reference is
    single-GPU) shard the validation cases over the ranks (``cases[rank::world]``) and all-reduce the metric sums: no
    rank waits in the next step's gradient all-reduce for a whole validation pass on rank 0 (RCCL's watchdog would
    abort a long 3-D one).  Every rank holds identical weights; the BatchNorm running statistics -- per-rank, there is
    no SyncBN -- are taken from rank 0 for the scoring, as the checkpoint is, and every rank's OWN running statistics are
    put back afterwards: validation never changes the training state, the ranks' running statistics evolve independently for the
    whole run (standard DDP without SyncBN) and rank 0 is canonical -- its buffers are what checkpoints hold and what
    ``resume`` broadcasts (train_common.broadcast_model_state).  Scalars, logs and checkpoints are rank 0's.  When the validation list file is missing (the synthetic runs) nothing is scored and ``finish`` writes
    the final weights under the best-model name, so that the inference CLIs always find their checkpoint."""

    def __init__(self, args, snapshot_path, scalars=None, rank=0, world=1, process_group=None):
        self.args, self.snapshot_path, self.scalars = args, snapshot_path, scalars
        self.rank, self.world = rank, world
        self.pg = process_group          # the trainer's group: validation collectives must run where the gradients do
        self.three_d = len(args.patch_size) == 3
        self.best = {}
        self.db_val = None
        listfile = os.path.join(args.root_path, "val.txt" if self.three_d else "val.list")
        self.available = os.path.exists(listfile)
        if self.available and not self.three_d:
            from dataloaders.dataset import BaseDataSets
            self.db_val = BaseDataSets(base_dir=args.root_path, split="val")

    def score(self, model):
        """(mean Dice, mean HD95, per-class [[dice, hd95], ...]) of ``model`` on the validation split."""
        args = self.args
        saved = None
        if self.world > 1:
            # score with rank 0's BatchNorm running statistics on every rank: ONE broadcast of the floating-point buffers
            # packed into a flat tensor (dozens of tiny collectives per model before).  The rank's own statistics are put
            # back after the scoring -- validation must not change the training state (per-rank running statistics, no
            # SyncBN, are what the reference's single-GPU semantics give every rank).
            bufs = [b for b in model.buffers() if b.is_floating_point() and b.numel()]
            if bufs:
                saved = [b.detach().clone() for b in bufs]
                flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
                src = torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0
                torch.distributed.broadcast(flat, src, group=self.pg)
                off = 0
                with torch.no_grad():
                    for b in bufs:
                        b.copy_(flat[off:off + b.numel()].view_as(b))
                        off += b.numel()
        err = None
        try:
            total, count = self._score_shard(model)
        except Exception as e:       # keep the collective below symmetric: the other ranks must not wait for this one
            if self.world <= 1:
                raise
            err, total, count = e, np.zeros((args.num_classes - 1, 2)), 0
        finally:
            if saved is not None:
                with torch.no_grad():
                    for b, v in zip(bufs, saved):
                        b.copy_(v)
        m = self._reduce(total, count, failed=err is not None)
        if err is not None:
            raise err
        return float(np.mean(m, axis=0)[0]), float(np.mean(m, axis=0)[1]), m

    def _score_shard(self, model):
        args = self.args
        if self.three_d:
            from val_3D import test_all_case
            total, count = test_all_case(model, args.root_path, test_list="val.txt", num_classes=args.num_classes,
                                         patch_size=args.patch_size, stride_xy=64, stride_z=64,
                                         shard=(self.rank, self.world))
        else:
            from val_2D import test_single_volume
            total, count = np.zeros((args.num_classes - 1, 2)), 0
            for i in range(self.rank, len(self.db_val), self.world):
                s = self.db_val[i]
                image = torch.from_numpy(np.asarray(s["image"])).unsqueeze(0)
                label = torch.from_numpy(np.asarray(s["label"])).unsqueeze(0)
                total = total + np.array(test_single_volume(image, label, model, classes=args.num_classes,
                                                            patch_size=args.patch_size), dtype=np.float64)
                count += 1
        return total, count

    def _reduce(self, total, count, failed=False):
        """Mean over all ranks' cases: all-reduce (sum) of the metric sums, the case count and an error flag -- a rank whose
        shard raised still takes part, and EVERY rank then fails (instead of the others blocking in this all-reduce until
        the process group's timeout)."""
        if self.world > 1:
            t = torch.tensor(np.concatenate([np.asarray(total, dtype=np.float64).ravel(), [float(count), float(failed)]]),
                             dtype=torch.float64,
                             device="cuda" if torch.distributed.get_backend(self.pg) == "nccl" else "cpu")
            torch.distributed.all_reduce(t, group=self.pg)
            t = t.cpu().numpy()
            if t[-1] > 0 and not failed:
                raise RuntimeError(f"validation failed on {int(t[-1])} other rank(s); see their logs")
            total, count = t[:-2].reshape(np.shape(total)), t[-2]
        return np.asarray(total, dtype=np.float64) / max(count, 1)

    def __call__(self, iter_num, models):
        """``models``: list of (tag, file prefix, best-file suffix, module), e.g. ('', '', 'best_model', model) or
        ('model1_', 'model1_', 'best_model1', model1)."""
        if not self.available or iter_num <= 0 or iter_num % 200:
            return
        for tag, prefix, best_name, model in models:
            was_training = model.training
            model.eval()
            perf, hd95, m = self.score(model)
            model.train(was_training)
            if self.rank != 0:
                continue
            if self.scalars is not None:
                for c in range(m.shape[0]):
                    self.scalars.add_scalar('info/%sval_%d_dice' % (tag, c + 1), m[c, 0], iter_num)
                    self.scalars.add_scalar('info/%sval_%d_hd95' % (tag, c + 1), m[c, 1], iter_num)
                self.scalars.add_scalar('info/%sval_mean_dice' % tag, perf, iter_num)
                self.scalars.add_scalar('info/%sval_mean_hd95' % tag, hd95, iter_num)
            if perf > self.best.get(best_name, 0.0):
                self.best[best_name] = perf
                torch.save(model.state_dict(), os.path.join(
                    self.snapshot_path, '%siter_%d_dice_%s.pth' % (prefix, iter_num, round(perf, 4))))
                torch.save(model.state_dict(), os.path.join(
                    self.snapshot_path, '%s_%s.pth' % (self.args.model, best_name)))
            if self.three_d:      # train_mean_teacher_3D.py:221-222
                logging.info('iteration %d : %sdice_score : %f %shd95 : %f' % (iter_num, tag, perf, tag, hd95))
            else:                 # train_mean_teacher_2D.py:292-293
                logging.info('iteration %d : %smean_dice : %f %smean_hd95 : %f' % (iter_num, tag, perf, tag, hd95))

    def finish(self, models):
        if self.rank != 0:
            return
        for tag, prefix, best_name, model in models:
            path = os.path.join(self.snapshot_path, '%s_%s.pth' % (self.args.model, best_name))
            if not os.path.exists(path):
                torch.save(model.state_dict(), path)
                logging.info("no validation score was recorded (%s): final weights saved as %s" %
                             ("no validation list under %s" % self.args.root_path if not self.available
                              else "fewer than 200 iterations or Dice 0", path))


def open_snapshot(args, rank, fmt="../model/{}_{}_labeled/{}"):
    """``fmt``: the snapshot directory pattern of the reference script being mirrored (it differs between scripts:
    train_mean_teacher_2D.py:328 has the ``_labeled`` suffix, train_mean_teacher_3D.py:252 and the cross-teaching /
    CPS scripts do not)."""
    snapshot_path = fmt.format(args.exp, args.labeled_num, args.model)
    if rank == 0:
        os.makedirs(snapshot_path, exist_ok=True)
        logging.basicConfig(filename=snapshot_path + "/log.txt", level=logging.INFO,
                            format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S', force=True)
        logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
        logging.info(str(args))
    return snapshot_path


def run_cross_teaching(args, make_model1, make_model2, log_every=1, label_dtype=torch.uint8, pseudo_ce=False,
                       make_ema=None, snapshot_fmt="../model/{}_{}/{}", init_fns=None):
    """Hot loop of train_cross_teaching_between_cnn_transformer_2D.py:208-300 (two students, no teacher); with
    ``pseudo_ce=True`` that of train_cross_pseudo_supervision_{2D,3D}.py (CE pseudo-supervision); with ``make_ema``
    (the EMA teacher of model2) that of train_cnn_meet_vit_2D.py:285-352."""
    from .step import CnnMeetVitTrainer, CrossTeachingTrainer
    rank, world, _ = setup_distributed()
    seed_everything(args)
    snapshot_path = open_snapshot(args, rank, snapshot_fmt)
    model1, model2 = make_model1(), make_model2()
    if init_fns is not None:          # e.g. kaiming_normal / xavier_normal of the CPS scripts
        init_fns[0](model1)
        init_fns[1](model2)
    ema_model = make_ema() if make_ema is not None else None
    if world > 1:
        broadcast_model_state(model1, model2)
    model1.train()
    model2.train()
    if ema_model is not None:
        for p in ema_model.parameters():
            p.detach_()
        if world > 1:
            broadcast_model_state(ema_model)
        ema_model.train()
        trainer = CnnMeetVitTrainer(model1, model2, ema_model, labeled_bs=args.labeled_bs,
                                    num_classes=args.num_classes, base_lr=args.base_lr,
                                    max_iterations=args.max_iterations, ema_decay=args.ema_decay,
                                    consistency=args.consistency, consistency_rampup=args.consistency_rampup,
                                    seed=args.seed + rank)
    else:
        trainer = CrossTeachingTrainer(model1, model2, labeled_bs=args.labeled_bs, num_classes=args.num_classes,
                                       base_lr=args.base_lr, max_iterations=args.max_iterations,
                                       consistency=args.consistency, consistency_rampup=args.consistency_rampup,
                                       seed=args.seed + rank, pseudo_ce=pseudo_ce)
    loader, source = make_loader(args, label_dtype, rank, world)
    if rank == 0:
        logging.info("{} iterations per epoch ({})".format(len(loader), source))
    scalars = ScalarLog(snapshot_path, enabled=(rank == 0))
    validator = Validator(args, snapshot_path, scalars, rank, world, process_group=trainer.pg)
    val_models = [('model1_', 'model1_', 'best_model1', model1), ('model2_', 'model2_', 'best_model2', model2)]
    if ema_model is not None:     # train_cnn_meet_vit_2D.py:441-468
        val_models.append(('ema_model_', 'ema_model_', 'best_ema_model', ema_model))
    iter_num, t0 = 0, time.time()
    max_epoch = args.max_iterations // len(loader) + 1
    for _epoch in range(max_epoch):
        for sampled_batch in loader:
            trainer.step(sampled_batch["image"], sampled_batch["label"])
            iter_num += 1
            if rank == 0 and iter_num % log_every == 0:
                s = trainer.losses()
                for k, v in s.items():
                    scalars.add_scalar('loss/' + k if k != 'consistency_weight' else 'consistency_weight/' + k, v,
                                       iter_num)
                logging.info('iteration %d : model1 loss : %f model2 loss : %f' %
                             (iter_num, s["model1_loss"], s["model2_loss"]))
            validator(iter_num, val_models)      # every rank: the validation cases are sharded over the ranks
            if rank == 0 and iter_num % 3000 == 0:
                for i, m in ((1, model1), (2, model2)):
                    path = os.path.join(snapshot_path, 'model%d_iter_%d.pth' % (i, iter_num))
                    torch.save(m.state_dict(), path)
                    logging.info("save model%d to %s" % (i, path))
            if iter_num >= args.max_iterations:
                break
        if iter_num >= args.max_iterations:
            break
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        logging.info("%d iterations in %.2f s (%.1f samples/s over %d GPU(s))" %
                     (iter_num, dt, iter_num * args.batch_size * world / dt, world))
        validator.finish(val_models)
        scalars.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return "Training Finished!"


def run_training(args, make_model, *, label_dtype, cons_start_iter, save_ema, log_every=1, trainer_cls=None,
                 snapshot_fmt="../model/{}_{}_labeled/{}"):
    """Hot loop of train_mean_teacher_2D.py:196-312 / train_mean_teacher_3D.py:128-230 (and, with
    ``trainer_cls=UAMTTrainer``, of train_uncertainty_aware_mean_teacher_{2D,3D}.py)."""
    from .step import MeanTeacherTrainer
    if trainer_cls is not None:
        MeanTeacherTrainer = trainer_cls
    rank, world, _ = setup_distributed()
    seed_everything(args)
    snapshot_path = open_snapshot(args, rank, snapshot_fmt)

    model = make_model()
    ema_model = make_model()
    for p in ema_model.parameters():       # create_model(ema=True): teacher params are detached
        p.detach_()
    if world > 1:                          # every rank starts from rank 0's weights and running statistics
        broadcast_model_state(model, ema_model)
    model.train()
    ema_model.train()

    trainer = MeanTeacherTrainer(model, ema_model, labeled_bs=args.labeled_bs, num_classes=args.num_classes,
                                 base_lr=args.base_lr, max_iterations=args.max_iterations, ema_decay=args.ema_decay,
                                 consistency=args.consistency, consistency_rampup=args.consistency_rampup,
                                 cons_start_iter=cons_start_iter, seed=args.seed + rank,
                                 # captured replay of a step that contains an RCCL collective is not verified on
                                 # hardware: --hip_graph is honoured for single-GPU runs only
                                 use_graph=bool(getattr(args, "hip_graph", 0)) and world == 1)
    if rank == 0 and getattr(args, "hip_graph", 0) and world > 1:
        logging.info("--hip_graph ignored for world_size %d (single-GPU only)" % world)
    loader, source = make_loader(args, label_dtype, rank, world)
    if rank == 0:
        logging.info("{} iterations per epoch ({})".format(len(loader), source))
    scalars = ScalarLog(snapshot_path, enabled=(rank == 0))
    validator = Validator(args, snapshot_path, scalars, rank, world, process_group=trainer.pg)
    val_models = [('', '', 'best_model', model)]
    iter_num = 0
    max_epoch = args.max_iterations // len(loader) + 1
    t0 = time.time()
    for _epoch in range(max_epoch):
        for sampled_batch in loader:
            trainer.step(sampled_batch["image"], sampled_batch["label"])
            iter_num += 1
            if rank == 0 and iter_num % log_every == 0:
                s = trainer.losses()          # the only device->host read of the step: ONE copy of trainer.out
                # all scalars of train_mean_teacher_2D.py:239-245.  lr_ is computed there (:234) BEFORE iter_num += 1
                # and logged under the incremented iter_num: base_lr * (1 - (n - 1) / max) ** 0.9 at tag n
                lr_ = args.base_lr * (1.0 - min(iter_num - 1, args.max_iterations) / args.max_iterations) ** 0.9
                scalars.add_scalar('info/lr', lr_, iter_num)
                scalars.add_scalar('info/total_loss', s["loss"], iter_num)
                scalars.add_scalar('info/loss_ce', s["loss_ce"], iter_num)
                scalars.add_scalar('info/loss_dice', s["loss_dice"], iter_num)
                scalars.add_scalar('info/consistency_loss', s["consistency_loss"], iter_num)
                scalars.add_scalar('info/consistency_weight', s["consistency_weight"], iter_num)
                logging.info('iteration %d : loss : %f, loss_ce: %f, loss_dice: %f' %
                             (iter_num, s["loss"], s["loss_ce"], s["loss_dice"]))
            validator(iter_num, val_models)      # every rank: the validation cases are sharded over the ranks
            if rank == 0 and iter_num % 3000 == 0:
                path = os.path.join(snapshot_path, 'iter_' + str(iter_num) + '.pth')
                torch.save(model.state_dict(), path)
                logging.info("save model to {}".format(path))
                if save_ema:
                    path = os.path.join(snapshot_path, 'ema_model_iter_' + str(iter_num) + '.pth')
                    torch.save(ema_model.state_dict(), path)
                    logging.info("save ema_model to {}".format(path))
            if iter_num >= args.max_iterations:
                break
        if iter_num >= args.max_iterations:
            break
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        logging.info("%d iterations in %.2f s (%.1f samples/s over %d GPU(s))" %
                     (iter_num, dt, iter_num * args.batch_size * world / dt, world))
        torch.save(model.state_dict(), os.path.join(snapshot_path, '{}_last_model.pth'.format(args.model)))
        validator.finish(val_models)
        scalars.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return "Training Finished!"

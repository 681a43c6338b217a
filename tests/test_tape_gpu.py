"""The step as a launch tape (mis_hip/lib.py::LaunchTape, step.py::_TapedStep): after two eager steps the trainer records one
step's C-ABI launches, stream dependencies and exchange callbacks and replays them -- one ctypes call per launch instead of the
Python op graph.  Training must be bit-identical to the eager step: same losses, weights, teacher and momentum after several
steps with dropout / DropPath drawing from the device-resident RNG state, fresh input tensors every step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(shape, C, ldt, n):
    g = torch.Generator().manual_seed(7)
    out = []
    for _ in range(n):
        v = torch.rand(shape, generator=g)
        l = torch.randint(0, C, (shape[0],) + tuple(shape[2:]), generator=g).to(ldt)
        out.append((v.cuda(), l.cuda()))
    return out


def _mt(kind, tape):
    from mis_hip.step import MeanTeacherTrainer
    from oracle import filler
    if kind == "swin":
        from config import lite_config
        from networks.vision_transformer import SwinUnet
        from oracle.swin import OracleSwinUnet
        sd0 = filler.fill_state_dict(OracleSwinUnet(4).new_state())
        make = lambda: SwinUnet(lite_config(), num_classes=4)
        shape, C, L, ldt = (4, 1, 224, 224), 4, 2, torch.uint8
    elif kind == "unet2d":
        from networks.net_factory import net_factory
        from oracle.nets import OracleUNet2D
        sd0 = filler.fill_state_dict(OracleUNet2D(1, 4).new_state())
        make = lambda: net_factory("unet", 1, 4)
        shape, C, L, ldt = (8, 1, 128, 128), 4, 4, torch.uint8
    else:
        from networks.net_factory_3d import net_factory_3d
        from oracle.nets import OracleUNet3D
        sd0 = filler.fill_state_dict(OracleUNet3D(2, 1).new_state())
        make = lambda: net_factory_3d("unet_3D", 1, 2)
        shape, C, L, ldt = (2, 1, 64, 64, 64), 2, 1, torch.int64
    m, e = make(), make()
    m.load_state_dict(sd0); e.load_state_dict(sd0)
    m.train(); e.train()
    tr = MeanTeacherTrainer(m, e, labeled_bs=L, num_classes=C, cons_start_iter=0, seed=5, iter_num=1200, use_tape=tape)
    return tr, m, e, _batches(shape, C, ldt, 7)


@pytest.mark.parametrize("kind", ["swin", "unet2d", "unet3d"])
def test_taped_mean_teacher_step_is_bit_identical_to_eager(kind):
    res = []
    for tape in (False, True):
        tr, m, e, batches = _mt(kind, tape)
        losses = []
        for v, l in batches:
            tr.step(v, l)
            losses.append(tr.out.clone())
        torch.cuda.synchronize()
        if tape:
            assert tr._tape is not None and len(tr._tape) > 50          # steps 4.. were replays
        res.append((torch.stack(losses), m.flat_param.clone(), e.flat_param.clone(), tr.momentum_buf.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_taped_cross_teaching_step_is_bit_identical_to_eager():
    from config import lite_config
    from mis_hip.step import CrossTeachingTrainer
    from networks.net_factory import net_factory
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.nets import OracleUNet2D
    from oracle.swin import OracleSwinUnet
    sds = [filler.fill_state_dict(OracleUNet2D(1, 4).new_state()), filler.fill_state_dict(OracleSwinUnet(4).new_state())]
    batches = _batches((4, 1, 224, 224), 4, torch.uint8, 6)
    res = []
    for tape in (False, True):
        models = [net_factory("unet", 1, 4), SwinUnet(lite_config(), num_classes=4)]
        for m, sd in zip(models, sds):
            m.load_state_dict(sd)
            m.train()
        tr = CrossTeachingTrainer(models[0], models[1], labeled_bs=2, num_classes=4, seed=9, iter_num=300, use_tape=tape)
        outs = []
        for v, l in batches:
            o1, o2 = tr.step(v, l)
            outs.append(torch.cat([o1, o2]).clone())
        torch.cuda.synchronize()
        res.append((torch.stack(outs), models[0].flat_param.clone(), models[1].flat_param.clone(), tr.mom1.clone(), tr.mom2.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_tape_refuses_another_input_geometry_and_survives_an_eager_forward_between_replays():
    tr, m, e, batches = _mt("unet2d", True)
    for v, l in batches[:4]:
        tr.step(v, l)
    ref_tr, ref_m, ref_e, _ = _mt("unet2d", False)
    for v, l in batches[:4]:
        ref_tr.step(v, l)
    # a validation-style forward on another geometry (grows scratch buffers the tape points into): replays stay correct
    m.eval()
    with torch.no_grad():
        m.forward_raw(torch.rand(16, 1, 256, 256, device="cuda"), no_backward=True)
    m.train()
    ref_m.eval()
    with torch.no_grad():
        ref_m.forward_raw(torch.rand(16, 1, 256, 256, device="cuda"), no_backward=True)
    ref_m.train()
    for v, l in batches[4:]:
        tr.step(v, l)
        ref_tr.step(v, l)
    torch.cuda.synchronize()
    assert torch.equal(m.flat_param, ref_m.flat_param) and torch.equal(e.flat_param, ref_e.flat_param)
    with pytest.raises(RuntimeError):
        tr.step(batches[0][0][:6], batches[0][1][:6])


def test_taped_cnn_meet_vit_step_rerecords_when_the_ramp_weights_change():
    """CnnMeetVitTrainer passes two HOST floats of iter_num (the ramp weights: they change every 150 iterations and at 1000) to
    its loss tails: the tape is recorded again when they change.  Eight steps across iteration 1050 (a ramp step) -- bit-identical
    to the eager trainer."""
    from config import lite_config
    from mis_hip.step import CnnMeetVitTrainer
    from networks.net_factory import net_factory
    from networks.vision_transformer import SwinUnet
    from oracle import filler
    from oracle.nets import OracleUNet2D
    from oracle.swin import OracleSwinUnet
    sds = [filler.fill_state_dict(OracleUNet2D(1, 4).new_state()), filler.fill_state_dict(OracleSwinUnet(4).new_state())]
    batches = _batches((4, 1, 224, 224), 4, torch.uint8, 8)
    res = []
    for tape in (False, True):
        models = [net_factory("unet", 1, 4), SwinUnet(lite_config(), num_classes=4), SwinUnet(lite_config(), num_classes=4)]
        for m, sd in zip(models, (sds[0], sds[1], sds[1])):
            m.load_state_dict(sd)
            m.train()
        tr = CnnMeetVitTrainer(models[0], models[1], models[2], labeled_bs=2, num_classes=4, seed=9, iter_num=1045, use_tape=tape)
        outs, tapes = [], []
        for v, l in batches:
            o1, o2 = tr.step(v, l)
            outs.append(torch.cat([o1, o2]).clone())
            if tr._tape is not None and all(tr._tape is not t for t in tapes):
                tapes.append(tr._tape)
        torch.cuda.synchronize()
        if tape:
            assert len(tapes) == 2                 # recorded at step 3 and again at iteration 1050
        res.append((torch.stack(outs), models[0].flat_param.clone(), models[1].flat_param.clone(), models[2].flat_param.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)

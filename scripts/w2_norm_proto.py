"""Cost of normalisation-on-read in the 2-D Winograd forward kernel (development tool; run through scripts/w2_norm_proto.sh).

For every 3x3 layer shape of the 2-D UNet at config 2's batch (24 slices of 256 x 256; reference code/networks/unet.py:31-47:
conv - BatchNorm - LeakyReLU) it times
  pass     mis_norm_act_fwd_g: raw -> activation (the pass the fusion would remove)
  plain    mis_conv2d_wino_fwd on the activation (the product path)
  fused    the -DMIS_W2_NORM=1 kernel on the RAW tensor, scale / shift / LeakyReLU applied as the patch is read from LDS
and checks fused == plain(pass(raw)) to fp32 rounding."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import lib as _l, ops  # noqa: E402

L = _l.load()
try:
    L.mis_debug_w2_norm
except AttributeError:
    raise SystemExit("this library was not built with -DMIS_W2_NORM=1 (use scripts/w2_norm_proto.sh)")
L.mis_debug_w2_norm.restype = ctypes.c_int
L.mis_debug_w2_norm.argtypes = [ctypes.c_void_p, ctypes.c_float]
SLOPE = 0.01


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def layer(N, Cin, Cout, S):
    g = torch.Generator(device="cuda").manual_seed(Cin * 1000 + S)
    raw = torch.randn(N, Cin, S, S, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (Cin * 9) ** -0.5
    mean = raw.mean((0, 2, 3)).contiguous()
    rstd = (raw.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt().contiguous()
    gamma = torch.rand(Cin, device="cuda", generator=g) + 0.5
    beta = torch.randn(Cin, device="cuda", generator=g) * 0.1
    act = torch.empty_like(raw)
    wino = ops.conv_wino_select(N, Cin, Cout, 1, S, S, (1, 3, 3))
    assert wino >= ops.WINO2D, (Cin, Cout, S, wino)
    wp = ops.conv_pack(w.view(Cout, Cin, 1, 3, 3), ops.conv_wino_pack_mode(wino, False))
    y_plain = torch.empty(N, Cout, S, S, device="cuda")
    y_fused = torch.empty_like(y_plain)
    ss = torch.stack([gamma * rstd, beta - mean * gamma * rstd], 1).contiguous()      # (scale, shift) per channel

    def f_pass():
        ops.norm_act_fwd(raw.unsqueeze(2), act.unsqueeze(2), False, mean, rstd, gamma, beta, SLOPE)

    def f_plain():
        L.mis_debug_w2_norm(None, 0.0)
        ops.conv_fwd(act.unsqueeze(2), wp, None, y_plain.unsqueeze(2), Cin, Cout, (1, 3, 3), wino=wino)

    def f_fused():
        L.mis_debug_w2_norm(ctypes.c_void_p(ss.data_ptr()), SLOPE)
        ops.conv_fwd(raw.unsqueeze(2), wp, None, y_fused.unsqueeze(2), Cin, Cout, (1, 3, 3), wino=wino)

    t_pass, t_plain, t_fused = timed(f_pass), timed(f_plain), timed(f_fused)
    L.mis_debug_w2_norm(None, 0.0)
    err = (y_fused - y_plain).abs().max().item() / y_plain.abs().max().item()
    assert err < 1e-5, err
    print(f"N{N} {Cin:3d}->{Cout:3d} {S:3d}^2: pass {t_pass:6.1f} us  plain {t_plain:6.1f} us  fused {t_fused:6.1f} us  "
          f"(+{t_fused - t_plain:5.1f} us = {100 * (t_fused / t_plain - 1):4.1f} %; pass + plain {t_pass + t_plain:6.1f})  "
          f"rel err {err:.1e}", flush=True)
    return t_pass, t_plain, t_fused


if __name__ == "__main__":
    # encoder second convolutions, decoder first (after the concatenation) and second convolutions of config 2's UNet
    tot = [0.0, 0.0, 0.0]
    for case in [(24, 16, 16, 256), (24, 32, 32, 128), (24, 64, 64, 64), (24, 128, 128, 32), (24, 256, 256, 16),
                 (24, 32, 16, 256), (24, 64, 32, 128), (24, 128, 64, 64), (24, 256, 128, 32)]:
        for i, t in enumerate(layer(*case)):
            tot[i] += t
    print(f"sum: pass {tot[0]:.0f} us, plain {tot[1]:.0f} us, fused {tot[2]:.0f} us: fused - plain = {tot[2] - tot[1]:+.0f} us "
          f"against {tot[0]:.0f} us of passes", flush=True)

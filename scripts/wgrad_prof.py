"""Where the z-ring Winograd weight-gradient kernel spends its cycles (development tool).

    bash scripts/wgrad_prof.sh                 builds a -DMIS_WR_PROF=1 library in /tmp and runs this file against it

The profiled kernel reads s_memtime (core clock) and s_memrealtime (100 MHz) once at its start and once at its end and leaves
(cycles, ticks, stages) per workgroup: cycles per stage independent of the DVFS state, and the clock the kernel ran at.
MIS_WR_EXTRA="-DMIS_WR_ABL=<bits>" builds the ablations (1 no DMA, 2 no transforms, 4 no patch loads, 8 no barrier, 16 no MFMAs;
results wrong, timing only); scripts/wgrad_prof.sh all runs the whole table."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cv-ssl-mis_amd"))
from mis_hip import lib as _l  # noqa: E402

L = _l.load()


def run(N, Cin, Cout, S, var):
    x = torch.randn(N, Cin, S, S, S, device="cuda")
    dy = torch.randn(N, Cout, S, S, S, device="cuda")
    nb = L.mis_conv3d_wino_wgrad_workspace_bytes(N, Cin, Cout, S, S, S, var)
    ws = torch.empty(nb // 4, device="cuda")
    dw = torch.zeros(Cout, Cin, 3, 3, 3, device="cuda")
    prof = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
    st = L.mis_debug_wgrad_prof(ctypes.c_void_p(prof.data_ptr()))
    if st != 0:
        raise SystemExit("this library was not built with -DMIS_WR_PROF=1 (use scripts/wgrad_prof.sh)")

    def go():
        _l.check(L.mis_conv3d_wino_wgrad(_l.ptr(x), Cin * S ** 3, _l.ptr(dy), Cout * S ** 3, _l.ptr(dw), _l.ptr(ws), nb, N,
                                         Cin, Cout, S, S, S, 0, var, _l.stream_ptr()), "wgrad")
    for _ in range(5):
        go()
    prof.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    p = prof.view(-1, 8).cpu()
    p = p[p[:, 5] > 0].double()
    stages, cyc, ref = p[:, 5], p[:, 4], p[:, 0]           # stages, core-clock cycles and 100 MHz ticks per workgroup
    ghz = cyc / ref * 0.1
    print(f"[{os.environ.get('MIS_WR_TAG', 'full'):>22s}] N{N} {Cin}->{Cout} {S}^3 v{var}: {e0.elapsed_time(e1) * 1e3:7.1f} us  "
          f"{cyc.mean() / stages.mean():7.0f} cycles per stage  core clock {ghz.mean():.3f} GHz  "
          f"(MFMA = 4096 cycles per stage: {4096 * stages.mean() / cyc.mean():.3f} of them)", flush=True)
    L.mis_debug_wgrad_prof(None)


if __name__ == "__main__":
    for case in [(8, 16, 16, 96, 3), (8, 48, 16, 96, 3), (8, 32, 32, 48, 4)]:
        run(*case)

"""Scan the gfx950 ISA of the kernels that mix inline-asm vector instructions with MFMAs for a read-after-write distance hipcc
cannot protect: an MFMA whose A / B operand register was written by a VALU instruction fewer than `MIN_GAP` instructions earlier.

Why: hipcc inserts the wait states a dependent MFMA needs behind VALU writes IT generates; an `asm volatile("v_pk_add_f32 ...")`
is opaque to its hazard recognizer.  Round 4: the scheduler hoisted the first MFMA of a run to two instructions behind the
inline-asm packed add producing its B operand (wino2d_wgrad_kernel<Wg2Cfg<1, 2, 16>>) and lanes 48 - 63 read the register before
it was written -- wrong weight gradients, timing-dependent.  tests/test_oracle_cpu.py runs this over every .hip that has both.

    python scripts/check_mfma_hazard.py [file.hip ...]        exit status 1 and the offending pairs when something is found"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cv-ssl-mis_amd", "csrc")
MIN_GAP = 2          # flagged: the MFMA is the 1st or 2nd instruction behind the write (the failure had ONE instruction in between;
                     # distances of 3 and more occur all over the tested kernels)
VALU = ("v_pk_", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_mov_b64", "v_max_f32",
        "v_accvgpr_read", "v_cndmask")


def _regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S",
                        "--cuda-device-only", path, "-o", out], check=True, cwd=CSRC, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out)]
    lines = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    found = []
    for i, l in enumerate(lines):
        if not l.startswith("v_mfma"):
            continue
        parts = [p.strip() for p in re.split(r",\s*(?![^\[]*\])", l.split(None, 1)[1])]
        src = set()
        for p in parts[1:3]:
            src |= _regs(p)
        for back in range(1, MIN_GAP + 1):
            if i - back < 0:
                break
            pl = lines[i - back]
            if pl.startswith("v_mfma"):
                break                      # an MFMA in between provides the wait states (32 cycles)
            if pl.startswith(VALU) and _regs(pl.split(None, 1)[1].split(",")[0]) & src:
                found.append((back, pl, l))
    return found


def main(argv):
    def mixes(f):
        t = open(os.path.join(CSRC, f)).read()
        return "mfma" in t and ('asm volatile("v_' in t or '#include "wino.h"' in t)
    files = argv or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and mixes(f)]
    bad = 0
    for f in files:
        hits = scan(os.path.join(CSRC, f) if not os.path.isabs(f) else f)
        print(f"{os.path.basename(f)}: {len(hits)} close VALU -> MFMA operand dependencies")
        for h in hits[:8]:
            print("   gap", h[0], "|", h[1][:90], "|", h[2][:80])
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -o <name> -- <cmd>
    python scripts/pmc_mfma.py <dir>/<name>_counter_collection.csv <out.csv>
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): rocprofv3 reports both counters summed over
the 8 XCDs, i.e. GRBM_GUI_ACTIVE is 8 x the elapsed cycles (r01: 361 M "active" cycles for 20.3 ms of kernel time), and the
MFMA counter sums the 1024 SIMDs (256 CUs x 4).  Cross-check: the fraction equals the executed-flop fraction of the
bench (wino_fwd_kernel: 0.50 here, 76.5 TF / 157.3 TF = 0.49 from HIP events)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace(" >", ">").strip()


def main(path, out):
    per = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
    rows = []
    for k, c in per.items():
        act = c.get("GRBM_GUI_ACTIVE", 0.0)
        if act <= 0:
            continue
        rows.append((act, k, cnt[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * act / 8.0)))
    rows.sort(reverse=True)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "gui_active_cycles_total", "mfma_busy_fraction"])
        for act, k, n, frac in rows:
            w.writerow([k, n, int(act), f"{frac:.4f}"])
    for act, k, n, frac in rows[:12]:
        print(f"{k[:70]:72s} {n:5d} launches  MFMA busy {frac:.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:])

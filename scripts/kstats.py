"""Per-kernel totals from a rocprofv3 --kernel-trace CSV: python scripts/kstats.py <kernel_trace.csv> [steps]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace(" >", ">")


def main(path, steps=1):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[k] += 1
    all_t = sum(tot.values())
    print(f"total {all_t / steps / 1e3:.3f} ms/step over {steps} steps, {sum(cnt.values()) // steps} launches/step")
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
        print(f"{t / steps:9.1f} us/step {100 * t / all_t:5.1f}%  x{cnt[k] / steps:6.1f}  avg {t / cnt[k]:8.1f} us  {k[:100]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)

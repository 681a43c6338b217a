#!/bin/bash
# HBM read bytes per launch of the Winograd weight-gradient kernels on the config-3 layer shapes (rocprofv3 --pmc FETCH_SIZE over
# `scripts/wino_bench.py wgrad ab`; read bytes = FETCH_SIZE KiB x 1024 x 2, the guide's gfx950 correction).
export TMPDIR=/tmp
out=${1:-gpurun_out/wgtraffic}
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/p" -o wg --output-format csv -- python scripts/wino_bench.py wgrad ab > "$out/run.log" 2>&1
python - "$out/p/wg_counter_collection.csv" <<'PY'
import csv, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(sys.argv[1], newline="")):
    if row["Counter_Name"] != "FETCH_SIZE" or "wgrad" not in row["Kernel_Name"] or "reduce" in row["Kernel_Name"]:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
    n = re.sub(r"^void |\(.*$", "", n)
    g = int(row["Grid_Size"]) if "Grid_Size" in row else 0
    a = acc[n][g]
    a[0] += 1; a[1] += float(row["Counter_Value"])
for n in sorted(acc):
    for g in sorted(acc[n]):
        c, kib = acc[n][g]
        print(f"{n:60s} grid {g:8d}  launches {c:4d}  read {kib * 2048 / c / 1e6:9.1f} MB per launch")
PY
tail -7 "$out/run.log"
rm -rf "$out/p"

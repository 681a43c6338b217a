#!/bin/bash
# HBM bytes per launch of the 2-D Winograd forward kernels on the config-2 layer shapes (two rocprofv3 --pmc passes over
# scripts/wino2d_bench.py; read bytes = FETCH_SIZE KiB x 1024 x 2 -- the guide's gfx950 correction --, written = WRITE_SIZE KiB x 1024).
export TMPDIR=/tmp
out=${1:-gpurun_out/w2traffic}
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d "$out/p_$c" -o w2 --output-format csv -- python scripts/wino2d_bench.py > "$out/run_$c.log" 2>&1
done
python - "$out/p_FETCH_SIZE/w2_counter_collection.csv" "$out/p_WRITE_SIZE/w2_counter_collection.csv" <<'PY'
import csv, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0, 0.0]))
for fi, path in enumerate(sys.argv[1:3]):
    for row in csv.DictReader(open(path, newline="")):
        if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE") or "wino2d_fwd" not in row["Kernel_Name"]:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
        n = re.sub(r"^void |\(.*$", "", n)
        g = int(row["Grid_Size"]) if "Grid_Size" in row else 0
        a = acc[n][g]
        a[2 * fi] += 1; a[2 * fi + 1] += float(row["Counter_Value"])
for n in sorted(acc):
    for g in sorted(acc[n]):
        c, kib, c2, kib2 = acc[n][g]
        print(f"{n:50s} grid {g:9d}  launches {c:4d}  read {kib * 2048 / max(c, 1) / 1e6:8.1f} MB  written {kib2 * 1024 / max(c2, 1) / 1e6:8.1f} MB per launch")
PY
rm -rf "$out"/p_*

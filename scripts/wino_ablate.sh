#!/bin/bash
# development: time the ablation builds made by scripts/wino_variants.sh (args = labels of v0, v1, ...) and collect SQ counters
# for the baseline build.  Run on the GPU box:  scripts/wino_ablate.sh base noDMA ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for lab in "$@"; do
    echo "== v$i $lab"
    MIS_HIP_LIB=$PWD/cv-ssl-mis_amd/mis_hip/libmis_hip_v$i.so MIS_WINO_DBG=$lab python scripts/wino_bench.py 2>&1 | grep dbg
    i=$((i + 1))
done

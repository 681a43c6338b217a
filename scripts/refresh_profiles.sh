#!/bin/bash
# Re-measure everything that DESIGN.md quotes, on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash scripts/refresh_profiles.sh r05'
# Writes gpurun_out/<tag>/: the default bench JSON line (config 3 + the "others" block + thread-swept CPU baseline),
# the bench lines of the other workloads, rocprofv3 kernel-trace statistics (CSV), and the two PMC passes that
# scripts/pmc_traffic.py turns into profiles/<tag>_<workload>_pmc_traffic.json.  PMC passes run without any trace
# domain but --kernel-trace (separate FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md "HBM").
# The bench lines time the product configuration (teacher forward and weight gradients on side streams); every rocprofv3
# pass runs `bench.py --serial` (side streams off), where a launch's duration is the kernel's own -- the same condition as
# the roofline region inside bench.py, so `AverageNs` of the dominant kernel agrees with `roofline.avg_launch_ms`.
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
for w in unet2d vnet uamt3d swin cross cross224 cnnvit unetr swinunetr; do
    python bench.py --workload $w --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
python bench.py --serial --no-cpu-baseline --no-others > "$out/bench_unet3d_serial.json" 2> "$out/bench_unet3d_serial.err"
for w in unet3d unet2d vnet swin; do
    rocprofv3 --kernel-trace --stats -d "$out/prof_$w" -o "$w" --output-format csv -- \
        python bench.py --workload $w --serial --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-kernel-events > "$out/prof_$w.log" 2>&1
    cp "$out/prof_$w/${w}_kernel_stats.csv" "$out/${tag}_${w}_kernel_stats.csv" 2>/dev/null
done
for w in ${PMC_WORKLOADS:-unet3d swin}; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_${w}_$c" -o "$w" --output-format csv -- \
            python bench.py --workload $w --serial --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-kernel-events \
            > "$out/pmc_${w}_$c.log" 2>&1
    done
    python scripts/pmc_traffic.py "$out/pmc_${w}_FETCH_SIZE/${w}_counter_collection.csv" \
        "$out/pmc_${w}_WRITE_SIZE/${w}_counter_collection.csv" "$out/${tag}_${w}_pmc_traffic.json" \
        "bench.py --workload $w, 4 steps" > "$out/pmc_${w}.txt" 2>&1
    rm -rf "$out/pmc_${w}_FETCH_SIZE" "$out/pmc_${w}_WRITE_SIZE"
done
# matrix-pipe busy fraction per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles)), config 3, serial
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$out/pmc_mfma" -o unet3d --output-format csv -- \
    python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-kernel-events > "$out/pmc_mfma.log" 2>&1
python scripts/pmc_mfma.py "$out/pmc_mfma/unet3d_counter_collection.csv" "$out/${tag}_unet3d_pmc_mfma.csv" > "$out/pmc_mfma.txt" 2>&1
rm -rf "$out/pmc_mfma"
rm -rf "$out"/prof_*/*_kernel_trace.csv
ls -R "$out" | head -60

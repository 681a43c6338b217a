#!/bin/bash
# Round 6: the subset of scripts/refresh_profiles.sh that the round's changes touch (token workloads + the headline), same method.
#   gpurun --timeout 2400 -- 'bash scripts/refresh_profiles_r06.sh r06'
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for w in swin cross cross224 cnnvit unetr; do
    python bench.py --workload $w --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
MIS_STEP_TAPE=0 python bench.py --workload swin --no-cpu-baseline > "$out/bench_swin_eager.json" 2> "$out/bench_swin_eager.err"
MIS_BENCH_GRAPH=1 python bench.py --workload swin --no-cpu-baseline > "$out/bench_swin_hipgraph.json" 2> "$out/bench_swin_hipgraph.err"
MIS_GEMM_REGA=0 python bench.py --workload swin --no-cpu-baseline > "$out/bench_swin_norega.json" 2> "$out/bench_swin_norega.err"
for w in unet3d swin cross; do
    rocprofv3 --kernel-trace --stats -d "$out/prof_$w" -o "$w" --output-format csv -- \
        python bench.py --workload $w --serial --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-kernel-events --no-traffic > "$out/prof_$w.log" 2>&1
    cp "$out/prof_$w/${w}_kernel_stats.csv" "$out/${tag}_${w}_kernel_stats.csv" 2>/dev/null
done
# the product configuration (side streams + tape) of SwinUnet: launches per step as the profiler counts them
rocprofv3 --kernel-trace --stats -d "$out/prof_swin_product" -o swin --output-format csv -- \
    python bench.py --workload swin --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-kernel-events --no-traffic > "$out/prof_swin_product.log" 2>&1
cp "$out/prof_swin_product/swin_kernel_stats.csv" "$out/${tag}_swin_product_kernel_stats.csv" 2>/dev/null
for w in swin unet3d; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_${w}_$c" -o "$w" --output-format csv -- \
            python bench.py --workload $w --serial --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-kernel-events --no-traffic \
            > "$out/pmc_${w}_$c.log" 2>&1
    done
    python scripts/pmc_traffic.py "$out/pmc_${w}_FETCH_SIZE/${w}_counter_collection.csv" \
        "$out/pmc_${w}_WRITE_SIZE/${w}_counter_collection.csv" "$out/${tag}_${w}_pmc_traffic.json" \
        "bench.py --workload $w, serial" > "$out/pmc_${w}.txt" 2>&1
    rm -rf "$out/pmc_${w}_FETCH_SIZE" "$out/pmc_${w}_WRITE_SIZE"
done
python scripts/gemm_nt_bench.py --split > "$out/${tag}_gemm_nt_staged.txt" 2>&1
python scripts/gemm_nt_bench.py --rega > "$out/${tag}_gemm_nt_rega.txt" 2>&1
python scripts/gemm_tn_bench.py > "$out/${tag}_gemm_tn.txt" 2>&1
rm -rf "$out"/prof_*/*_kernel_trace.csv
ls "$out" | head -60
